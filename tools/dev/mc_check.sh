cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_mc.py tests/test_gpu_octree.py tests/test_gpu_dist.py -q -m gpu -x 2>&1 | tail -3
for R in 66 258; do echo "== ring $R"; SURS_MC_RING=$R python tools/gpu_mc_time.py 512 2>&1 | grep -v "^[EW]20" | tail -4; SURS_MC_RING=$R python tools/gpu_octree_time.py 2>&1 | grep -v "^[EW]20" | tail -2; done
