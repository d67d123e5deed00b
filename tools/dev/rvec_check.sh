cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_query.py -q -m gpu -x 2>&1 | tail -3
for S in 1 0 1 0; do for P in fp32 bf16; do echo -n "rvec_small $S: "; SURS_RVEC_SMALL=$S python tools/gpu_points_loop.py $P 80 2>&1 | grep -v "^[EW]20" | tail -1; done; done
python -m pytest tests/test_gpu_octree.py tests/test_gpu_model.py -q -m gpu -x 2>&1 | tail -2
