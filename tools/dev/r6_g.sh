cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for S in 1 0; do for P in fp32 bf16; do
  echo -n "speculate $S: "; SURS_POINT_RUNS_SPECULATE=$S python tools/gpu_points_loop.py $P 80 2>&1 | grep -v "^[EW]20" | tail -1
done; done; done
python -m pytest tests/test_gpu_query.py tests/test_gpu_mc.py -q -m gpu -x 2>&1 | tail -3
