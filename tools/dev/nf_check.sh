cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_query.py -q -m gpu -x -k "nonfinite or point_runs" 2>&1 | tail -3
for T in 0 1 0 1; do for P in fp32 bf16; do echo -n "torch check $T: "; LOOP_TORCH_FINITE_CHECK=$T python tools/gpu_points_loop.py $P 80 2>&1 | grep -v "^[EW]20" | tail -1; done; done
python -m pytest tests/test_gpu_model.py -q -m gpu -x 2>&1 | tail -2
