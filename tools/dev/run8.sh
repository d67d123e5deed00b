cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s2; mkdir -p $O
python -m pytest tests/test_gpu_model.py tests/test_gpu_encoder_ops.py tests/test_gpu_input.py tests/test_gpu_dist.py -x -q > $O/tests_graph.log 2>&1
tail -5 $O/tests_graph.log
