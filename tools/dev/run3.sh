cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s2
mkdir -p $O
python -m pytest tests/test_gpu_query.py -x -q -k "point_runs" > $O/tests_runs.log 2>&1
tail -15 $O/tests_runs.log
python -m pytest tests/test_gpu_model.py -x -q -k "facade" > $O/tests_facade.log 2>&1
tail -8 $O/tests_facade.log
