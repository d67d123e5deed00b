cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s2
mkdir -p $O
SURS_CC_PARTS=0 python tools/gpu_grid_time.py 512 > $O/grid_time_cc3.log 2>&1; cat $O/grid_time_cc3.log
python tools/gpu_grid_time.py 512 > $O/grid_time_cc1.log 2>&1; cat $O/grid_time_cc1.log
SURS_CC_PARTS=0 python tools/gpu_grid_time.py 512 >> $O/grid_time_cc3.log 2>&1; tail -1 $O/grid_time_cc3.log
python tools/gpu_grid_time.py 512 >> $O/grid_time_cc1.log 2>&1; tail -1 $O/grid_time_cc1.log
python -m pytest tests/test_gpu_precision.py tests/test_gpu_fullvolume.py tests/test_gpu_query.py -x -q > $O/tests_cc1.log 2>&1
tail -5 $O/tests_cc1.log
