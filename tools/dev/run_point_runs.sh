cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s2
mkdir -p $O
python -m pytest tests/test_gpu_query.py tests/test_gpu_model.py -x -q -s -k "point_runs or facade" 2>&1 | grep -E "point runs|passed|failed|Error" | tail -12
for P in fp32 bf16; do
python tools/gpu_points_loop.py $P 40 grid 2>&1 | tail -1
SURS_POINT_RUNS=0 python tools/gpu_points_loop.py $P 40 grid 2>&1 | tail -1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/loop_fp32 -o loop -- python3 tools/gpu_points_loop.py fp32 40 grid > $O/loop_fp32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/loop_bf16 -o loop -- python3 tools/gpu_points_loop.py bf16 40 grid > $O/loop_bf16.log 2>&1
