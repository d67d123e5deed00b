cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s2
rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_octree.py -x -q > $O/octree_tests.log 2>&1
tail -3 $O/octree_tests.log
python tools/gpu_octree_time.py 512 fp32 > $O/octree_time.log 2>&1
cat $O/octree_time.log
python tools/enc_time.py 512 fp32 > $O/enc_time.log 2>&1
cat $O/enc_time.log
rocprofv3 --kernel-trace --output-format csv -d $O/enc_trace -o enc -- python3 tools/enc_time.py 512 fp32 > $O/enc_trace.log 2>&1
find $O -name "*.csv" | head
