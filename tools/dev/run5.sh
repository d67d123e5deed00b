cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s2
mkdir -p $O
export OCTREE_ONLY=1
rocprofv3 --kernel-trace --output-format csv -d $O/oct_trace -o oct -- python3 tools/gpu_octree_time.py 512 fp32 > $O/oct_trace.log 2>&1
tail -2 $O/oct_trace.log
