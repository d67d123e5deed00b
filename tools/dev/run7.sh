cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s2
mkdir -p $O
SURS_ENC_GRAPH=0 python tools/enc_time.py 512 fp32 2>&1 | tail -4
python tools/enc_time.py 512 fp32 2>&1 | tail -4
export OCTREE_ONLY=1
python tools/gpu_octree_time.py 512 fp32 2>&1 | tail -1
SURS_GRID_F32_PASSES=1 python tools/gpu_octree_time.py 512 fp32 2>&1 | tail -1
unset OCTREE_ONLY
python -m pytest tests/test_gpu_model.py tests/test_gpu_encoder_ops.py tests/test_gpu_input.py -x -q > $O/tests_graph.log 2>&1
tail -5 $O/tests_graph.log
