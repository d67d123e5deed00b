cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_octree.py -x -q -s -k "reduced_precision" 2>&1 | grep -E "octree|passed|failed|Error|assert" | head -20
export OCTREE_ONLY=1
python tools/gpu_octree_time.py 512 fp32 2>&1 | tail -1
SURS_OCT_SWEEP=1 python tools/gpu_octree_time.py 512 bf16 2>&1 | tail -1
SURS_OCT_SWEEP=1 python tools/gpu_octree_time.py 512 fp16 2>&1 | tail -1
