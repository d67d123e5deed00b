cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export OCTREE_ONLY=1
python tools/gpu_octree_time.py 512 fp32 2>&1 | tail -1
SURS_GRID_F32_PASSES=1 python tools/gpu_octree_time.py 512 fp32 2>&1 | tail -1
python tools/gpu_octree_time.py 512 fp32 2>&1 | tail -1
SURS_GRID_F32_PASSES=1 python tools/gpu_octree_time.py 512 fp32 2>&1 | tail -1
