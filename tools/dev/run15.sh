cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export OCTREE_ONLY=1
for i in 1 2 3; do
SURS_AB_CONVERT_LATE=1 python tools/gpu_octree_time.py 512 fp32 2>&1 | tail -1
python tools/gpu_octree_time.py 512 fp32 2>&1 | tail -1
done
