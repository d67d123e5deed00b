cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== new"; python tools/dev/sweep_hash.py 256 2>&1 | grep -v "^[EW]20" | tail -3
echo "== old colsum"; SURS_LIB_PATH=$PWD/abl/libsurs_oldcolsum.so python tools/dev/sweep_hash.py 256 2>&1 | grep -v "^[EW]20" | tail -3
python -m pytest tests/test_gpu_query.py tests/test_gpu_fullvolume.py tests/test_gpu_octree.py -q -m gpu -x 2>&1 | tail -3
