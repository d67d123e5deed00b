cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/dev/v12_ab.sh body default before
bash tools/dev/v12_ab.sh noise default before
python -m pytest tests/test_gpu_fullvolume.py tests/test_gpu_query.py -q -m gpu -x 2>&1 | tail -2
