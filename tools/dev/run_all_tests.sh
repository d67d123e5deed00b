cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s2; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/gputest_full.log 2>&1
tail -6 $O/gputest_full.log
