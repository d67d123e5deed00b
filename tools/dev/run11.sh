cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2 3 4; do python tools/enc_time.py 512 fp32 > /tmp/enc_$i.log 2>&1 & done
wait
grep -h digest /tmp/enc_*.log
echo "--- sequential"
python tools/enc_time.py 512 fp32 2>&1 | grep digest
python tools/enc_time.py 512 fp32 2>&1 | grep digest
