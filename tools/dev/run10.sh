cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SURS_CONV_TALL=0 python tools/enc_time.py 512 fp32 2>&1 | tail -4
python tools/enc_time.py 512 fp32 2>&1 | tail -4
SURS_CONV_TALL=0 python tools/enc_time.py 512 fp32 2>&1 | tail -4
python tools/enc_time.py 512 fp32 2>&1 | tail -4
