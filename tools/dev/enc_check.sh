cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for T in 0 1; do python tools/enc_time.py 512 fp32 2>&1 | grep -v "^[EW]20" | tail -4; done
python -m pytest tests/test_gpu_encoder_ops.py tests/test_gpu_encoder_net.py -q -m gpu -x 2>&1 | tail -3
