cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_encoder_ops.py tests/test_gpu_encoder_net.py tests/test_gpu_model.py tests/test_gpu_dist.py -q -m gpu -x 2>&1 | tail -4
for T in 256 512 128; do echo "== tall $T"; SURS_CONV_TALL_MIN_WG=$T python tools/enc_time.py 512 fp32 2>&1 | grep -v "^[EW]20" | tail -4; done
