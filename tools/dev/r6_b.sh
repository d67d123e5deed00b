cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6b; mkdir -p $O
python -m pytest tests/test_gpu_encoder_net.py tests/test_gpu_model.py -q -m gpu -x > $O/t1.log 2>&1; tail -15 $O/t1.log
for N in 1 0; do for T in 512 256 128; do
  echo "== native $N threshold $T"; SURS_ENC_NATIVE=$N SURS_CONV_BIG_MIN_WG=$T python tools/enc_time.py 512 fp32 2>&1 | grep -v "^[EW]20" | tail -4
done; done
python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_encoder_net.py --deselect tests/test_gpu_model.py > $O/t2.log 2>&1; tail -5 $O/t2.log
