cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6c; mkdir -p $O
python -m pytest tests/test_gpu_encoder_net.py -q -m gpu -x > $O/t1.log 2>&1; tail -12 $O/t1.log
for S in 1 0; do
  echo "== separate sum $S"; SURS_ENC_SEPARATE_SUM=$S python tools/enc_time.py 512 fp32 2>&1 | grep -v "^[EW]20" | tail -4
done
echo "== conv trace"; SURS_LIB_PATH=$PWD/abl/libsurs_convtrace.so SURS_CONV_TRACE=1 python tools/dev/conv_trace.py 2>&1 | grep -v "^[EW]20" | tail -40
python -m pytest tests/test_gpu_model.py tests/test_gpu_dist.py -q -m gpu -x > $O/t2.log 2>&1; tail -5 $O/t2.log
