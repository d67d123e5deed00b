cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6d; mkdir -p $O
python -m pytest tests/test_gpu_encoder_net.py tests/test_gpu_encoder_ops.py -q -m gpu -x > $O/t1.log 2>&1; tail -12 $O/t1.log
for S in 1 0; do
  echo "== separate sum $S"; SURS_ENC_SEPARATE_SUM=$S python tools/enc_time.py 512 fp32 2>&1 | grep -v "^[EW]20" | tail -4
done
echo "== conv trace"; SURS_LIB_PATH=$PWD/abl/libsurs_convtrace.so SURS_CONV_TRACE=1 python tools/dev/conv_trace.py 2>&1 | grep -v "^[EW]20" | grep -v "^conv_x3.*\n" | awk 'NR%3!=1' | tail -30
python -m pytest tests/test_gpu_model.py tests/test_gpu_dist.py -q -m gpu -x > $O/t2.log 2>&1; tail -5 $O/t2.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -o enc -- python3 tools/enc_time.py 512 fp32 > $O/enc_time.log 2>&1
