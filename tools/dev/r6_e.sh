cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6e; mkdir -p $O
python -m pytest tests/test_gpu_encoder_net.py tests/test_gpu_encoder_ops.py -q -m gpu -x > $O/t1.log 2>&1; tail -12 $O/t1.log
for S in 0 1; do
  echo "== bicubic block $S"; SURS_BICUBIC_BLOCK=$S python tools/enc_time.py 512 fp32 2>&1 | grep -v "^[EW]20" | tail -4
done
python -m pytest tests/test_gpu_model.py -q -m gpu -x > $O/t2.log 2>&1; tail -5 $O/t2.log
