cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/dev/v12_ab.sh body default before
bash tools/dev/v12_ab.sh noise default before
mkdir -p gpurun_out/fl
for L in default before; do
  if [ "$L" = default ]; then unset SURS_LIB_PATH; else export SURS_LIB_PATH=$PWD/abl/libsurs_$L.so; fi
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/fl/write_$L -o w -- python3 tools/gpu_grid_once.py 512 bf16 > gpurun_out/fl/write_$L.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fl/fetch_$L -o f -- python3 tools/gpu_grid_once.py 512 bf16 > gpurun_out/fl/fetch_$L.log 2>&1
done
unset SURS_LIB_PATH
python -m pytest tests/test_gpu_fullvolume.py -q -m gpu -x 2>&1 | tail -2
