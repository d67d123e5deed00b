#!/bin/bash
# Builds timing-only variants of the library with parts of the column kernel removed (SURS_ABL=n) into gpurun_out/abl/.
# Run here (no GPU needed); then on the GPU box: for n in 0..6: SURS_LIB_PATH=gpurun_out/abl/libsurs_abl$n.so python tools/gpu_probe.py
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC="$ROOT/super-resolution-3d-human-shape-from-a-single-low-resolution-image_amd/csrc"
mkdir -p "$ROOT/abl"
for n in 0 1 2 3 4 5 6; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -DSURS_ABL=$n -shared -o "$ROOT/abl/libsurs_abl$n.so" \
      "$SRC"/surs_query.hip "$SRC"/surs_mc.hip "$SRC"/surs_encoder.hip "$SRC"/surs_octree.hip "$SRC"/surs_pack.cpp "$SRC"/surs_api.cpp "$SRC"/surs_obj.cpp &
done
wait
ls -la "$ROOT/abl"
