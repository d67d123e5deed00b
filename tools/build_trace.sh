#!/bin/bash
# Builds the diagnostic variant of the library (in-kernel phase stamps of the v3 column kernel) into abl/libsurs_trace.so.
# On the GPU box: SURS_V3_TRACE=1 SURS_GRID_KERNEL=3 SURS_LIB_PATH=abl/libsurs_trace.so python tools/gpu_grid_time.py
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
S="$ROOT/super-resolution-3d-human-shape-from-a-single-low-resolution-image_amd/csrc"
mkdir -p "$ROOT/abl"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -DSURS_V3_TRACE "$@" -shared -o "$ROOT/abl/libsurs_trace.so" \
    "$S"/*.hip "$S"/*.cpp
