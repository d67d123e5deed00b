#!/bin/bash
# build_variant.sh NAME [-Dflags...]: abl/libsurs_NAME.so = the library with surs_query.hip rebuilt under the given flags
set -e
name=$1; shift
cd "$(dirname "$0")/../../super-resolution-3d-human-shape-from-a-single-low-resolution-image_amd/csrc"
d=/tmp/bv_$name; mkdir -p $d
for f in surs_encoder.hip surs_mc.hip surs_octree.hip surs_api.cpp surs_obj.cpp surs_pack.cpp; do cp build/$f.o $d/; done
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-function "$@" -c surs_query.hip -o $d/surs_query.hip.o
mkdir -p ../../abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../abl/libsurs_$name.so $d/*.o
echo built abl/libsurs_$name.so
