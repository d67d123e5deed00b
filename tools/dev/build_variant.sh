#!/bin/bash
# build_variant.sh NAME FILE [-Dflags...]: abl/libsurs_NAME.so = the library with csrc/FILE rebuilt under the given flags
set -e
name=$1; file=$2; shift; shift
cd "$(dirname "$0")/../../super-resolution-3d-human-shape-from-a-single-low-resolution-image_amd/csrc"
d=/tmp/bv_$name; rm -rf $d; mkdir -p $d
for f in build/*.o; do [ "$(basename $f)" != "$file.o" ] && cp $f $d/; done
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-function "$@" -c $file -o $d/$file.o
mkdir -p ../../abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../abl/libsurs_$name.so $d/*.o
echo built abl/libsurs_$name.so
