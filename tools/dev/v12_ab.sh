#!/bin/bash
# v12_ab.sh FIELD lib1 lib2 ...: 128-plane slab times of kernel 12 under each library (and of kernel 10 under the first), two rounds,
# one process per measurement, on whatever GPU this runs on.  "default" = the in-tree library.
field=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset SURS_LIB_PATH; else export SURS_LIB_PATH=$PWD/abl/libsurs_$lib.so; fi
    echo "round $round $lib v12: $(timeout 300 python tools/dev/v12_time.py 12 $field bf16 6 2>/dev/null | tail -1)"
  done
  unset SURS_LIB_PATH
  echo "round $round default v10: $(timeout 300 python tools/dev/v12_time.py 10 $field bf16 6 2>/dev/null | tail -1)"
done
