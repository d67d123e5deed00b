#!/bin/bash
# v11_ab.sh FIELD lib...: 128-plane slab of the fp32-grade column kernel per library (abl/libsurs_NAME.so), two rounds, min / median ms, volume hash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
field=$1; shift
for round in 1 2; do
for l in "$@"; do
  echo -n "$l: "; SURS_LIB_PATH=abl/libsurs_$l.so python tools/dev/v12_time.py 11 $field fp32 6 2>&1 | tail -1
done
done
