#!/bin/bash
# Compile surs_query.hip for gfx950 and print the resource usage of the v12 kernels (+ where scratch traffic sits relative to the MFMAs).
cd "$(dirname "$0")/../../super-resolution-3d-human-shape-from-a-single-low-resolution-image_amd/csrc"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-function --cuda-device-only -S "$@" \
   -Rpass-analysis=kernel-resource-usage surs_query.hip -o /tmp/q.s 2>&1 | grep -A10 "grid_mlp_kernel_v12ILi1" | grep -E "VGPRs|Scratch|Spill|error" 
grep -E "error" /tmp/q.err 2>/dev/null
python3 - <<'PY'
import re
s=open('/tmp/q.s').read()
m=re.search(r'^_ZN4surs19grid_mlp_kernel_v12ILi1EEEvNS_8GridArgsE:(.*?)\.Lfunc_end', s, re.S|re.M)
body=m.group(1).split('\n')
mf=[i for i,l in enumerate(body) if 'v_mfma' in l]
sc=[i for i,l in enumerate(body) if 'scratch_' in l]
print(len(body),'lines; mfma',len(mf),'scratch',len(sc))
if mf:
    lo,hi=mf[0],mf[-1]
    print('scratch ops before first mfma',sum(1 for i in sc if i<lo),'between',sum(1 for i in sc if lo<=i<=hi),'after',sum(1 for i in sc if i>hi))
PY
