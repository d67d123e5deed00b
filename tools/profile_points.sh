#!/bin/bash
# Run ON THE GPU BOX from the repository root: rocprofv3 evidence for the fp32 point path (surs_query_points, 2 000 000 points):
# per-kernel statistics and separate --pmc passes (MFMA busy + clock, FETCH_SIZE, WRITE_SIZE) -> gpurun_out/prof_points/.
# tools/points_pmc_summarize.py rNN reduces them to profiles/rNN_points_pmc_summary.json.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_points
rm -rf $O
mkdir -p $O
python3 tools/gpu_points_time.py 50000 400000 2000000 > $O/points_time.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o points -- python3 tools/gpu_points_time.py 2000000 > $O/stats.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o mfma -- python3 tools/gpu_points_time.py 2000000 > $O/pmc_mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o fetch -- python3 tools/gpu_points_time.py 2000000 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o write -- python3 tools/gpu_points_time.py 2000000 > $O/pmc_write.log 2>&1
cat $O/points_time.log | grep -v amdgpu
