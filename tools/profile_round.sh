#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repository root: collects this round's evidence under gpurun_out/prof/.
#   1. rocprofv3 --kernel-trace --stats of the default bench.py command (per-kernel time),
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, L2 hits, clock, MFMA busy cycles) of the column-kernel sweep at R=512, as the MI355X guide prescribes,
#   3. the bench line itself (un-profiled).
# Copy what is to be judged into profiles/ afterwards (tools/pmc_summarize.py writes profiles/pmc_summary.json).
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof
mkdir -p $O
python3 bench.py > $O/bench_line.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --no-cpu-baseline > $O/bench_profiled.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o fetch -- python3 tools/gpu_grid_once.py 512 bf16 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o write -- python3 tools/gpu_grid_once.py 512 bf16 > $O/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -o l2 -- python3 tools/gpu_grid_once.py 512 bf16 > $O/pmc_l2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/pmc_clk -o clk -- python3 tools/gpu_grid_once.py 512 bf16 > $O/pmc_clk.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_mfma -o mfma -- python3 tools/gpu_grid_once.py 512 bf16 > $O/pmc_mfma.log 2>&1
find $O -name "*.csv" | head -30
