#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repository root: collects this round's evidence under gpurun_out/prof/.
#   1. the bench line itself (un-profiled),
#   2. rocprofv3 --kernel-trace --stats of the same bench.py command without the CPU leg / extras (per-kernel time),
#   3. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, L2 hits, clock, MFMA busy cycles) of the column-kernel sweep at R=512,
#      for the bf16 kernel (v10) and the fp32-grade kernel (v11), as the MI355X guide prescribes (one counter group per run).
# Copy what is to be judged into profiles/ afterwards: tools/pmc_summarize.py rNN writes profiles/pmc_summary.json.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof
rm -rf $O
mkdir -p $O
python3 bench.py > $O/bench_line.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --no-cpu-baseline --no-extras > $O/bench_profiled.json 2> $O/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats32 -o bench -- python3 bench.py --no-cpu-baseline --no-extras --precision fp32 > $O/bench_profiled_fp32.json 2> $O/stats32.err
for P in bf16 fp32; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$P -o fetch -- python3 tools/gpu_grid_once.py 512 $P > $O/pmc_fetch_$P.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$P -o write -- python3 tools/gpu_grid_once.py 512 $P > $O/pmc_write_$P.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2_$P -o l2 -- python3 tools/gpu_grid_once.py 512 $P > $O/pmc_l2_$P.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_clk_$P -o clk -- python3 tools/gpu_grid_once.py 512 $P > $O/pmc_clk_$P.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_mfma_$P -o mfma -- python3 tools/gpu_grid_once.py 512 $P > $O/pmc_mfma_$P.log 2>&1
done
# SQ counters of the bf16 column kernel (two passes of <= 8 counters) and the fp32-grade one (one pass)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq1_bf16 -o sq -- python3 tools/gpu_grid_once.py 512 bf16 > $O/pmc_sq1_bf16.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_sq2_bf16 -o sq -- python3 tools/gpu_grid_once.py 512 bf16 > $O/pmc_sq2_bf16.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq1_fp32 -o sq -- python3 tools/gpu_grid_once.py 512 fp32 > $O/pmc_sq1_fp32.log 2>&1
sha256sum super-resolution-3d-human-shape-from-a-single-low-resolution-image_amd/libsurs_hip.so > $O/lib_sha256.txt
find $O -name "*.csv" | head -40
