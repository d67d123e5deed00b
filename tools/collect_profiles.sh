#!/bin/bash
# collect_profiles.sh rNN: copies what tools/profile_round.sh + profile_round_extra.sh left under gpurun_out/prof/ (scratch) into
# profiles/ (tracked) under the round's names, and reduces the --pmc passes (tools/pmc_summarize.py rNN -> profiles/pmc_summary.json).
set -e
T=${1:?round tag, e.g. r06}
cd "$(dirname "$0")/.."
S=gpurun_out/prof
cp $S/bench_line.json profiles/${T}_bench_line.json
cp $S/stats/bench_kernel_stats.csv profiles/${T}_bench_kernel_stats.csv
cp $S/stats32/bench_kernel_stats.csv profiles/${T}_bench_fp32_kernel_stats.csv
cp $S/sweep/sweep_kernel_stats.csv profiles/${T}_sweep_only_kernel_stats.csv
cp $S/mc/mc_kernel_stats.csv profiles/${T}_mc_kernel_stats.csv
cp $S/enc32/enc_kernel_stats.csv profiles/${T}_encoder_fp32_kernel_stats.csv
cp $S/oct/oct_kernel_stats.csv profiles/${T}_octree_kernel_stats.csv
cp $S/loop32/loop_kernel_stats.csv profiles/${T}_reference_loop_fp32_kernel_stats.csv
cp $S/loop16/loop_kernel_stats.csv profiles/${T}_reference_loop_bf16_kernel_stats.csv
for f in body noise; do   # (the tool prints its progress lines in front of the JSON object)
  python3 - "$S/slab_stage_times_$f.json" "profiles/${T}_slab_stage_times_$f.json" <<'PY'
import json, sys
txt = open(sys.argv[1]).read()
json.dump(json.loads(txt[txt.index('{"resolution"'):]), open(sys.argv[2], "w"), indent=1)
PY
done
python3 tools/pmc_summarize.py $T > /dev/null
ls -la profiles/${T}_* profiles/pmc_summary.json | awk '{print $5, $9}'
