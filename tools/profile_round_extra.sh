#!/bin/bash
# Run ON THE GPU BOX after tools/profile_round.sh: the rest of a round's evidence under gpurun_out/prof/ -
# marching-cubes, encoder and octree kernel statistics, the slab-mode rank timelines.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/mc -o mc -- python3 tools/gpu_mc_time.py 512 > $O/mc_time.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc32 -o enc -- python3 tools/enc_time.py 512 fp32 > $O/enc_time_fp32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sweep -o sweep -- python3 tools/gpu_grid_once.py 512 bf16 > $O/sweep_only.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/oct -o oct -- python3 tools/gpu_octree_time.py > $O/octree_time.log 2>&1
python3 tools/gpu_slab_stage_times.py 512 bf16 body > $O/slab_stage_times_body.json 2> $O/slab_body.err
python3 tools/gpu_slab_stage_times.py 512 bf16 noise > $O/slab_stage_times_noise.json 2> $O/slab_noise.err
find $O -name "*stats.csv" | head -20
rocprofv3 --kernel-trace --stats --output-format csv -d $O/loop32 -o loop -- python3 tools/gpu_points_loop.py fp32 40 grid > $O/points_loop_fp32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/loop16 -o loop -- python3 tools/gpu_points_loop.py bf16 40 grid > $O/points_loop_bf16.log 2>&1
