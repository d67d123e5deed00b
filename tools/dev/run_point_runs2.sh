cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_query.py tests/test_gpu_model.py -x -q -k "point_runs or facade or grid_fp32 or column_kernel" 2>&1 | tail -2
python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); c=d['config']
print(d['ms_per_step'], c['stage_ms_rank0'])
for k in ('reference_loop','reference_loop_reduced'): print(k, c[k]['ms_per_50k_chunk'], c[k]['value'])
print('octree', c['octree_mode']['octree']['seconds'])"
