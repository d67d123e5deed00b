cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in enc2 enc1; do
SURS_PREP_STREAM=$v python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pipelined $v', d['ms_per_step'], d['config']['stage_ms_rank0'])"
done
SURS_SWEEP_PIPELINE=0 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('serial   ', d['ms_per_step'], d['config']['stage_ms_rank0'])"
SURS_PREP_STREAM=enc2 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pipelined enc2', d['ms_per_step'], d['config']['stage_ms_rank0'])"
