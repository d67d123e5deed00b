cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('taper  ', d['ms_per_step'], d['config']['stage_ms_rank0'])"
SURS_SLAB_COLUMNS=32768 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('equal64', d['ms_per_step'], d['config']['stage_ms_rank0'])"
done
