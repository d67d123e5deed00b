cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s2; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 600 $O/bench.err
python -c "
import json;d=json.load(open('$O/bench_line.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['config'].get('fp32_ms_per_step'))
c=d['config']
for k in ('reference_loop','reference_loop_reduced','octree_mode','subject_pipeline'):
    print(k, json.dumps(c.get(k))[:400])
"
