cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6f; mkdir -p $O
python -m pytest tests/test_gpu_query.py tests/test_gpu_model.py -q -m gpu -x > $O/t1.log 2>&1; tail -4 $O/t1.log
python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 300 $O/bench.err
python -c "
import json;d=json.load(open('$O/bench_line.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'])
print({k:v for k,v in d['config'].items() if not isinstance(v,(dict,list,str))})
print(d['config']['reference_loop']['ms_per_50k_chunk_passes'], d['config']['reference_loop_reduced']['ms_per_50k_chunk_passes'])
"
python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_query.py --deselect tests/test_gpu_model.py > $O/t2.log 2>&1; tail -4 $O/t2.log
