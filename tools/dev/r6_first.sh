# round 6, first GPU call: the whole GPU suite (with the new 8-rank / normals cases), the bench line, the encoder's kernel trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6a; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/gputest_full.log 2>&1
tail -6 $O/gputest_full.log
python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 600 $O/bench.err
python -c "
import json;d=json.load(open('$O/bench_line.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'])
print({k:v for k,v in d['config'].items() if not isinstance(v,(dict,list,str))})
"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -o enc -- python3 tools/enc_time.py 512 fp32 > $O/enc_time.log 2>&1
tail -5 $O/enc_time.log
ls $O/enc
