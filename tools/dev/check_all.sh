cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6i; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/gputest_full.log 2>&1; tail -4 $O/gputest_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 8 --steps 2 --warmup 1 --resolution 128 --backend gloo > $O/bench8.json 2> $O/bench8.err; tail -c 400 $O/bench8.err; python -c "
import json
l=[x for x in open('$O/bench8.json') if x.startswith('{')]
d=json.loads(l[0]); print(d['n_gpus'], d['ms_per_step'], d['scaling'], d['config']['parallelism'], d['config']['replicas']['ms_per_step'])"
