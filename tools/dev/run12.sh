cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s2; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/bench_trace -o bench -- python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $O/bench_traced.json 2> $O/bench_traced.err
tail -c 300 $O/bench_traced.json
