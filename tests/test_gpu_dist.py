"""BASELINE configs[3]'s path on one GPU: dist.reconstruction_sharded with 2 and 3 processes (ragged slabs) against the
single-process reconstruction - meshes bit-identical, both precisions, one-piece and streamed per-slab extraction
(tools/gpu_slab_check.py; gloo with host staging, since RCCL refuses two ranks on one device)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.parametrize("world,R", [(2, 40), (3, 50)])
def test_sharded_reconstruction_equals_single_process(world, R):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_slab_check.py"), str(world), str(R)], capture_output=True,
                       text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.count("slab == one piece") == 4 and "MISMATCH" not in r.stdout, r.stdout
    assert "flat field raises on every rank" in r.stdout and "no error on a flat field" not in r.stdout, r.stdout


@pytest.mark.parametrize("mode", ["slab", "replicas"])
def test_bench_multi_rank_paths_on_one_gpu(mode):
    """bench.py's N > 1 code paths (BASELINE configs[3] = slab, configs[4] = replicas), launched as the driver launches them
    (torch.distributed.run, one process per rank) but with `--backend gloo` so that two ranks can share the one GPU of this box:
    one JSON line from rank 0, the contract's keys, whole-job queries (slab: one grid; replicas: one grid per rank)."""
    import json
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--resolution", "64", "--mode", mode, "--backend", "gloo"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["unit"] == "queries/s" and d["value"] > 0
    assert d["scaling"] == ("strong" if mode == "slab" else "weak")
    assert d["config"]["queries_per_step"] == 64 ** 3 * (1 if mode == "slab" else 2)
    assert d["config"]["parallelism"] == mode + "2" and d["config"]["mesh"]["verts_hr"] > 0
    assert abs(d["value"] - d["config"]["queries_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert "cpu_baseline" not in d and d["roofline"]["avg_launch_ms"] > 0
