"""BASELINE configs[3]'s path on one GPU: dist.reconstruction_sharded with 2 and 3 processes (ragged slabs) against the
single-process reconstruction - meshes bit-identical, both precisions, one-piece and streamed per-slab extraction
(tools/gpu_slab_check.py; gloo with host staging, since RCCL refuses two ranks on one device)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.parametrize("world,R", [(2, 40), (3, 50), (8, 76), (2, 512), (4, 512), (8, 512)])
def test_sharded_reconstruction_equals_single_process(world, R):
    """(world, 512): BASELINE's full grid, 2 and 4 ranks sharing the one GPU - the slab kernels, the halo / counts / boundary-id
    exchange and the shared-memory mesh delivery at the size configs[3] runs them; (8, 512): configs[3]'s own rank count, as eight
    processes on the one GPU (no 8-GPU node has been available); (8, 76): eight ragged slabs (9 or 10 planes each).  R <= 128 also
    runs want_normals=True (the volumes gathered on rank 0)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_slab_check.py"), str(world), str(R)], capture_output=True,
                       text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.count("slab == one piece") == (5 if R <= 128 else 4) and "MISMATCH" not in r.stdout, r.stdout
    assert "flat field raises on every rank" in r.stdout and "no error on a flat field" not in r.stdout, r.stdout


@pytest.mark.parametrize("world,R,H", [(2, 64, 256), (4, 128, 512), (8, 128, 512)])
def test_sharded_encoder_is_bit_identical(world, R, H):
    """dist.encode_sharded: the super-resolution net on each rank's image strip (recomputed 128-column halo, no exchange), the
    feature_lr strips all-gathered, filter_lr replicated, filter_hr on the strip - the feature maps every rank holds and the
    sharded meshes must equal the replicated encoder's and the single-GPU reconstruction's bit for bit."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_slab_check.py"), str(world), str(R), str(H)], capture_output=True,
                       text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.count("sharded encoder used") == world and "DIFFERENT" not in r.stdout and "fallback" not in r.stdout, r.stdout
    assert "sharded == replicated == one GPU" in r.stdout and "MISMATCH" not in r.stdout, r.stdout


def test_super_res_strip_equals_full_columns():
    """encoder.super_res_strip: every strip of an 8-way split of the 512 x 512 image's maps - interior strips (halo on both
    sides), border strips (the convolutions' own zero padding on one side) - equals the columns of the full maps bit for bit."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import common
    from surs_amd import encoder, model, weights
    from surs_amd.model import _as_img
    dev = torch.device("cuda:0")
    net = model.SuRSNet(common.opt()).to(device=dev)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in common.state_dict().items()})
    net.eval()
    img = torch.from_numpy(weights.synthetic_image(512, seed=1)).to(dev)
    W = net._encoder_weights()
    x = _as_img(img)
    full = encoder.super_res(W, x)
    hwc = lambda t: t.buf.view(t.h, t.w, t.c)
    for a, b in ((0, 32), (32, 64), (94, 128), (126, 160), (222, 256), (0, 256)):
        got = encoder.super_res_strip(W, x, a, b)
        for g, f, sc in zip(got, full, (4, 1, 4)):
            assert (g.h, g.w, g.c) == (f.h, sc * (b - a), f.c)
            assert torch.equal(hwc(g), hwc(f)[:, sc * a:sc * b, :]), (a, b, sc)


@pytest.mark.parametrize("mode", ["default", "slab", "replicas"])
def test_bench_multi_rank_paths_on_one_gpu(mode):
    """bench.py's N > 1 code paths, launched as the driver launches them (torch.distributed.run, one process per rank, `--gpus N` and
    nothing else - mode "default") but with `--backend gloo` so that two ranks can share the one GPU of this box: ONE JSON line from
    rank 0 with the contract's keys.  The default launch measures BOTH configurations north_star names: the headline is the slab run
    (BASELINE configs[3]: strong scaling, one grid for the whole job, sharded encoder, --precision bf16) and config.replicas holds the
    one-subject-per-GPU leg in fp16 (configs[4]: weak scaling, one grid per rank) with its own step time and stage times.  `--mode slab`
    / `--mode replicas` run one leg as the headline; a replicas run that is not fp16 must not call itself configs[4]."""
    import json
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--resolution", "64", "--backend", "gloo"] + ([] if mode == "default" else ["--mode", mode])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    head = "replicas" if mode == "replicas" else "slab"
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["unit"] == "queries/s" and d["value"] > 0
    assert d["scaling"] == ("strong" if head == "slab" else "weak")
    assert d["config"]["queries_per_step"] == 64 ** 3 * (1 if head == "slab" else 2)
    assert d["config"]["parallelism"] == head + "2" and d["config"]["mesh"]["verts_hr"] > 0
    assert abs(d["value"] - d["config"]["queries_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert "cpu_baseline" not in d and d["roofline"]["avg_launch_ms"] > 0
    assert d["config"]["sharded_encoder"] is (head == "slab") and d["config"]["encoder_precision"].startswith("fp32-grade")
    w = d["config"]["workload"]
    if head == "slab":
        assert d["dtype"] == "bf16" and "configs[3]" in w and d["config"]["scaling_target_read_from"]
    else:
        assert "configs[4] itself names fp16" in w        # (--precision bf16: the same code path, not that configuration)
    if mode == "default":
        rp = d["config"]["replicas"]
        assert rp["dtype"] == "fp16" and rp["scaling"] == "weak" and "configs[4]" in rp["workload"] and rp["sharded_encoder"] is False
        assert rp["queries_per_step"] == 2 * 64 ** 3 and rp["value"] > 0 and rp["mesh"]["verts_hr"] > 0
        assert abs(rp["value"] - rp["queries_per_step"] / (rp["ms_per_step"] * 1e-3)) / rp["value"] < 1e-6
        assert set(rp["stage_ms_rank0"]) >= {"encoder", "query", "mesh"} and rp["roofline"]["avg_launch_ms"] > 0
    else:
        assert "replicas" not in d["config"]


def test_nccl_backend_world_size_one():
    """The RCCL code path of dist.py that a single-GPU box can execute: init_process_group("nccl", device_id=...) as bench.py
    calls it, all_gather_rows on device tensors, an Exchange with an empty point-to-point batch, barrier, gather_slabs and
    reconstruction_sharded with one rank (no 8-GPU node has been available to run more)."""
    code = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(29400 + os.getpid() %% 500)
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
from surs_amd import dist as sdist
rows = sdist.all_gather_rows([1.0, 2.5, float('nan')], dev)
assert rows.shape == (1, 3) and rows[0, 1] == 2.5 and np.isnan(rows[0, 2])
sdist.Exchange().start().wait()
t = torch.arange(8 * 4 * 4, dtype=torch.float32, device=dev).reshape(8, 4, 4)
assert sdist.gather_slabs(t, 8) is t
dist.barrier()
x = torch.ones(4, device=dev); dist.all_reduce(x); assert float(x.sum()) == 4.0
sub = dist.new_group([0])
assert sdist.all_gather_rows([3.0], dev, sub)[0, 0] == 3.0 and sdist._global_rank(sub, 0) == 0
dist.destroy_process_group()
print('nccl world 1 ok')
""" % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "nccl world 1 ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
