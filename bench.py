"""Headline benchmark: one dense 512^3 reconstruction of one subject (BASELINE.json configs[2] at N=1,
configs[3] at N>1) = image encoder -> 134 217 728 occupancy queries (bf16 MFMA column kernel) -> 2x Lewiner
marching cubes -> meshes on the host.  A "step" is one such reconstruction on a synthetic 512x512 image with
seeded random-init weights; `value` is queries per second over the whole job.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

N > 1, default `--mode replicas` (BASELINE configs[4], weak scaling): one subject per GPU (image seeds 1..N), every rank
runs the whole reconstruction, no data-path collective - subjects are independent, which is how a batch of inputs
is served; `value` = N x 512^3 queries per max-over-ranks step time.
`--mode slab` (BASELINE configs[3], strong scaling of ONE subject): the grid is split into contiguous x-slabs, one per
rank, every rank runs the (small) encoder redundantly, the occupancy slabs are gathered to rank 0 over RCCL (the one real
exchange step of the path), rank 0 extracts the meshes.  Rank 0 prints ONE JSON line.  The CPU baseline leg (rank 0, N=1 only) times the oracle - test infrastructure, the
checker, never the thing measured - on a bounded sample of the same grid.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_QUERY = 4564998        # 2 x 2 282 499 MAC, both classifiers (SURVEY.md 8d / BASELINE.md section 3)
PEAK_MFMA = {"bf16": 2.5e15, "fp16": 2.5e15}   # dense, MI355X_MICROARCH.md
RES = 512
IMG = 512


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--resolution", type=int, default=RES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="replicas", choices=["replicas", "slab"],
                    help="N>1 only: one subject per GPU (weak) or one subject split into x-slabs + RCCL gather (strong)")
    ap.add_argument("--image", default="noise", choices=["smooth", "noise"],
                    help="synthetic input: white noise x mask (SURVEY 8d, default) or band-limited; with random-init weights both give a noise-like field")
    args = ap.parse_args()

    import torch.distributed as dist
    from surs_amd import _lib, dist as sdist, mesh_util, model, options, train_util, weights
    import ctypes as C

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" %
                         (args.gpus, world, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    R = args.resolution
    opt = options.BaseOptions().parse(["--loadSize", "1024", "--residual", "--b_min", "-0.5", "-0.5", "-0.5", "--b_max", "0.5",
                                       "0.5", "0.5", "--resolution", str(R), "--precision", args.precision])
    sd = weights.synthetic_state_dict(opt, seed=0)
    net = model.SuRSNet(opt).to(device=dev)
    net.load_state_dict(sd)
    net.eval()
    slab = world > 1 and args.mode == "slab"
    make_image = weights.smooth_image if args.image == "smooth" else weights.synthetic_image
    image = torch.from_numpy(make_image(IMG, seed=1 if slab else 1 + rank)).to(dev)
    calib = train_util.gen_calib().to(dev)
    b_min, b_max = np.array([-0.5] * 3), np.array([0.5] * 3)
    lib = _lib.lib()

    stage_ms = {"encoder": 0.0, "query": 0.0, "gather": 0.0, "mesh": 0.0}
    last = {}
    streamed = [False]

    def step(timed):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        _, f_lr, f_hr = net.super_res(image)
        net.filter_hr(f_hr)
        net.filter_lr(f_lr)
        ev[1].record()
        m = None
        if not slab:
            # the product path of reconstruction(): marching cubes and the mesh copies pipelined into the sweep (from the
            # second reconstruction on: the first one sizes the mesh buffers); ev[2] = end of the sweep's last launch
            m = mesh_util.reconstruction_streamed(opt, net, calib, R, b_min, b_max, None, want_normals=False, timing=ev[2])
            streamed[0] = m is not None
            if m is not None:
                ev[3].record()
        if m is None:
            i0, i1 = sdist.slab_range(R, rank, world) if slab else (0, R)
            vh, vl, mat = mesh_util.eval_volumes(opt, net, calib, R, b_min, b_max, None, i0, i1)
            ev[2].record()
            full_hr = sdist.gather_slabs(vh, R, 0) if slab else vh
            full_lr = sdist.gather_slabs(vl, R, 0) if slab else vl
            ev[3].record()
            if rank == 0 or not slab:
                # gen_mesh keeps vertices and faces only (lib/train_util.py:72)
                m = mesh_util.meshes_from_volumes(net, [full_hr, full_lr], mat, want_normals=False)
        if m is not None:
            last["verts_hr"], last["faces_hr"], last["verts_lr"], last["faces_lr"] = len(m[0]), len(m[1]), len(m[4]), len(m[5])
        ev[4].record()
        if timed:
            torch.cuda.synchronize()
            # streamed path: "query" = the sweep (the marching cubes of finished layers run beside it on their own streams),
            # "mesh" = what is left after the last launch (last layers, last copies); the slab path also has "gather"
            names = ("encoder", "query", "mesh", "gather") if streamed[0] else ("encoder", "query", "gather", "mesh")
            for k, (a, b) in zip(names, zip(ev[:-1], ev[1:])):
                stage_ms[k] += a.elapsed_time(b)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    lib.surs_profile_enable(1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    dt = time.perf_counter() - t0
    launches, kms, kpts = C.c_double(0), C.c_double(0), C.c_double(0)
    lib.surs_profile_read(C.byref(launches), C.byref(kms), C.byref(kpts))
    lib.surs_profile_enable(0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        queries = float(R) ** 3 * (1 if (slab or world == 1) else world)   # replicas: one full grid per rank
        ms_per_step = dt / args.steps * 1e3
        value = queries * args.steps / dt
        # dominant kernel: grid_mlp_kernel on this rank, HIP events around every launch
        k_avg_ms = kms.value / max(launches.value, 1.0)
        k_pts_per_launch = kpts.value / max(launches.value, 1.0)
        achieved = k_pts_per_launch * FLOP_PER_QUERY / (k_avg_ms * 1e-3) / 1e12 if k_avg_ms > 0 else 0.0
        peak = PEAK_MFMA[args.precision] / 1e12
        traffic = mfma_busy = None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc):   # counters are collected in their own rocprofv3 --pmc passes (tools/profile_round.sh)
            try:
                summary = json.load(open(pmc))
                traffic = summary.get("grid_mlp_kernel_hbm_bytes_per_launch")
                mfma_busy = summary.get("mfma_busy_fraction")
            except Exception:
                traffic = mfma_busy = None
        out = {
            "metric": "occupancy queries/sec (dense %d^3 reconstruction: encoder + query sweep + 2x marching cubes)" % R,
            "value": value, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if slab else "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]: %s 512x512 synthetic image%s, %d^3 grid each, bf16 MFMA classifier cores, "
                                   "HIP marching cubes x2 pipelined into the sweep%s" % (2 if world == 1 else (3 if slab else 4),
                                                                 "one" if (slab or world == 1) else str(world),
                                                                 "" if (slab or world == 1) else "s (one subject per GPU, replicas)", R,
                                                                 ", x-slab per rank + RCCL gather" if slab else ""),
                       "resolution": R, "image": IMG, "image_kind": args.image, "queries_per_step": int(queries),
                       "reconstruction_s": ms_per_step / 1e3,
                       "stage_ms_rank0": {k: v / args.steps for k, v in stage_ms.items()},
                       "mesh": dict(last), "parallelism": ("slab%d" % world) if slab else ("replicas%d" % world)},
            "roofline": {"kernel": "grid_mlp_kernel_v%s<%s>" % (os.environ.get("SURS_GRID_KERNEL", "3")[:1], args.precision), "bound": "mfma", "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "avg_launch_ms": k_avg_ms, "queries_per_launch": k_pts_per_launch,
                         "flop_per_query_algorithmic": FLOP_PER_QUERY, "flop_per_query_executed": 2752512,
                         "mfma_busy_fraction_pmc": mfma_busy},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(net, sd, R, b_min, b_max)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(net, sd, R, b_min, b_max):
    """The oracle (a CPU port of the reference path, all host threads) on a bounded sample: one x-plane of the grid."""
    import oracle
    fl = net.im_feat_list_lr[-1][0].cpu().numpy()
    fh = net.im_feat_list_hr[0][0].cpu().numpy()
    n = min(R * R, 262144)
    start = (R // 2) * R * R
    pts = oracle.grid_points(R, b_min, b_max, start, start + n)
    calib = np.diag([2.0, -2.0, 2.0, 1.0]).astype(np.float32)
    oracle.query(sd, pts[:, :4096], calib, fl, fh, 1024, 200.0)   # warm up threads / caches
    t = time.perf_counter()
    oracle.query(sd, pts, calib, fl, fh, 1024, 200.0)
    dt = time.perf_counter() - t
    return {"value": n / dt, "unit": "queries/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": "%d grid points (the x-plane i=%d of the %d^3 grid) through oracle.query, fp32, OpenMP" % (n, R // 2, R),
            "seconds": dt}


if __name__ == "__main__":
    main()
