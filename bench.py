"""Headline benchmark: one dense 512^3 reconstruction of one subject (BASELINE.json configs[2] at N=1,
configs[3] / configs[4] at N>1) = image encoder -> 134 217 728 occupancy queries (column kernel on the matrix cores) ->
2x Lewiner marching cubes -> meshes on the host.  A "step" is one such reconstruction on a synthetic 512x512 image with
seeded random-init weights; `value` is queries per second over the whole job.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`--precision bf16` (default: what BASELINE configs[2] names) | `fp16` (configs[4]) | `fp32` (the parity-grade sweep:
column kernel v11, split-f16 operands, logits within 1e-4 of the reference).  At N=1 the line also carries
`config.fp32_mode` - the same step timed in fp32 - and `config.precision_acceptance` - what the reduced precisions do to
the field and the meshes at full size against the fp32 sweep (tools/precision_report.py; asserted in
tests/test_gpu_precision.py).

N > 1, default `--mode both`: BOTH configurations north_star names, in one invocation and one JSON line.
  * the headline = `slab` (BASELINE configs[3], STRONG scaling of ONE subject - the number the ">= 6x at 8 GPUs" target is read
    from): the grid is split into contiguous x-slabs, one per rank; super_res runs on the rank's image strip, feature_lr is
    all-gathered, every rank extracts the mesh of its own slab while it sweeps (marching cubes sharded with a one-plane halo); the
    ranks exchange the halo planes, the per-rank vertex / face counts and the vertex ids of the boundary planes, and rank 0
    receives meshes, not volumes (dist.encode_sharded, dist.reconstruction_sharded).  `value` = 512^3 queries per max-over-ranks
    step time, `scaling` = "strong", classifier precision = --precision (bf16).
  * `config.replicas` = BASELINE configs[4] (weak scaling): one subject per GPU (image seeds 1..N), classifiers in FP16 as that
    config names, every rank runs the whole reconstruction, no data-path collective; its own value (N x 512^3 per max-over-ranks
    step), ms_per_step and stage_ms.
`--mode slab` / `--mode replicas` run one of them as the headline (replicas then in --precision; the workload string says
configs[4] only when that is fp16).  Rank 0 prints ONE JSON line.  The CPU baseline leg (rank 0, N=1 only) times the oracle - test infrastructure, the
checker, never the thing measured - on a bounded sample of the same grid.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_QUERY = 4564998        # 2 x 2 282 499 MAC, both classifiers, as the reference computes them
FLOP_EXECUTED = 2752512         # what the column kernels run through the matrix pipe per query and per product (A.4)
PRODUCTS = {"bf16": 1, "fp16": 1, "fp32": 3}   # MFMA products per MAC (fp32: hi*hi + hi*lo + lo*hi on f16 parts)
PEAK_MFMA = 2.5e15              # dense bf16 / f16, MI355X_MICROARCH.md
RES = 512
IMG = 512
REF_LOOP_PASSES = 7   # timed passes of the reference-loop leg (40 chunks each): median reported, min / max beside it


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--resolution", type=int, default=RES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip config.fp32_mode and config.precision_acceptance (N=1)")
    ap.add_argument("--mode", default="both", choices=["both", "replicas", "slab"],
                    help="N>1 only: slab = one subject split into x-slabs (strong scaling, configs[3]); replicas = one subject per GPU "
                         "(weak, configs[4]); both (default) = slab as the headline + the fp16 replicas leg in config.replicas")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="N>1: nccl = RCCL, one rank per GPU (the measured configuration); gloo = host-staged exchange, ranks may share "
                         "a GPU (LOCAL_RANK modulo the device count) - lets the N>1 code paths run on a single-GPU box (tests)")
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
    if world > 1:   # (before the first HIP call: the runtime reads it when it initialises)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.backend == "gloo":
        local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        sdist.init_host_group()   # (the host-side row exchanges of slab mode: created now, collectively, not inside the first reconstruction)

    R = args.resolution
    flags = ["--loadSize", "1024", "--residual", "--b_min", "-0.5", "-0.5", "-0.5", "--b_max", "0.5", "0.5", "0.5",
             "--resolution", str(R)]
    opt = options.BaseOptions().parse(flags + ["--precision", args.precision])
    sd = weights.synthetic_state_dict(opt, seed=0)
    net = model.SuRSNet(opt).to(device=dev)
    net.load_state_dict(sd)
    net.eval()
    slab = world > 1 and args.mode in ("slab", "both")      # the headline leg of this run
    make_image = weights.smooth_image if args.image == "smooth" else weights.synthetic_image
    image_one = torch.from_numpy(make_image(IMG, seed=1)).to(dev)                 # slab: every rank the same subject
    image_own = torch.from_numpy(make_image(IMG, seed=1 + rank)).to(dev)          # replicas: one subject per rank
    calib = train_util.gen_calib().to(dev)
    b_min, b_max = np.array([-0.5] * 3), np.array([0.5] * 3)
    lib = _lib.lib()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(o, steps, warmup, net=net, slab=slab):
        """warmup untimed steps, then `steps` timed ones between barriers; -> (seconds, per-stage ms, mesh sizes, kernel timing)."""
        stage_ms = {"encoder": 0.0, "query": 0.0, "exchange": 0.0, "mesh": 0.0}
        last = {}
        image = image_one if (slab or world == 1) else image_own

        def step(timed):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
            if slab:   # super_res on this rank's image strip, feature_lr all-gathered, filter_hr on the strip (dist.encode_sharded)
                sdist.encode_sharded(net, image, calib, R, b_min, b_max)
            else:
                _, f_lr, f_hr = net.super_res(image)
                net.filter_hr(f_hr)
                net.filter_lr(f_lr)
            ev[1].record()
            if slab:
                m = sdist.reconstruction_sharded(o, net, calib, R, b_min, b_max, want_normals=False, timing=ev[2], copy_out=False)
            else:
                # the product path of reconstruction(): marching cubes and the mesh copies pipelined into the sweep (from
                # the second reconstruction on: the first one sizes the mesh buffers); ev[2] = end of the sweep's last launch
                m = mesh_util.reconstruction_streamed(o, net, calib, R, b_min, b_max, None, want_normals=False, timing=ev[2])
                if m is None:
                    vh, vl, mat = mesh_util.eval_volumes(o, net, calib, R, b_min, b_max, None)
                    ev[2].record()
                    m = mesh_util.meshes_from_volumes(net, [vh, vl], mat, want_normals=False)   # gen_mesh keeps vertices and faces only
            if m is not None:
                last["verts_hr"], last["faces_hr"], last["verts_lr"], last["faces_lr"] = len(m[0]), len(m[1]), len(m[4]), len(m[5])
            ev[3].record()
            if timed:
                torch.cuda.synchronize()
                # "query" = the sweep (the marching cubes of finished layers run beside it on their own streams), "mesh" = what is
                # left after the last launch (last layers, last copies; in slab mode also the exchange of counts / ids / meshes)
                for k, (a, b) in zip(("encoder", "query", "mesh"), zip(ev[:-1], ev[1:])):
                    stage_ms[k] += a.elapsed_time(b)

        for _ in range(warmup):
            step(False)
        lib.surs_profile_enable(1)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(True)
        barrier()
        dt = time.perf_counter() - t0
        launches, kms, kpts, ktiles, ksteps = C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0)
        lib.surs_profile_read_ksteps(C.byref(ktiles), C.byref(ksteps))   # (column kernels v10 / v11: data-dependent layer-1 k-steps)
        lib.surs_profile_read(C.byref(launches), C.byref(kms), C.byref(kpts))
        lib.surs_profile_enable(0)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        k_avg_ms = kms.value / max(launches.value, 1.0)
        kavg = ksteps.value / ktiles.value if ktiles.value > 0 else None
        last["column_kernel"] = getattr(net._workspace(), "kernel_choice", (0, None))   # (version the probe picked, listed channels per tile)
        return dt, {k: v / steps for k, v in stage_ms.items()}, dict(last), (k_avg_ms, kpts.value / max(launches.value, 1.0), kavg)

    def roofline(prec, k_avg_ms, k_pts, kavg=None, kver=0):
        """The column kernel of this precision, timed with HIP events around every launch on its launch stream."""
        nprod = PRODUCTS[prec]
        kver = kver or int(os.environ.get("SURS_GRID_F32_KERNEL" if prec == "fp32" else "SURS_GRID_KERNEL", "11" if prec == "fp32" else "12"))
        alg = k_pts * FLOP_PER_QUERY / (k_avg_ms * 1e-3) / 1e12 if k_avg_ms > 0 else 0.0
        # what the matrix pipe executes per query: the dense cores of layers 1-3 (column-constant reduction, A.4); with column
        # kernels v10 / v11: layer 1 is one affine k-step + the measured residual k-steps (16 channels x 512 rows each) instead of 64
        # (the affine k-step is one product per MAC also in the fp32-grade kernel)
        flop_exec = FLOP_EXECUTED * nprod if kavg is None else 2 * (2 * (512 * 256 + 256 * 128) * nprod + (kavg * nprod + 1.0) * 16 * 512 * 2)
        exe = k_pts * flop_exec / (k_avg_ms * 1e-3) / 1e12 if k_avg_ms > 0 else 0.0
        peak = PEAK_MFMA / 1e12
        r = {"kernel": "grid_mlp_kernel_v%d (split-f16, 3 products per MAC)" % kver if prec == "fp32" else
                       "grid_mlp_kernel_v%d<%s>%s" % (kver, prec, " (+ grid_mlp_kernel_v10 on the tiles it hands over: inside the timed launches)" if kver == 12 else ""),
             "bound": "mfma", "unit": "TFLOP/s",
             # contract fields = what the matrix pipe EXECUTES per launch / the launch's duration against the dense f16 / bf16 MFMA
             # peak: a utilisation.  (The column kernels restate layer 1 exactly - DESIGN 4.1c - and execute far fewer FLOP than
             # the reference's dense products; the algorithmic rate is kept below as a labelled extra and is NOT a utilisation.)
             "achieved": exe, "peak": peak, "frac": exe / peak,
             "achieved_executed": exe, "frac_executed": exe / peak, "mfma_products_per_mac": nprod,
             "achieved_algorithmic": alg, "frac_algorithmic": alg / (peak / nprod),
             "frac_algorithmic_note": "reference FLOP (4 564 998 per query, SURVEY 8d) / time / (peak / products per MAC); restated layer 1: "
                                      "not a utilisation, can exceed 1",
             "avg_launch_ms": k_avg_ms, "queries_per_launch": k_pts,
             "flop_per_query_algorithmic": FLOP_PER_QUERY, "flop_per_query_executed": flop_exec,
             "traffic": None}
        if kavg is not None:
            r["layer1_residual_ksteps_per_tile"] = kavg
            r["layer1_listed_channels_per_tile_upper"] = 16.0 * kavg
        # HBM bytes / MFMA-busy cycles come from separate rocprofv3 --pmc passes (tools/profile_round.sh) and are only valid
        # for the library they were collected on: they are reported when the sha256 of libsurs_hip.so matches, else null
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc):
            try:
                summary = json.load(open(pmc))
                entry = summary.get("kernels", {}).get(prec)
                same = summary.get("lib_sha256") == hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()
                r["pmc"] = {"source": "profiles/pmc_summary.json (separate rocprofv3 --pmc passes, not this run)",
                            "lib_sha256_profiled": summary.get("lib_sha256"), "matches_this_library": same}
                if entry and same:
                    # per launch like `achieved`: the counters were collected on full batches (entry["queries_per_launch"]); the
                    # streamed reconstruction ends on a few shorter launches, so its average launch is smaller
                    scale = k_pts / float(entry["queries_per_launch"]) if entry.get("queries_per_launch") else 1.0
                    r["traffic"] = entry.get("hbm_bytes_per_launch") * scale if entry.get("hbm_bytes_per_launch") is not None else None
                    r["mfma_busy_fraction_pmc"] = entry.get("mfma_busy_fraction")
            except Exception:
                pass
        return r

    dt, stage_ms, last, (k_avg_ms, k_pts, k_ks) = run(opt, args.steps, args.warmup)
    kver_head = int(last.pop("column_kernel", (0, None))[0] or 0)

    replicas = None
    if world > 1 and args.mode == "both":
        # BASELINE configs[4]: one subject per GPU, classifiers in fp16 (a network object of its own: the blob is packed per precision)
        o16 = options.BaseOptions().parse(flags + ["--precision", "fp16"])
        net16 = model.SuRSNet(o16).to(device=dev)
        net16.load_state_dict(sd)
        net16.eval()
        d16, st16, last16, (k16, p16, ks16) = run(o16, args.steps, args.warmup, net=net16, slab=False)
        kv16 = int(last16.pop("column_kernel", (0, None))[0] or 0)
        replicas = {"workload": "BASELINE configs[4]: %d subjects (one per GPU), %d^3 grid each, loadSize 1024 (SR path), fp16 classifier cores "
                                "on MFMA, fp32-grade encoder, no data-path collective" % (world, R),
                    "dtype": "fp16", "scaling": "weak", "value": float(R) ** 3 * world * args.steps / d16, "unit": "queries/s",
                    "ms_per_step": d16 / args.steps * 1e3, "steps": args.steps, "warmup": args.warmup, "queries_per_step": int(float(R) ** 3 * world),
                    "stage_ms_rank0": st16, "mesh": last16, "encoder_precision": "fp32-grade (two f16 parts)", "sharded_encoder": False,
                    "roofline": roofline("fp16", k16, p16, ks16, kv16)}
        del net16

    extras = {}
    if world == 1 and not args.no_extras and args.precision != "fp32":
        # the same step in the parity-grade precision: fp32-grade column kernel AND fp32-grade encoder (a network object of its own:
        # --precision bf16 runs the encoder's 3x3 convolutions on one f16 product per MAC)
        o32 = options.BaseOptions().parse(flags + ["--precision", "fp32"])
        net32 = model.SuRSNet(o32).to(device=dev)
        net32.load_state_dict(sd)
        net32.eval()
        d32, st32, last32, (k32, p32, ks32) = run(o32, 2, 2, net=net32)   # (a new object: one reconstruction sizes the mesh buffers, one warms the streamed path)
        del net32
        kv32 = int(last32.pop("column_kernel", (0, None))[0] or 0)
        extras["fp32_ms_per_step"] = d32 / 2 * 1e3      # (the parity-mode step, also at the top level of config)
        extras["fp32_mode"] = {"dtype": "fp32", "value": float(R) ** 3 * 2 / d32, "unit": "queries/s", "ms_per_step": d32 / 2 * 1e3,
                               "steps": 2, "warmup": 2, "stage_ms": st32, "mesh": last32, "roofline": roofline("fp32", k32, p32, ks32, kv32),
                               "tolerance": "logits within 1e-4 of the reference's fp32 path (tests/test_gpu_query.py, test_gpu_model.py)"}
    if world == 1 and not args.no_extras:
        # The restated column kernels' cost depends on the weights and features (how many layer-0 channels change LeakyReLU branch
        # inside a z tile).  (1) dense_floor: the same step on the dense-layer-1 kernels (v3 / v5) - the data-independent guarantee,
        # and what the host falls back to above 400 listed channels per tile.  (2) listed_sensitivity: the sweep alone with layer 0's
        # depth column (conv0.weight[:, 320]) scaled by 1, 4, 16, 60 - listed channels per tile, time on the restated kernel, and
        # what native.grid_kernel_for picks.
        try:
            lib.surs_set_grid_kernel(5 if args.precision == "fp32" else 3)
            os.environ["SURS_GRID_AUTO"] = "0"      # (the host's per-sweep choice - a per-call option - would override the process setting)
            try:
                dfl, stf, lastf, (kf, pf, _) = run(opt, 2, 1)
            finally:
                os.environ.pop("SURS_GRID_AUTO", None)
            lib.surs_set_grid_kernel(0)
            extras["dense_floor"] = {"kernel": "grid_mlp_kernel_v5" if args.precision == "fp32" else "grid_mlp_kernel_v3<%s>" % args.precision,
                                     "value": float(R) ** 3 * 2 / dfl, "unit": "queries/s", "ms_per_step": dfl / 2 * 1e3, "stage_ms": stf,
                                     "avg_launch_ms": kf, "note": "dense layer 1: independent of weights and features"}
            from surs_amd import native
            fl, fh = net.features()
            ws = net._workspace()
            from surs_amd import sdf
            mat = sdf.create_grid(R, R, R, b_min, b_max)[1][:3].reshape(-1)
            cal = calib[0].cpu().numpy().reshape(-1)[:12]
            zmul, zdiv = net._zscale()
            vh = torch.empty((R, R, R), dtype=torch.float32, device=dev)
            vl = torch.empty_like(vh)
            rows = []
            for gain in (1.0, 4.0, 16.0, 60.0):
                sdg = {k: np.array(v, copy=True) for k, v in sd.items() if k.startswith("mlp_")}
                for mm in ("mlp_lr.", "mlp_hr."):
                    sdg[mm + "conv0.weight"][:, 320] *= gain
                blob, _ = native.pack_mlp(sdg, args.precision, dev)
                tile = 64 if args.precision == "fp32" else 128
                lr, hr = native.probe_listed(R // 2, R, R, tile, mat, cal, zmul, zdiv, fl, fh, blob, ws)
                pick = native.grid_kernel_for(R, R, R, mat, cal, zmul, zdiv, fl, fh, blob, args.precision, ws)
                row = {"gain": gain, "listed_lr_per_tile": lr, "listed_hr_per_tile_upper": hr,
                       "host_picks": pick or (11 if args.precision == "fp32" else 12)}
                kerns = (("restated_ms", 11), ("dense_ms", 5)) if args.precision == "fp32" else (("streamed_v12_ms", 12), ("restated_ms", 10), ("dense_ms", 3))
                for name, kern in kerns:
                    native.query_grid(0, 32, R, R, mat, cal, zmul, zdiv, fl, fh, blob, args.precision, ws, vh[:32], vl[:32], kernel=kern)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    native.query_grid(0, R, R, R, mat, cal, zmul, zdiv, fl, fh, blob, args.precision, ws, vh, vl, kernel=kern)
                    torch.cuda.synchronize()
                    row[name] = (time.perf_counter() - t0) * 1e3
                rows.append(row)
            del vh, vl
            extras["listed_sensitivity"] = {"what": "%d^3 sweep alone (no encoder, no marching cubes), layer-0 depth column x gain" % R,
                                            "dense_threshold_listed": native.LISTED_DENSE_THRESHOLDS[args.precision], "rows": rows}
        except Exception as e:
            lib.surs_set_grid_kernel(0)
            extras["dense_floor"] = {"error": repr(e)}
    if world == 1 and not args.no_extras:
        # a run of subjects (apps/eval_SuRS.py:74-80) - per subject: decoded 8-bit pixels -> img_LR -> encoder ->
        # reconstruction (no OBJ files).  sequential = host normalise + upload + gen_mesh's order; pipelined = device input stage,
        # next subject's decode / upload / encoder under the current sweep (train_util.gen_mesh_pipelined)
        try:
            from surs_amd import data
            K = 4
            ds = data.SyntheticDataset(opt, n=K, size=IMG)
            raws = [ds.get_raw_item(i) for i in range(K)]

            def sequential():
                for raw in raws:
                    m = raw["mask"].astype(np.float32) / np.float32(255.0)
                    x = (raw["rgb"].astype(np.float32) / np.float32(255.0) - np.float32(0.5)) / np.float32(0.5)
                    img = torch.from_numpy(np.ascontiguousarray((m[None] * x.transpose(2, 0, 1))[None])).to(dev)
                    _, f_lr, f_hr = net.super_res(img)
                    net.filter_hr(f_hr)
                    net.filter_lr(f_lr)
                    mesh_util.reconstruction(opt, net, dev, calib, R, b_min, b_max, use_octree=False, want_normals=False)

            class Raws:
                def get_raw_item(self, i):
                    return raws[i]                # decoded pixels held in memory on both sides

            def pipelined():
                train_util.gen_mesh_pipelined(opt, net, dev, Raws(), range(K), None, use_octree=False, write=False)

            times = {}
            for name, fn in (("sequential", sequential), ("pipelined", pipelined), ("sequential", sequential), ("pipelined", pipelined)):
                fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                times[name] = min(times.get(name, 1e9), time.perf_counter() - t0)
            extras["subject_pipeline"] = {"subjects": K, "resolution": R, "precision": args.precision,
                                          "sequential_subjects_per_s": K / times["sequential"],
                                          "pipelined_subjects_per_s": K / times["pipelined"],
                                          "note": "decoded 8-bit pixels in host memory -> meshes in host memory, no OBJ files; the sweep keeps "
                                                  "the GPU busy in both, the pipeline hides the input stage and the host gaps"}
        except Exception as e:
            extras["subject_pipeline"] = {"error": repr(e)}
    if world == 1 and not args.no_extras:
        # (a) "drops in unchanged": the reference's own sweep loop (lib/sdf.py:32-45 batch_eval over lib/mesh_util.py:20-28 eval_func)
        # around the facade - 50 000 points per call, query_mr + query_sr + get_preds()[0][0].detach().cpu().numpy() - timed on
        # 40 chunks of a 512^3 grid's points; (b) gen_mesh's DEFAULT sweep (use_octree=True, lib/train_util.py:53,73) at 512^3 in the
        # parity precision on the smooth body field (the noise field leaves nothing to skip)
        try:
            from surs_amd import sdf
            o32 = options.BaseOptions().parse(flags + ["--precision", "fp32"])
            n32 = model.SuRSNet(o32).to(device=dev)
            n32.load_state_dict(sd)
            n32.eval()
            n32.im_feat_list_lr, n32.im_feat_list_hr = net.im_feat_list_lr, net.im_feat_list_hr
            ns = 50000
            nchunks = min(40, max(1, R ** 3 // ns))
            # (grid points as create_grid lays them out - float64 matmul on the voxel indices - from the middle of the grid; the 3.2 GB
            #  coordinate array of a 512^3 grid is not materialised for 40 chunks)
            mat4 = sdf.create_grid(R, R, R, b_min, b_max)[1]
            idx = np.arange(nchunks * ns, dtype=np.int64) + max(0, (R // 2) * R * R - nchunks * ns // 2)
            ijk = np.stack([idx // (R * R), (idx // R) % R, idx % R]).astype(np.float64)
            pts_all = np.matmul(mat4[:3, :3], ijk) + mat4[:3, 3:4]

            def eval_func(nn, points):   # lib/mesh_util.py:20-28 (np.repeat is what makes the strided chunk view contiguous)
                points = np.expand_dims(points, axis=0)
                points = np.repeat(points, nn.num_views, axis=0)
                samples = torch.from_numpy(points).to(device=dev).float()
                nn.query_mr(samples, calib)
                nn.query_sr(samples, calib)
                return nn.get_preds()[0][0].detach().cpu().numpy(), nn.get_preds()[1][0].detach().cpu().numpy()

            out_hr, out_lr = np.zeros(pts_all.shape[1]), np.zeros(pts_all.shape[1])

            def loop(nn):                # lib/sdf.py:32-45
                for i in range(pts_all.shape[1] // ns):
                    out_hr[i * ns:(i + 1) * ns], out_lr[i * ns:(i + 1) * ns] = eval_func(nn, pts_all[:, i * ns:(i + 1) * ns])

            for key, nn, what in (("reference_loop", n32, "fp32 (the facade's default; the chunks are runs of grid points with one image position each: "
                                                           "surs_query_points_columns, the fp32-grade column kernel v11 in tile mode)"),
                                  ("reference_loop_reduced", net, "--precision %s (the same runs on the 16-bit column kernel v10)" % args.precision)):
                if key == "reference_loop_reduced" and args.precision == "fp32":
                    continue
                loop(nn)                 # warm-up pass (kernel attributes, workspace, clocks)
                torch.cuda.synchronize()
                passes = []
                for _ in range(REF_LOOP_PASSES):
                    t0 = time.perf_counter()
                    loop(nn)
                    torch.cuda.synchronize()
                    passes.append(time.perf_counter() - t0)
                nch = pts_all.shape[1] // ns
                tl = float(np.median(passes))
                extras[key] = {"what": "the reference's eval_grid loop (50 000-point chunks, host numpy in / out, lib/sdf.py:32-45) "
                                       "around SuRSNet.query_mr / query_sr / get_preds, " + what,
                               "points": int(nch * ns), "passes": REF_LOOP_PASSES, "seconds": tl, "value": nch * ns / tl,
                               "unit": "queries/s", "ms_per_50k_chunk": tl / nch * 1e3,
                               "ms_per_50k_chunk_min_median_max": [min(passes) / nch * 1e3, tl / nch * 1e3, max(passes) / nch * 1e3],
                               "ms_per_50k_chunk_passes": [t / nch * 1e3 for t in passes],
                               # (the reference's own sweep of a 512^3 grid is 2 685 such calls: north_star's "< 2 s" through the
                               #  UNCHANGED loop; tests/test_gpu_parity_fullsize.py runs all of them against the product's sweep)
                               "seconds_per_512_grid_at_this_rate": tl / nch * -(-R ** 3 // ns)}
                if key == "reference_loop":
                    ref_hr = out_hr.copy()
                else:
                    extras[key]["max_abs_docc_vs_fp32_loop"] = float(np.abs(out_hr - ref_hr).max())
        except Exception as e:
            extras["reference_loop"] = {"error": repr(e)}
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import precision_report as pr
            o32 = options.BaseOptions().parse(flags + ["--precision", "fp32"])
            nb = model.SuRSNet(o32).to(device=dev)
            full = dict(sd)
            full.update(weights.body_state_dict(o32))
            nb.load_state_dict(full)
            nb.eval()
            fl_b, fh_b = weights.body_features(IMG // 2, 2 * IMG)
            feats = (pr._upload(fl_b, dev), pr._upload(fh_b, dev))
            res = {}
            for name, use_oct in (("dense", False), ("octree", True)):
                for rep in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    out = mesh_util.reconstruction(o32, nb, dev, calib, R, b_min, b_max, use_octree=use_oct, features=feats, want_normals=False)
                    torch.cuda.synchronize()
                    res[name] = {"seconds": time.perf_counter() - t0, "verts_hr": int(len(out[0])), "verts_lr": int(len(out[4]))}
            if args.precision != "fp32":
                # opt-in: the levels in the sweep's precision (--octree_precision sweep: the 16-bit column kernel on the lattice lists;
                # bounded by tests/test_gpu_octree.py::test_octree_levels_in_reduced_precision) - the default keeps them fp32-grade
                osw = options.BaseOptions().parse(flags + ["--precision", args.precision, "--octree_precision", "sweep"])
                ns_ = model.SuRSNet(osw).to(device=dev)
                ns_.load_state_dict(full)
                ns_.eval()
                for rep in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    out = mesh_util.reconstruction(osw, ns_, dev, calib, R, b_min, b_max, use_octree=True, features=feats, want_normals=False)
                    torch.cuda.synchronize()
                    res["octree_sweep_precision"] = {"precision": args.precision, "seconds": time.perf_counter() - t0,
                                                     "verts_hr": int(len(out[0])), "verts_lr": int(len(out[4]))}
            extras["octree_mode"] = {"what": "mesh_util.reconstruction(use_octree=True) - gen_mesh's default, lib/sdf.py:55-120 - against the "
                                             "dense sweep, %d^3, fp32, smooth body field (weights.body_*); the octree's output differs from "
                                             "the dense one by construction (interpolated blocks, shared-dirty artefact)" % R,
                                     "octree": res["octree"], "dense": res["dense"]}
            if "octree_sweep_precision" in res:
                extras["octree_mode"]["octree_sweep_precision_opt_in"] = res["octree_sweep_precision"]
        except Exception as e:
            extras["octree_mode"] = {"error": repr(e)}
    if world == 1 and not args.no_extras and R == RES:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import precision_report as pr
            acc = {}
            for name, inputs in (("body", pr.body_inputs(dev)), ("noise", (sd,) + tuple(net.features()))):
                rep = pr.report(inputs[0], inputs[1], inputs[2], R, dev)
                acc[name] = {p: {t: {"max_abs_dlogit": rep[p][t]["max_abs_dlogit"], "mean_abs_dlogit": rep[p][t]["mean_abs_dlogit"],
                                     "flipped_voxels": rep[p][t]["flipped_voxels"],
                                     "verts": rep[p][t]["mesh"]["verts"], "verts_fp32": rep[p][t]["mesh"]["verts_ref"],
                                     "faces": rep[p][t]["mesh"]["faces"], "faces_fp32": rep[p][t]["mesh"]["faces_ref"],
                                     "nearest_vertex_voxels": {s: {k: rep[p][t]["mesh"][s].get(k) for k in ("mean", "p999", "max", "unmatched", "n")}
                                                               for s in ("to_ref", "from_ref")}}
                                 for t in ("hr", "lr")} for p in ("bf16", "fp16")}
                acc[name]["sweep_s"] = rep["sweep_s"]
            if args.precision != "fp32" and getattr(opt, "encoder_precision", "auto") == "f16":
                # --encoder_precision f16 (opt-in) runs the encoder's 3x3 convolutions on ONE f16 product per MAC:
                # the whole reduced pipeline against the whole fp32-grade one (noise field)
                er = pr.encoder_report(dev, R, precisions=(args.precision,))
                acc["encoder_f16"] = {"im_feat_lr": er["im_feat_lr"], "im_feat_hr": er["im_feat_hr"],
                                      args.precision: {t: {k: er[args.precision][t][k] for k in ("max_abs_dlogit", "mean_abs_dlogit", "flipped_voxels")}
                                                       for t in ("hr", "lr")},
                                      "what": "f16-product encoder + %s sweep against fp32-grade encoder + fp32-grade sweep" % args.precision}
            acc["reference"] = "fp32-grade sweep (column kernel v11) on the same features and weights, 512^3"
            acc["why_bf16"] = ("BASELINE configs[2] names bf16: fp32's exponent range, no activation can overflow; fp16 (configs[4]) is "
                               "8x tighter at 0.93x the rate but saturates at 65504 - `--precision fp16` / `fp32` select the others")
            extras["precision_acceptance"] = acc
        except Exception as e:   # measurement extra: never lose the bench line over it
            extras["precision_acceptance"] = {"error": repr(e)}

    if rank == 0:
        queries = float(R) ** 3 * (1 if (slab or world == 1) else world)   # replicas: one full grid per rank
        ms_per_step = dt / args.steps * 1e3
        value = queries * args.steps / dt
        # (configs[4] names fp16: a replicas run in another precision is the same code path, not that configuration)
        cfg = "configs[2]" if world == 1 else ("configs[3]" if slab else ("configs[4]" if args.precision == "fp16" else
                                                                         "configs[4]'s replicas in %s (configs[4] itself names fp16)" % args.precision))
        enc_reduced = getattr(opt, "encoder_precision", "auto") == "f16"
        out = {
            "metric": "occupancy queries/sec (dense %d^3 reconstruction: encoder + query sweep + 2x marching cubes)" % R,
            "value": value, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if slab else "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "BASELINE %s: %s 512x512 synthetic image%s, %d^3 grid each, %s classifier cores on MFMA, %s encoder, "
                                   "HIP marching cubes x2 pipelined into the sweep%s" %
                                   (cfg, "one" if (slab or world == 1) else str(world),
                                    "" if (slab or world == 1) else "s (one subject per GPU, replicas)", R,
                                    "split-f16 (fp32-grade)" if args.precision == "fp32" else args.precision,
                                    "f16-product (reduced, opt-in)" if enc_reduced else "fp32-grade",
                                    ", x-slab per rank (super_res per image strip, feature_lr all-gathered), marching cubes per slab, meshes to rank 0" if slab else ""),
                       "encoder_precision": "f16 (one product per MAC in the 3x3 convolutions)" if enc_reduced else "fp32-grade (two f16 parts, three products per MAC)",
                       "sharded_encoder": bool(slab),
                       "scaling_target_read_from": "this line's value at N = 1, 2, 4, 8 (slab, strong scaling): north_star's >= 6x at 8 GPUs" if (slab or world == 1) else None,
                       "resolution": R, "image": IMG, "image_kind": args.image, "queries_per_step": int(queries),
                       "reconstruction_s": ms_per_step / 1e3, "stage_ms_rank0": stage_ms,
                       "mesh": last, "parallelism": ("slab%d" % world) if slab else ("replicas%d" % world),
                       "backend": None if world == 1 else args.backend},
            "roofline": roofline(args.precision, k_avg_ms, k_pts, k_ks, kver_head),
        }
        if replicas is not None:
            out["config"]["replicas"] = replicas
        out["config"].update(extras)
        out["config"].update(flat_scalars(stage_ms, extras, args.precision))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(net, sd, R, b_min, b_max)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def flat_scalars(stage_ms, extras, precision):
    """The figures a reader of the driver's parsed line needs, as FLAT scalars of config (nested objects do not survive there)."""
    out = {}
    if isinstance(stage_ms, dict):
        out["encoder_ms"] = stage_ms.get("encoder")
        out["sweep_ms"] = stage_ms.get("query")
        out["tail_ms"] = stage_ms.get("mesh")
    oc = extras.get("octree_mode") or {}
    if "octree" in oc:
        out["octree_s"] = oc["octree"]["seconds"]
        out["octree_dense_s"] = oc["dense"]["seconds"]
    for key, name in (("reference_loop", "ref_loop_fp32_ms"), ("reference_loop_reduced", "ref_loop_reduced_ms")):
        rl = extras.get(key) or {}
        if "ms_per_50k_chunk" in rl:
            out[name] = rl["ms_per_50k_chunk"]
            out[name + "_min"], _, out[name + "_max"] = rl["ms_per_50k_chunk_min_median_max"]
            out[name.replace("_ms", "_s_per_512_grid")] = rl["seconds_per_512_grid_at_this_rate"]
    fl = extras.get("dense_floor") or {}
    if "ms_per_step" in fl:
        out["dense_floor_ms_per_step"] = fl["ms_per_step"]
    # counters are collected by rocprofv3 --pmc on the shipped library (tools/profile_round.sh), not in this run: the committed summary
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_summary.json")) as f:
            pm = json.load(f)
        k = pm["kernels"].get("fp32" if precision == "fp32" else "bf16") or {}
        out["hbm_bytes_per_launch"] = k.get("hbm_bytes_per_launch")
        out["scratch_bytes_per_lane"] = k.get("scratch_bytes_per_lane")
        out["pmc_source"] = "profiles/pmc_summary.json (%s, lib %s)" % (pm.get("round"), str(pm.get("lib_sha256"))[:12])
    except Exception:   # noqa: BLE001 - a missing summary must not cost the bench line
        pass
    return out


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
            "seconds": dt,
            "note": "a plain OpenMP loop on every host thread of the GPU box: a stated baseline, not a like-for-like figure.  The "
                    "reference itself (PyTorch CPU, 8 cores of the build container) evaluates 3.8 - 4.9e4 queries/s "
                    "(profiles/r02_reference_cpu_times.json, tools/ref_time.py)"}


if __name__ == "__main__":
    main()
