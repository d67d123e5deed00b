"""One rank's timeline of a slab-mode (BASELINE configs[3]) reconstruction, measured on ONE dedicated MI355X.

No multi-GPU node has been available, and several processes sharing one GPU (tests/test_gpu_dist.py) time nothing useful.  This
tool runs the code of ONE rank r of P - dist.encode_sharded + dist.reconstruction_sharded_once, unchanged - with the collectives
replaced by local stand-ins that move the same bytes on the device / host (the halo plane is computed beforehand and copied in where
the receive would land; the gathers return this rank's row P times; the mesh delivery writes this rank's part into a block sized
for P parts).  What it measures: the compute of the critical path of one rank (strip encoder, slab sweep with the extraction
pipelined into it, the tail after the last launch, fix-up, the copy of the rank's mesh part) on a GPU of its own.  What it does NOT
contain: the latency of the real collectives (1 all-gather of feature_lr strips, 1 halo send / receive, 3 small host-side
all-gathers, 1 boundary-id send / receive) - DESIGN.md section 7 adds those as stated constants.

    python tools/gpu_slab_stage_times.py [R] [precision] [body|noise]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from surs_amd import dist as sd, mesh_util, model, native, options, train_util, weights  # noqa: E402


class FakeWork:
    def wait(self):
        pass


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    field = sys.argv[3] if len(sys.argv) > 3 else "noise"
    dev = native.require_gpu()
    flags = ["--loadSize", "1024", "--residual", "--b_min", "-0.5", "-0.5", "-0.5", "--b_max", "0.5", "0.5", "0.5", "--precision", prec]
    opt = options.BaseOptions().parse(flags)
    sdict = weights.synthetic_state_dict(opt, seed=0)
    if field == "body":
        sdict = dict(sdict)
        sdict.update(weights.body_state_dict(opt))
    net = model.SuRSNet(opt).to(device=dev)
    net.load_state_dict(sdict)
    net.eval()
    image = torch.from_numpy(weights.synthetic_image(512, seed=1)).to(dev)
    calib = train_util.gen_calib().to(dev)
    b_min, b_max = np.array([-0.5] * 3), np.array([0.5] * 3)
    body_feats = None
    if field == "body":
        import precision_report as pr
        fl, fh = weights.body_features(256, 1024)
        body_feats = (pr._upload(fl, dev), pr._upload(fh, dev))

    def encode_full():
        _, f_lr, f_hr = net.super_res(image)
        net.filter_hr(f_hr)
        net.filter_lr(f_lr)

    def set_body():
        if body_feats is not None:   # (the body field's features are synthetic: the encoder is timed, its output replaced)
            from surs_amd.model import _as_nchw_view
            net.im_feat_list_lr, net.im_feat_list_hr = [_as_nchw_view(body_feats[0])], [_as_nchw_view(body_feats[1])]

    # ---- one GPU, the product path: the reference point of the speed-up
    out = {"resolution": R, "precision": prec, "field": field, "ranks": {}}
    for _ in range(2):
        encode_full()
        set_body()
        mesh_util.reconstruction(opt, net, dev, calib, R, b_min, b_max, use_octree=False, want_normals=False)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        encode_full()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        set_body()
        mesh_util.reconstruction(opt, net, dev, calib, R, b_min, b_max, use_octree=False, want_normals=False)
        torch.cuda.synchronize()
        ts.append((t1 - t0, time.perf_counter() - t1))
    out["one_gpu"] = {"encoder_ms": min(t[0] for t in ts) * 1e3, "reconstruction_ms": min(t[1] for t in ts) * 1e3}
    out["one_gpu"]["step_ms"] = out["one_gpu"]["encoder_ms"] + out["one_gpu"]["reconstruction_ms"]
    print("one GPU:", out["one_gpu"], flush=True)

    # ---- the planes of the whole grid that ranks receive as halos (computed once, outside every timed region)
    from surs_amd.sdf import create_grid
    _, mat = create_grid(R, R, R, b_min, b_max)
    m12 = mat[:3].reshape(-1)
    cal = calib[0].cpu().numpy().reshape(-1)[:12]
    real_world, real_gather, real_exchange, real_agree, real_allgather, real_staged = (sd._world, sd.all_gather_rows, sd.Exchange, sd._agree,
                                                                                      sd.dist.all_gather, sd._host_staged)

    for P in (2, 4, 8):
        # (the body field's surface does not reach the outer slabs of an 8-way split: a rank of its own would stop at "no surface")
        for r in (sorted({P // 2 - 1, P // 2}) if field == "body" else sorted({0, P // 2, P - 1})):
            i0, i1 = sd.slab_range(R, r, P)
            encode_full()
            set_body()
            fl, fh = net.features()
            halo = [None, None]
            if r < P - 1:
                vh, vl = native.query_grid(i1, i1 + 1, R, R, m12, cal, *net._zscale(), fl, fh, net._mlp_blob(), prec, net._workspace())
                halo = [vh[0].clone(), vl[0].clone()]

            class Ex:
                def __init__(self, group=None):
                    self.n = 0

                def send(self, t, dst):
                    t.contiguous()

                def recv(self, out_t, src):
                    if out_t.dtype == torch.float32 and out_t.dim() == 2 and halo[0] is not None:
                        out_t.copy_(halo[self.n % 2])
                        self.n += 1
                    else:
                        out_t.zero_()

                def start(self):
                    return self

                def wait(self):
                    pass

            def fake_all_gather(outs, mine, group=None):
                for o in outs:
                    o.copy_(mine)
                return FakeWork()

            sd._world = lambda group=None, P=P, r=r: (P, r)
            sd.all_gather_rows = lambda row, device, group=None, P=P: np.tile(np.asarray([float(v) for v in row], np.float64)[None], (P, 1))
            sd.Exchange = Ex
            sd._agree = lambda ok, d, g: ok
            sd.dist.all_gather = fake_all_gather
            sd._host_staged = lambda t, group=None: False
            try:
                best = None
                for it in range(4):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    used = sd.encode_sharded(net, image, calib, R, b_min, b_max)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    set_body()
                    ev = torch.cuda.Event(enable_timing=True)
                    ev0 = torch.cuda.Event(enable_timing=True)
                    ev0.record()
                    sd.reconstruction_sharded_once(opt, net, calib, R, b_min, b_max, dst=r, timing=ev, copy_out=False)
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
                    rec = {"sharded_encoder": bool(used), "encoder_ms": (t1 - t0) * 1e3, "sweep_ms": ev0.elapsed_time(ev),
                           "reconstruction_ms": (t2 - t1) * 1e3, "tail_ms": (t2 - t1) * 1e3 - ev0.elapsed_time(ev)}
                    rec["step_ms"] = rec["encoder_ms"] + rec["reconstruction_ms"]
                    if it >= 1 and (best is None or rec["step_ms"] < best["step_ms"]):
                        best = rec
                out["ranks"]["P%d_r%d" % (P, r)] = best
                print("P = %d, rank %d (planes %d..%d):" % (P, r, i0, i1), {k: (round(v, 2) if isinstance(v, float) else v) for k, v in best.items()},
                      flush=True)
            finally:
                sd._world, sd.all_gather_rows, sd.Exchange, sd._agree, sd.dist.all_gather = real_world, real_gather, real_exchange, real_agree, real_allgather
                sd._host_staged = real_staged
                sd.SharedMeshStore.release_all()
        worst = max(v["step_ms"] for k, v in out["ranks"].items() if k.startswith("P%d_" % P))
        out["P%d" % P] = {"slowest_rank_step_ms": worst, "speedup_without_collective_latency": out["one_gpu"]["step_ms"] / worst}
        print("P = %d: slowest emulated rank %.2f ms -> %.2fx (collective latencies not included)" % (P, worst, out["one_gpu"]["step_ms"] / worst),
              flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
