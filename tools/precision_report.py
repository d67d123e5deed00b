"""What the reduced-precision column kernels (bf16 / fp16) do to the RESULT of a dense reconstruction, measured against the
fp32-grade sweep (column kernel v11, itself held to the reference's goldens at 1e-4) on the same features and weights:

  * field:  max / mean |d logit|, max |d occupancy|, number of voxels on the other side of the 0.5 level
  * mesh :  vertex / face counts and their deltas, and a symmetric nearest-vertex distance in voxel units between the two
            meshes (for every vertex of one mesh the closest vertex of the other among the vertices within half a voxel
            per axis; "unmatched" = none there) - mean, 99.9th percentile, max and the unmatched count, both directions.

Two fields: `noise` = the bench's field (seeded random weights + the encoder's features of the synthetic image: a level
crossing in nearly every column) and `body` = weights.body_state_dict + weights.body_features (smooth, one closed blob of
body-like extent).  Measurement code (tests/test_gpu_precision.py asserts on it, bench.py reports it): torch is used
for the statistics; everything measured runs through the C ABI.

    python tools/precision_report.py [R] [body|noise|both]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from surs_amd import native, options, weights  # noqa: E402

FLAGS = ["--loadSize", "1024", "--residual", "--b_min", "-0.5", "-0.5", "-0.5", "--b_max", "0.5", "0.5", "0.5"]
CALIB = np.diag([2.0, -2.0, 2.0, 1.0]).astype(np.float32)


def _upload(feat_chw, dev):
    c, h, w = feat_chw.shape
    t = torch.from_numpy(np.ascontiguousarray(feat_chw.transpose(1, 2, 0))).to(dev)
    return native.Img(h, w, c, c, t.reshape(-1))


def field_stats(v, ref, chunk=1 << 24, plim=1e-6):
    """v, ref: float32 device volumes of occupancies.  Logits are recovered in float64 from voxels with plim < p < 1 - plim: an fp32
    occupancy resolves its logit to 6e-8 / min(p, 1 - p) only (1e-5 at plim = 0.0067, i.e. |logit| < 5; 0.06 at plim = 1e-6)."""
    a, b = v.reshape(-1), ref.reshape(-1)
    n = a.numel()
    mx_l = sum_l = 0.0
    mx_p = 0.0
    flips = 0
    cnt = 0
    for s in range(0, n, chunk):
        x, y = a[s:s + chunk].double(), b[s:s + chunk].double()
        mx_p = max(mx_p, (x - y).abs().max().item())
        flips += int(((x > 0.5) != (y > 0.5)).sum().item())
        ok = (x > plim) & (x < 1 - plim) & (y > plim) & (y < 1 - plim)
        d = (torch.log(x / (1 - x)) - torch.log(y / (1 - y))).abs()[ok]
        if d.numel():
            mx_l = max(mx_l, d.max().item())
            sum_l += d.sum().item()
            cnt += d.numel()
    return {"max_abs_dlogit": mx_l, "mean_abs_dlogit": sum_l / max(cnt, 1), "max_abs_docc": mx_p, "flipped_voxels": flips,
            "flipped_fraction": flips / n}


def nearest_vertex_distance(a, b, R, sample=None, seed=0, chunk=1 << 21):
    """For every vertex of a ([N,3] float32, index coordinates) the distance to the closest vertex of b among those in the
    eight unit cells around it (all vertices within 0.5 per axis are found, so a reported distance < 0.5 is the true
    nearest distance).  Returns a float32 tensor with inf where those cells hold no vertex of b."""
    if sample is not None and a.shape[0] > sample:
        g = torch.Generator(device="cpu").manual_seed(seed)
        a = a[torch.randperm(a.shape[0], generator=g)[:sample].to(a.device)]
    kb = ((b[:, 0].floor().long().clamp(0, R - 1) * R + b[:, 1].floor().long().clamp(0, R - 1)) * R
          + b[:, 2].floor().long().clamp(0, R - 1))
    kb, order = torch.sort(kb)
    bs = b[order]
    nb = bs.shape[0]
    out = torch.full((a.shape[0],), float("inf"), dtype=torch.float32, device=a.device)
    for s in range(0, a.shape[0], chunk):
        p = a[s:s + chunk]
        best = torch.full((p.shape[0],), float("inf"), dtype=torch.float32, device=a.device)
        for ox in (-0.5, 0.5):
            for oy in (-0.5, 0.5):
                for oz in (-0.5, 0.5):
                    c = (p + torch.tensor([ox, oy, oz], device=p.device)).floor().long()
                    valid = ((c >= 0) & (c < R)).all(1)
                    key = (c[:, 0] * R + c[:, 1]) * R + c[:, 2]
                    lo = torch.searchsorted(kb, key)
                    hi = torch.searchsorted(kb, key, right=True)
                    cnt = torch.where(valid, hi - lo, torch.zeros_like(lo))
                    for m in range(int(cnt.max().item()) if cnt.numel() else 0):
                        ok = cnt > m
                        idx = (lo + m).clamp(max=nb - 1)
                        d = (bs[idx] - p).pow(2).sum(1)
                        best = torch.where(ok & (d < best), d, best)
        out[s:s + chunk] = best.sqrt()
    return out


def _dist_summary(d):
    ok = torch.isfinite(d)
    m = d[ok]
    if m.numel() == 0:
        return {"n": int(d.numel()), "unmatched": int(d.numel())}
    return {"n": int(d.numel()), "unmatched": int((~ok).sum().item()), "mean": m.mean().item(),
            "p999": torch.quantile(m[:: max(1, m.numel() // 4000000)].double(), 0.999).item(), "max": m.max().item()}


def mesh_stats(ws, vol, ref, R, sample=1 << 20):
    v, f, _, _ = native.marching_cubes_lewiner(vol, 0.5, ws, want_normals=False)
    vr, fr, _, _ = native.marching_cubes_lewiner(ref, 0.5, ws, want_normals=False)
    return {"verts": int(v.shape[0]), "faces": int(f.shape[0]), "verts_ref": int(vr.shape[0]), "faces_ref": int(fr.shape[0]),
            "to_ref": _dist_summary(nearest_vertex_distance(v, vr, R, sample)),
            "from_ref": _dist_summary(nearest_vertex_distance(vr, v, R, sample, seed=1))}


def sweeps(sd, Fl, Fh, R, precisions, dev, zmul=512, zdiv=200.0, kernel=0):
    """{precision: (vol_hr, vol_lr)} of the dense R^3 sweep, plus the sweep times.  kernel: column-kernel version (0 = default)."""
    mlp = {k: v for k, v in sd.items() if k.startswith("mlp_")}
    cal = CALIB.reshape(-1)[:12]
    m = np.eye(4)
    m[0, 0] = m[1, 1] = m[2, 2] = 1.0 / R
    m[:3, 3] = -0.5
    ws = native.Workspace(dev)
    out, times = {}, {}
    for prec in precisions:
        blob, _ = native.pack_mlp(mlp, prec, dev)
        vh = torch.empty((R, R, R), dtype=torch.float32, device=dev)
        vl = torch.empty_like(vh)
        native.query_grid(0, min(R, 8), R, R, m[:3].reshape(-1), cal, zmul, zdiv, Fl, Fh, blob, prec, ws, vh[:8], vl[:8], kernel=kernel)   # warm-up
        torch.cuda.synchronize()
        t = time.perf_counter()
        native.query_grid(0, R, R, R, m[:3].reshape(-1), cal, zmul, zdiv, Fl, Fh, blob, prec, ws, vh, vl, kernel=kernel)
        torch.cuda.synchronize()
        times[prec] = time.perf_counter() - t
        out[prec] = (vh, vl)
    return out, times, ws


def report(sd, Fl, Fh, R, dev, precisions=("bf16", "fp16"), sample=1 << 20):
    vols, times, ws = sweeps(sd, Fl, Fh, R, ("fp32",) + tuple(precisions), dev)
    rep = {"resolution": R, "reference": "fp32-grade column kernel (v11)", "sweep_s": times}
    for prec in precisions:
        r = {}
        for i, tag in enumerate(("hr", "lr")):
            r[tag] = field_stats(vols[prec][i], vols["fp32"][i])
            r[tag]["mesh"] = mesh_stats(ws, vols[prec][i], vols["fp32"][i], R, sample=sample)
        rep[prec] = r
    return rep


def body_inputs(dev, hl=256, hh=1024):
    opt = options.BaseOptions().parse(FLAGS)
    fl, fh = weights.body_features(hl, hh)
    return weights.body_state_dict(opt), _upload(fl, dev), _upload(fh, dev)


def noise_inputs(dev, H=512, encoder_precision="fp32"):
    """The bench's field: seeded random weights, the encoder's features of the H x H synthetic image.  encoder_precision: "fp32"
    (two f16 parts, fp32-grade) or "f16" (one f16 product per MAC in the 3x3 convolutions: the encoder of --precision bf16 / fp16)."""
    from surs_amd import model
    opt = options.BaseOptions().parse(FLAGS + ["--encoder_precision", encoder_precision])
    sd = weights.synthetic_state_dict(opt, seed=0)
    net = model.SuRSNet(opt).to(device=dev)
    net.load_state_dict(sd)
    net.eval()
    _, f_lr, f_hr = net.super_res(torch.from_numpy(weights.synthetic_image(H, seed=1)).to(dev))
    net.filter_hr(f_hr)
    net.filter_lr(f_lr)
    Fl, Fh = net.features()
    return sd, Fl, Fh, net


def encoder_report(dev, R=512, precisions=("bf16", "fp16"), sample=1 << 20):
    """What the reduced-precision ENCODER (--encoder_precision f16, opt-in) adds: the feature maps
    against the fp32-grade encoder's, and the whole reduced pipeline (f16 encoder + 16-bit sweep) against the whole fp32-grade one
    (fp32-grade encoder + fp32-grade sweep) on the bench's noise field, in the terms of report()."""
    sd, Fl, Fh, keep = noise_inputs(dev, encoder_precision="fp32")
    sd2, Gl, Gh, keep2 = noise_inputs(dev, encoder_precision="f16")
    rep = {"resolution": R}
    for tag, a, b in (("im_feat_lr", Gl, Fl), ("im_feat_hr", Gh, Fh)):
        d = (a.buf - b.buf).abs()
        rep[tag] = {"max_abs_err_over_absmax": d.max().item() / b.buf.abs().max().item(),
                    "mean_abs_err_over_mean_abs": d.mean().item() / b.buf.abs().mean().item()}
    ref, t_ref, ws = sweeps(sd, Fl, Fh, R, ("fp32",), dev)
    for prec in precisions:
        new, t_new, _ = sweeps(sd, Gl, Gh, R, (prec,), dev)
        r = {}
        for i, tag in enumerate(("hr", "lr")):
            r[tag] = field_stats(new[prec][i], ref["fp32"][i])
            r[tag]["mesh"] = mesh_stats(ws, new[prec][i], ref["fp32"][i], R, sample=sample)
        rep[prec] = r
        del new
    return rep


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    which = sys.argv[2] if len(sys.argv) > 2 else "both"
    dev = native.require_gpu()
    out = {}
    if which in ("body", "both"):
        sd, Fl, Fh = body_inputs(dev)
        out["body"] = report(sd, Fl, Fh, R, dev)
    if which in ("noise", "both"):
        sd, Fl, Fh, keep = noise_inputs(dev)
        out["noise"] = report(sd, Fl, Fh, R, dev)
    if which in ("encoder", "both"):
        out["encoder_f16"] = encoder_report(dev, R)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
