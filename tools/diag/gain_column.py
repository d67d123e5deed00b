"""Diagnostic: one column of the gain-60 field: v11 vs v5 vs the point path per voxel, with the f16 parts of zf."""
import os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import precision_report as pr
import oracle
from surs_amd import native

R = 512
dev = native.require_gpu()
sd, Fl, Fh, keep = pr.noise_inputs(dev)
sd = {k: (v.clone() if torch.is_tensor(v) else np.array(v, copy=True)) for k, v in sd.items()}
gain = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
for m in ("mlp_lr.", "mlp_hr."):
    sd[m + "conv0.weight"][:, 320] *= gain
v11, _, ws = pr.sweeps(sd, Fl, Fh, R, ("fp32",), dev, kernel=11)
v5, _, _ = pr.sweeps(sd, Fl, Fh, R, ("fp32",), dev, kernel=5)
blob, _ = native.pack_mlp({k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in sd.items() if k.startswith("mlp_")}, "fp32", dev)
lg = lambda p: torch.log(p.double() / (1 - p.double()))
cal = pr.CALIB.reshape(-1)[:12]
for (I, J) in ((328, 459), (73, 480), (407, 94)):
    s0 = (I * R + J) * R
    pts = torch.from_numpy(oracle.grid_points(R, [-0.5] * 3, [0.5] * 3, s0, s0 + R)).to(dev)
    phr, plr, lhr, llr = native.query_points(pts, cal, 512, 200.0, Fl, Fh, blob, ws, want_logits=True)
    a = lg(v11["fp32"][1].view(-1)[s0:s0 + R]); b = lg(v5["fp32"][1].view(-1)[s0:s0 + R])
    zf = (pts[2] * 2.0 * 512.0 / 200.0)
    z0 = zf.half(); r1 = zf - z0.float(); z1 = r1.half(); r2 = r1 - z1.float(); z2 = r2.half()
    d11 = (a - llr.double()).abs().cpu().numpy(); d5 = (b - llr.double()).abs().cpu().numpy()
    print("column", I, J, "lr: max |v11 - pts| %.2e, |v5 - pts| %.2e" % (d11.max(), d5.max()))
    for k in np.argsort(-d11)[:10]:
        print("   k=%3d  zf=% .7f  z1=% .3e z2=% .3e  d11=%.2e d5=%.2e  logit %.4f" % (k, zf[k].item(), z1[k].item(), z2[k].item(), d11[k], d5[k], llr[k].item()))
    print("   per tile max d11:", [float("%.1e" % d11[t * 64:(t + 1) * 64].max()) for t in range(8)])
    print("   first 16 d11:", ["%.0e" % v for v in d11[:16]])
