"""Diagnostic: where does the restated fp32-grade column kernel (v11) leave the dense one (v5) on fields whose layer-0 depth column
is scaled (gain)?  Prints, per gain, the whole-volume max / mean |d logit| and the structure of the largest differences.
    python tools/diag/gain_outliers.py [R]"""
import os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import precision_report as pr
from surs_amd import native, _lib

R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = native.require_gpu()
sd0, Fl, Fh, keep = pr.noise_inputs(dev)
L = _lib.lib()
lg = lambda p: torch.log(p.double() / (1 - p.double()))
for gain in (1.0, 4.0, 16.0, 60.0):
    sd = {k: (v.clone() if torch.is_tensor(v) else np.array(v, copy=True)) for k, v in sd0.items()}
    for m in ("mlp_lr.", "mlp_hr."):
        sd[m + "conv0.weight"][:, 320] *= gain
    ref, _, _ = pr.sweeps(sd, Fl, Fh, R, ("fp32",), dev, kernel=5)
    for split in (0, 3):
        L.surs_set_operand_split(split)
        new, _, _ = pr.sweeps(sd, Fl, Fh, R, ("fp32",), dev, kernel=11)
        L.surs_set_operand_split(0)
        for i, tag in enumerate(("hr", "lr")):
            a, b = new["fp32"][i], ref["fp32"][i]
            ok = (a > 0.0067) & (a < 0.9933) & (b > 0.0067) & (b < 0.9933)
            d = torch.zeros_like(a, dtype=torch.float32)
            d[ok] = (lg(a[ok]) - lg(b[ok])).abs().float()
            top = torch.topk(d.view(-1), 12)
            idx = top.indices.cpu().numpy()
            ii, jj, kk = idx // (R * R), (idx // R) % R, idx % R
            print("gain %g split %d %s: max %.3e mean %.3e  n>1e-4: %d  n>3e-5: %d" % (gain, split, tag, d.max().item(), d[ok].mean().item(),
                  int((d > 1e-4).sum()), int((d > 3e-5).sum())), flush=True)
            print("   top:", [(int(x), int(y), int(z), "%.1e" % v) for x, y, z, v in zip(ii, jj, kk, top.values.cpu().numpy())][:8])
            # how are the large differences distributed over columns / tiles?
            big = (d > 3e-5)
            if int(big.sum()):
                per_col = big.view(R * R, R).sum(1)
                print("   columns with any: %d; max per column %d; per z-tile histogram %s" % (int((per_col > 0).sum()), int(per_col.max()),
                      big.view(R * R, R // 64, 64).sum((0, 2)).cpu().numpy().tolist()))
