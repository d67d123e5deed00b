"""Where does column kernel v10 differ from v7?  (debug aid)  python tools/gpu_v10_debug.py R prec"""
import os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import precision_report as pr
from surs_amd import native
R = int(sys.argv[1]); prec = sys.argv[2]
dev = native.require_gpu()
sd, Fl, Fh, keep = pr.noise_inputs(dev)
a, _, _ = pr.sweeps(sd, Fl, Fh, R, (prec,), dev, kernel=7)
for rep in range(2):
    b, _, _ = pr.sweeps(sd, Fl, Fh, R, (prec,), dev, kernel=10)
    for i, tag in enumerate(("hr", "lr")):
        x, y = a[prec][i], b[prec][i]
        bad = ~((x == y) | (torch.isnan(x) & torch.isnan(y)))
        nan = torch.isnan(y)
        print("rep", rep, tag, "mismatch", int(bad.sum()), "nan", int(nan.sum()), "of", x.numel())
        if bad.any():
            idx = bad.nonzero()
            cols = torch.unique(idx[:, 0] * R + idx[:, 1])
            print("   columns with mismatch:", cols.numel(), "of", R * R, "first", cols[:8].tolist())
            tiles = torch.unique(idx[:, 2] // 128)
            print("   z tiles:", tiles.tolist(), " first mismatches:", idx[:6].tolist())
            c0 = idx[0]
            print("   column", c0[:2].tolist(), "v7", x[c0[0], c0[1], max(0, c0[2]-2):c0[2]+4].tolist(), "v10", y[c0[0], c0[1], max(0, c0[2]-2):c0[2]+4].tolist())
            per_col = bad.view(R * R, R).sum(1)
            print("   mismatching voxels per bad column: min %d max %d" % (int(per_col[per_col > 0].min()), int(per_col.max())))
