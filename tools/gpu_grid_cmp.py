"""Compares a column-kernel version (SURS_GRID_KERNEL of this process) against volumes saved by another run.

    SURS_GRID_KERNEL=3 python tools/gpu_grid_cmp.py save /tmp/v3.npz
    SURS_GRID_KERNEL=10 python tools/gpu_grid_cmp.py cmp  /tmp/v3.npz
"""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common, gpu_common as g, oracle
from surs_amd import native

mode, path = sys.argv[1], sys.argv[2]
fl, fh = common.synth_features()
Fl, Fh = g.upload_nhwc(fl), g.upload_nhwc(fh)
ws = native.Workspace(g.dev())
out = {}
for dt in ("bf16", "fp16"):
    for R in (40, 136):
        mat = oracle.coords_matrix(R, [-0.5] * 3, [0.5] * 3)[:3].reshape(-1)
        b = g.blob("f16" if dt == "fp16" else "bf16")
        runs = []
        for rep in range(3):
            vh, vl = native.query_grid(0, R, R, R, mat, common.CALIB.reshape(-1)[:12], 512, 200.0, Fl, Fh, b, dt, ws)
            runs.append((vh.cpu().numpy().copy(), vl.cpu().numpy().copy()))
        stable = all((r[0] == runs[0][0]).all() and (r[1] == runs[0][1]).all() for r in runs)
        out["%s_%d_hr" % (dt, R)], out["%s_%d_lr" % (dt, R)] = runs[0]
        print(dt, R, "stable" if stable else "UNSTABLE", "nan" if np.isnan(runs[0][0]).any() else "finite")
if mode == "save":
    np.savez(path, **out)
else:
    ref = np.load(path)
    for k in sorted(out):
        d = np.abs(out[k].astype(np.float64) - ref[k]).max()
        print(k, "max|diff| = %.3e" % d, "bitwise" if (out[k] == ref[k]).all() else "")
