"""Runs the fp32 point evaluator on fixed inputs and saves / compares the four outputs.  The layer-kernel generation is
chosen by the environment (read once per process): SURS_GEMM_X3=0 (fp32 MFMA), SURS_GEMM_BIG=0 (128x128 split-bf16 kernel
for every layer), SURS_GEMM_WAVES=16.  `python tools/gpu_points_cmp.py save|cmp file.npz`"""
import os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common, gpu_common as g
from surs_amd import native, weights

mode, path = sys.argv[1], sys.argv[2]
fl, fh = common.synth_features()
Fl, Fh = g.upload_nhwc(fl), g.upload_nhwc(fh)
ws = native.Workspace(g.dev())
out = {}
for n in (777, 20000):
    pts = torch.from_numpy(weights.synthetic_points(n, seed=11)).to(g.dev())
    r = native.query_points(pts, common.CALIB.reshape(-1)[:12], 512, 200.0, Fl, Fh, g.blob("bf16"), ws, want_logits=True)
    for name, t in zip(("pred_hr", "pred_lr", "logit_hr", "logit_lr"), r):
        out["%s_%d" % (name, n)] = t.cpu().numpy()
# a small bf16 sweep: its column constants come from the same layer kernels (transposed-output form)
import oracle
R = 48
mat = oracle.coords_matrix(R, [-0.5] * 3, [0.5] * 3)[:3].reshape(-1)
vh = torch.empty((R, R, R), dtype=torch.float32, device=g.dev()); vl = torch.empty_like(vh)
native.query_grid(0, R, R, R, mat, common.CALIB.reshape(-1)[:12], 512, 200.0, Fl, Fh, g.blob("bf16"), "bf16", ws, vh, vl)
out["grid_hr_%d" % R] = vh.cpu().numpy(); out["grid_lr_%d" % R] = vl.cpu().numpy()
if mode == "save":
    np.savez(path, **out)
else:
    ref = np.load(path)
    for k in sorted(out):
        print("%s max|diff|= %.3e equal=%d" % (k, float(np.abs(out[k] - ref[k]).max()), int(np.array_equal(out[k], ref[k]))))
