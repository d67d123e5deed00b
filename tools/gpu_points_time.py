"""Times the fp32 point evaluator (surs_query_points: BASELINE configs[1]) for a few batch sizes."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common, gpu_common as g
from surs_amd import native, weights
fl, fh = common.synth_features(hl=256, hh=1024)
Fl, Fh = g.upload_nhwc(fl), g.upload_nhwc(fh)
ws = native.Workspace(g.dev())
b = g.blob("bf16")
for n in [int(a) for a in sys.argv[1:]] or (50000, 400000, 2000000):
    pts = torch.from_numpy(weights.synthetic_points(n, seed=2)).to(g.dev())
    f = lambda: native.query_points(pts, common.CALIB.reshape(-1)[:12], 512, 200.0, Fl, Fh, b, ws)
    f(); torch.cuda.synchronize()
    t = time.time()
    reps = 5
    for _ in range(reps): f()
    torch.cuda.synchronize()
    dt = (time.time() - t) / reps
    print("fp32 points n=%d: %.3f ms  %.3e pts/s  (%.1f TFLOP/s algorithmic)" % (n, dt * 1e3, n / dt, n / dt * 4564998 / 1e12))
