"""Slab time of one column kernel (argv: kernel field precision reps) on this process's library: prints min and median ms."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import precision_report as pr
from surs_amd import native
kv = int(sys.argv[1]); field = sys.argv[2]; prec = sys.argv[3]; reps = int(sys.argv[4])
dev = native.require_gpu()
sd, Fl, Fh = pr.body_inputs(dev) if field == "body" else pr.noise_inputs(dev)[:3]
R = 512
mlp = {k: v for k, v in sd.items() if k.startswith("mlp_")}
cal = pr.CALIB.reshape(-1)[:12]
m = np.eye(4); m[0, 0] = m[1, 1] = m[2, 2] = 1.0 / R; m[:3, 3] = -0.5
ws = native.Workspace(dev)
blob, _ = native.pack_mlp(mlp, prec, dev)
vh = torch.empty((128, R, R), dtype=torch.float32, device=dev); vl = torch.empty_like(vh)
ts = []
for rep in range(reps + 1):
    torch.cuda.synchronize(); t = time.perf_counter()
    native.query_grid(192, 320, R, R, m[:3].reshape(-1), cal, 512, 200.0, Fl, Fh, blob, prec, ws, vh, vl, kernel=kv)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
ts = sorted(ts[1:])
import hashlib
print("%.3f %.3f %s" % (ts[0], ts[len(ts) // 2], hashlib.sha256(vh.cpu().numpy().tobytes()).hexdigest()[:12]))
