"""Quick v12 check on one 64-plane slab of the 512^3 noise (or body) field: bits against v10 + slab times (3 repetitions)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import precision_report as pr
from surs_amd import native
field = sys.argv[1] if len(sys.argv) > 1 else "noise"
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = native.require_gpu()
if field == "body":
    sd, Fl, Fh = pr.body_inputs(dev)
else:
    sd, Fl, Fh, keep = pr.noise_inputs(dev)
R = 512
mlp = {k: v for k, v in sd.items() if k.startswith("mlp_")}
cal = pr.CALIB.reshape(-1)[:12]
m = np.eye(4); m[0, 0] = m[1, 1] = m[2, 2] = 1.0 / R; m[:3, 3] = -0.5
ws = native.Workspace(dev)
blob, _ = native.pack_mlp(mlp, prec, dev)
res, tm = {}, {}
for kv in (10, 12):
    vh = torch.empty((64, R, R), dtype=torch.float32, device=dev); vl = torch.empty_like(vh)
    ts = []
    for rep in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        native.query_grid(192, 256, R, R, m[:3].reshape(-1), cal, 512, 200.0, Fl, Fh, blob, prec, ws, vh, vl, kernel=kv)
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
    res[kv] = (vh, vl); tm[kv] = ts[1:]
same = [bool(torch.equal(res[10][i], res[12][i])) for i in range(2)]
print("%s %s: bitwise hr/lr %s; v10 slab ms %s; v12 slab ms %s; ratio %.3f" % (field, prec, same, ["%.2f" % t for t in tm[10]], ["%.2f" % t for t in tm[12]], min(tm[10]) / min(tm[12])), flush=True)
