"""Times the bf16 grid sweep at R=256 (used with SURS_LIB_PATH for the ablation variants)."""
import os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common, gpu_common as g, oracle
from surs_amd import native
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
fl, fh = common.synth_features(hl=256, hh=1024) if R >= 512 else common.synth_features()
Fl, Fh = g.upload_nhwc(fl), g.upload_nhwc(fh)
ws = native.Workspace(g.dev())
mat = oracle.coords_matrix(R, [-0.5] * 3, [0.5] * 3)[:3].reshape(-1)
b = g.blob("bf16")
vh = torch.empty((R, R, R), dtype=torch.float32, device=g.dev()); vl = torch.empty_like(vh)
f = lambda: native.query_grid(0, R, R, R, mat, common.CALIB.reshape(-1)[:12], 512, 200.0, Fl, Fh, b, "bf16", ws, vh, vl)
f(); torch.cuda.synchronize()
t = time.time()
for _ in range(3): f()
torch.cuda.synchronize()
dt = (time.time() - t) / 3
import hashlib
dg = hashlib.sha256(vh.cpu().numpy().tobytes() + vl.cpu().numpy().tobytes()).hexdigest()[:16]
print("%s  %.4f s  %.3e pts/s  volumes %s" % (os.environ.get("SURS_LIB_PATH", "default"), dt, R ** 3 / dt, dg))
