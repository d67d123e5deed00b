import os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import precision_report as pr
from surs_amd import native
dev = native.require_gpu()
sd, Fl, Fh, keep = pr.noise_inputs(dev)
R = 512
mlp = {k: v for k, v in sd.items() if k.startswith("mlp_")}
cal = pr.CALIB.reshape(-1)[:12]
m = np.eye(4); m[0, 0] = m[1, 1] = m[2, 2] = 1.0 / R; m[:3, 3] = -0.5
ws = native.Workspace(dev)
blob, _ = native.pack_mlp(mlp, "bf16", dev)
vh = torch.empty((64, R, R), dtype=torch.float32, device=dev); vl = torch.empty_like(vh)
native.query_grid(0, 64, R, R, m[:3].reshape(-1), cal, 512, 200.0, Fl, Fh, blob, "bf16", ws, vh, vl, kernel=12)
torch.cuda.synchronize()
a = vl.cpu().numpy(); b = vh.cpu().numpy()
bad = np.argwhere(a != b)
k = bad[:, 2]
print("halves differ at", len(bad), "by storing wave ((k % 128) // 32):", np.bincount((k % 128) // 32, minlength=4), "by voxel quarter:", np.bincount((k % 32) // 16, minlength=2))
