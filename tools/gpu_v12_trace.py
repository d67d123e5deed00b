"""Phase stamps of a column kernel on the bench's noise field (diagnostic library: tools/build_trace.sh), one 64-plane slab of 512^3:
    SURS_V3_TRACE=1 SURS_LIB_PATH=abl/libsurs_trace.so python tools/gpu_v12_trace.py [kernel] [precision]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import precision_report as pr
from surs_amd import native
kernel = int(sys.argv[1]) if len(sys.argv) > 1 else 12
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
field = sys.argv[3] if len(sys.argv) > 3 else "noise"
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
vh = torch.empty((64, R, R), dtype=torch.float32, device=dev); vl = torch.empty_like(vh)
for rep in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    native.query_grid(192, 256, R, R, m[:3].reshape(-1), cal, 512, 200.0, Fl, Fh, blob, prec, ws, vh, vl, kernel=kernel)
    torch.cuda.synchronize()
    print("kernel %d %s %s: slab of 64 planes %.2f ms" % (kernel, prec, field, 1e3 * (time.perf_counter() - t)), flush=True)
