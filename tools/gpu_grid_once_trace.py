"""One sweep at R (default 256) in the given precision with the diagnostic library's phase stamps printed (see tools/build_trace.sh):
    SURS_V3_TRACE=1 SURS_LIB_PATH=abl/libsurs_trace.so python tools/gpu_grid_once_trace.py fp32"""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common, gpu_common as g, oracle
from surs_amd import native
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 256
fl, fh = common.synth_features()
Fl, Fh = g.upload_nhwc(fl), g.upload_nhwc(fh)
ws = native.Workspace(g.dev())
mat = oracle.coords_matrix(R, [-0.5] * 3, [0.5] * 3)[:3].reshape(-1)
b = g.blob("f16" if dt == "fp16" else "bf16")
vh = torch.empty((R, R, R), dtype=torch.float32, device=g.dev()); vl = torch.empty_like(vh)
for _ in range(3):
    native.query_grid(0, R, R, R, mat, common.CALIB.reshape(-1)[:12], 512, 200.0, Fl, Fh, b, dt, ws, vh, vl)
torch.cuda.synchronize()
