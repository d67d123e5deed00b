"""Per-step timeline of the split-bf16 layer kernel (diagnostic build: tools/build_trace.sh -DSURS_GEMM_TRACE).
On the GPU box: SURS_GEMM_TRACE=1 SURS_LIB_PATH=abl/libsurs_trace.so python tools/gpu_gemm_trace.py"""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common, gpu_common as g
from surs_amd import native, weights
fl, fh = common.synth_features(hl=256, hh=1024)
Fl, Fh = g.upload_nhwc(fl), g.upload_nhwc(fh)
ws = native.Workspace(g.dev())
b = g.blob("bf16")
n = 2000000
pts = torch.from_numpy(weights.synthetic_points(n, seed=2)).to(g.dev())
for _ in range(2):
    native.query_points(pts, common.CALIB.reshape(-1)[:12], 512, 200.0, Fl, Fh, b, ws)
torch.cuda.synchronize()
