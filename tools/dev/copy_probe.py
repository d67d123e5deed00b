"""D2H copies beside the column kernel: do they need compute units (blit kernels) or travel on the SDMA engines?"""
import os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common, gpu_common as g, oracle
from surs_amd import native
R = 512
fl, fh = common.synth_features(hl=256, hh=1024)
Fl, Fh = g.upload_nhwc(fl), g.upload_nhwc(fh)
ws = native.Workspace(g.dev())
mat = oracle.coords_matrix(R, [-0.5] * 3, [0.5] * 3)[:3].reshape(-1)
b = g.blob("bf16")
vh = torch.empty((R, R, R), dtype=torch.float32, device=g.dev()); vl = torch.empty_like(vh)
sweep = lambda: native.query_grid(0, R, R, R, mat, common.CALIB.reshape(-1)[:12], 512, 200.0, Fl, Fh, b, "bf16", ws, vh, vl)
src = torch.empty(256 << 20, dtype=torch.uint8, device=g.dev())
dst = [torch.empty(256 << 20, dtype=torch.uint8, pin_memory=True) for _ in range(4)]
side = torch.cuda.Stream()
def copies():
    with torch.cuda.stream(side):
        for d in dst: d.copy_(src, non_blocking=True)
        e = torch.cuda.Event(enable_timing=True); e.record(side)
    return e
def timed(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3, r
sweep(); torch.cuda.synchronize(); copies(); torch.cuda.synchronize()
tc = min(timed(copies)[0] for _ in range(3))
tsw = min(timed(sweep)[0] for _ in range(3))
def both():
    s0 = torch.cuda.Event(enable_timing=True); s0.record()
    e = copies(); sweep()
    s1 = torch.cuda.Event(enable_timing=True); s1.record()
    torch.cuda.synchronize()
    return s0.elapsed_time(e), s0.elapsed_time(s1)
rb = [both() for _ in range(3)]
print("env %s: 1 GiB D2H alone %.1f ms (%.1f GB/s); sweep alone %.1f ms; together: copies done after %.1f ms, sweep done after %.1f ms"
      % (os.environ.get("PROBE_TAG", "default"), tc, 1.0737 / tc * 1e3, tsw, min(r[0] for r in rb), min(r[1] for r in rb)))
