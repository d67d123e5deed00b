"""Exploratory timing on the GPU box (not a test): grid kernel throughput, MC time."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
import gpu_common as g  # noqa: E402
import mc_volumes  # noqa: E402
import oracle  # noqa: E402
from surs_amd import native  # noqa: E402


def timeit(fn, n=3):
    torch.cuda.synchronize()
    fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n


def main():
    print("device", native.device_info())
    fl, fh = common.synth_features(hl=256, hh=1024) if "--big" in sys.argv else common.synth_features()
    Fl, Fh = g.upload_nhwc(fl), g.upload_nhwc(fh)
    ws = native.Workspace(g.dev())
    for R in (64, 128, 256):
        mat = oracle.coords_matrix(R, [-0.5] * 3, [0.5] * 3)[:3].reshape(-1)
        for dt in ("bf16", "fp16"):
            b = g.blob("f16" if dt == "fp16" else "bf16")
            vh = torch.empty((R, R, R), dtype=torch.float32, device=g.dev())
            vl = torch.empty_like(vh)
            t = timeit(lambda: native.query_grid(0, R, R, R, mat, common.CALIB.reshape(-1)[:12], 512, 200.0, Fl, Fh, b, dt, ws, vh, vl), 2)
            print("grid %s R=%d: %.4f s  %.3e pts/s  (%.1f TFLOP/s algorithmic)" % (dt, R, t, R ** 3 / t, R ** 3 / t * 4564998 / 1e12))
        if R <= 64:
            t = timeit(lambda: native.query_grid(0, R, R, R, mat, common.CALIB.reshape(-1)[:12], 512, 200.0, Fl, Fh, g.blob("bf16"), "fp32", ws), 1)
            print("grid fp32 R=%d: %.4f s  %.3e pts/s" % (R, t, R ** 3 / t))
    pts = torch.rand((3, 50000), device=g.dev()) - 0.5
    t = timeit(lambda: native.query_points(pts, common.CALIB.reshape(-1)[:12], 512, 200.0, Fl, Fh, g.blob("bf16"), ws), 3)
    print("query_points fp32 50k: %.4f s  %.3e pts/s" % (t, 50000 / t))
    for n in (128, 256, 512):
        vol = torch.from_numpy(mc_volumes.blob(n)).to(g.dev())
        t = timeit(lambda: native.marching_cubes_lewiner(vol, 0.5, ws), 2)
        print("mc blob %d: %.4f s (%.1f Mvox/s)" % (n, t, n ** 3 / t / 1e6))


if __name__ == "__main__":
    main()
