"""fp32-grade column kernel (v5, split-f16 operands) against the fp32 point path (split-bf16 layer kernels, itself held to
the reference's goldens at 1e-4) on the same grid voxels, in logit space, and its sweep time.

    python tools/gpu_v5_check.py [R_time]
"""
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
import oracle  # noqa: E402
from surs_amd import native  # noqa: E402


def logit(p):
    p = p.double()
    return torch.log(p / (1.0 - p))


def main():
    Rt = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = g.dev()
    fl, fh = common.synth_features()
    Fl, Fh = g.upload_nhwc(fl), g.upload_nhwc(fh)
    ws = native.Workspace(dev)
    blob = g.blob("bf16")
    cal = common.CALIB.reshape(-1)[:12]
    for R in (40, 64):
        mat = oracle.coords_matrix(R, [-0.5] * 3, [0.5] * 3)[:3].reshape(-1)
        vh, vl = native.query_grid(0, R, R, R, mat, cal, 512, 200.0, Fl, Fh, blob, "fp32", ws)
        pts = torch.from_numpy(oracle.grid_points(R, [-0.5] * 3, [0.5] * 3)).to(dev)
        n = pts.shape[1]
        worst = [0.0, 0.0, 0.0, 0.0]
        for s in range(0, n, 65536):
            phr, plr, lhr, llr = native.query_points(pts[:, s:s + 65536].contiguous(), cal, 512, 200.0, Fl, Fh, blob, ws, want_logits=True)
            a, b = vh.view(-1)[s:s + 65536], vl.view(-1)[s:s + 65536]
            worst[0] = max(worst[0], (a - phr).abs().max().item())
            worst[1] = max(worst[1], (b - plr).abs().max().item())
            worst[2] = max(worst[2], (logit(a) - lhr.double()).abs().max().item())
            worst[3] = max(worst[3], (logit(b) - llr.double()).abs().max().item())
        print("R=%d  v5 vs fp32 point path: max|d occ| hr %.3e lr %.3e   max|d logit| hr %.3e lr %.3e  finite=%s" %
              (R, worst[0], worst[1], worst[2], worst[3], bool(torch.isfinite(vh).all() and torch.isfinite(vl).all())))
    R = Rt
    mat = oracle.coords_matrix(R, [-0.5] * 3, [0.5] * 3)[:3].reshape(-1)
    vh = torch.empty((R, R, R), dtype=torch.float32, device=dev)
    vl = torch.empty_like(vh)
    for prec, bl in (("fp32", blob), ("bf16", blob), ("fp16", g.blob("f16"))):
        f = lambda: native.query_grid(0, R, R, R, mat, cal, 512, 200.0, Fl, Fh, bl, prec, ws, vh, vl)
        f()
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        dt = (time.time() - t) / 2
        print("sweep %s R=%d: %.4f s  %.3e pts/s  (512^3 at this rate: %.3f s)" % (prec, R, dt, R ** 3 / dt, 512 ** 3 / (R ** 3 / dt)))


if __name__ == "__main__":
    main()
