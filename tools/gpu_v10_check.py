"""The eight-wave column kernels against their four-wave forms (the same restated arithmetic: v10 vs v7 in bf16 / fp16, v11 vs
v8 fp32-grade): bitwise comparison on small and full-size grids (noise field; R = 24 runs many chunks per tile) and sweep
times on the same device.

    python tools/gpu_v10_check.py [R_big] [kernels, e.g. 7,10 | 8,11] [precisions, e.g. bf16,fp16 | fp32]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import precision_report as pr  # noqa: E402
from surs_amd import native  # noqa: E402


def main():
    Rb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    kernels = [int(k) for k in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["7", "10"])]
    precs = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ("bf16", "fp16")
    dev = native.require_gpu()
    sd, Fl, Fh, keep = pr.noise_inputs(dev)
    for R in (24, 40, 136, Rb):
        res = {}
        for kv in kernels:
            vols, times, _ = pr.sweeps(sd, Fl, Fh, R, precs, dev, kernel=kv)
            res[kv] = vols
            print("R=%d kernel v%d sweep seconds: %s" % (R, kv, {k: round(v, 4) for k, v in times.items()}), flush=True)
        for prec in precs:
            for i, tag in enumerate(("hr", "lr")):
                a, b = res[kernels[0]][prec][i], res[kernels[-1]][prec][i]
                d = (a - b).abs().max().item()
                print("R=%d %s %s: v%d vs v%d max|diff| %.3e equal=%d finite=%d" % (R, prec, tag, kernels[-1], kernels[0], d, int(torch.equal(a, b)),
                                                                                    int(bool(torch.isfinite(b).all()))), flush=True)
        del res


if __name__ == "__main__":
    main()
