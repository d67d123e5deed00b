"""Column kernel v7 (affine restatement of layer 1 + residuals of the listed channels) against v3 and the fp32-grade sweep
on the same features and weights: logit-space differences and sweep times.

    python tools/gpu_v7_check.py [R] [noise|body]
"""
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import precision_report as pr  # noqa: E402
from surs_amd import _lib, native  # noqa: E402


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    which = sys.argv[2] if len(sys.argv) > 2 else "noise"
    dev = native.require_gpu()
    if which == "body":
        sd, Fl, Fh = pr.body_inputs(dev)
    else:
        sd, Fl, Fh, keep = pr.noise_inputs(dev)
    L = _lib.lib()
    out = {}
    for kv in (3, 7):
        L.surs_set_grid_kernel(kv)
        vols, times, ws = pr.sweeps(sd, Fl, Fh, R, ("bf16", "fp16"), dev)
        out[kv] = (vols, times)
        print("kernel v%d sweep seconds at R=%d: %s" % (kv, R, {k: round(v, 4) for k, v in times.items()}), flush=True)
    L.surs_set_grid_kernel(5)
    v5, t5, _ = pr.sweeps(sd, Fl, Fh, R, ("fp32",), dev)
    L.surs_set_grid_kernel(8)
    v8, t8, _ = pr.sweeps(sd, Fl, Fh, R, ("fp32",), dev)
    L.surs_set_grid_kernel(0)
    print("fp32-grade sweep seconds at R=%d: v5 %.4f  v8 %.4f" % (R, t5["fp32"], t8["fp32"]))
    for i, tag in enumerate(("hr", "lr")):
        st = pr.field_stats(v8["fp32"][i], v5["fp32"][i])
        print("v8 vs v5 fp32 %s: max|dlogit| %.3e mean %.3e flips %d finite %s" % (
            tag, st["max_abs_dlogit"], st["mean_abs_dlogit"], st["flipped_voxels"], bool(torch.isfinite(v8["fp32"][i]).all())))
    ref = v5["fp32"]
    for prec in ("bf16", "fp16"):
        for kv in (3, 7):
            v = out[kv][0][prec]
            for i, tag in enumerate(("hr", "lr")):
                st = pr.field_stats(v[i], ref[i])
                print("v%d %s %s vs fp32: max|dlogit| %.4f mean %.5f flips %d finite %s" % (
                    kv, prec, tag, st["max_abs_dlogit"], st["mean_abs_dlogit"], st["flipped_voxels"], bool(torch.isfinite(v[i]).all())))
        for i, tag in enumerate(("hr", "lr")):
            st = pr.field_stats(out[7][0][prec][i], out[3][0][prec][i])
            print("v7 vs v3 %s %s: max|dlogit| %.4f mean %.5f flips %d" % (prec, tag, st["max_abs_dlogit"], st["mean_abs_dlogit"], st["flipped_voxels"]))


if __name__ == "__main__":
    main()
