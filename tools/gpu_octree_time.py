"""Times reconstruction(use_octree=True) (gen_mesh's default, lib/sdf.py:55-120) against the dense sweep on the smooth body field.

    python tools/gpu_octree_time.py [R] [precision]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import precision_report as pr  # noqa: E402
from surs_amd import mesh_util, model, native, options, weights  # noqa: E402


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
    dev = native.require_gpu()
    opt = options.BaseOptions().parse(pr.FLAGS + ["--precision", prec] + (["--octree_precision", "sweep"] if os.environ.get("SURS_OCT_SWEEP") else []))
    net = model.SuRSNet(opt).to(device=dev)
    full = weights.synthetic_state_dict(opt, seed=0)
    full.update(weights.body_state_dict(opt))
    net.load_state_dict(full)
    net.eval()
    fl, fh = weights.body_features(256, 1024)
    feats = (pr._upload(fl, dev), pr._upload(fh, dev))
    calib = torch.from_numpy(pr.CALIB).to(dev)[None]
    for use_oct in ((True,) if os.environ.get("OCTREE_ONLY") else (False, True)):
        for rep in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            out = mesh_util.reconstruction(opt, net, dev, calib, R, np.array([-0.5] * 3), np.array([0.5] * 3), use_octree=use_oct,
                                           features=feats)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
        print("%s R=%d %s: %.3f s, %d / %d vertices" % ("octree" if use_oct else "dense ", R, prec, dt, len(out[0]), len(out[4])), flush=True)


if __name__ == "__main__":
    main()
