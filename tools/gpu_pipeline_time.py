"""Subjects per second of a run of subjects: gen_mesh per subject (host normalise + upload, encoder, reconstruction)
against train_util.gen_mesh_pipelined (device input stage, next subject's decode / upload / encoder under the current sweep).
Decoded 8-bit pixels are held in memory on both sides (file decoding is host work outside this package).

    python tools/gpu_pipeline_time.py [R] [K] [precision]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from surs_amd import data, mesh_util, model, options, train_util, weights  # noqa: E402


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    dev = torch.device("cuda:0")
    opt = options.BaseOptions().parse(["--loadSize", "1024", "--residual", "--b_min", "-0.5", "-0.5", "-0.5", "--b_max", "0.5", "0.5", "0.5",
                                       "--resolution", str(R), "--precision", prec])
    net = model.SuRSNet(opt).to(device=dev)
    net.load_state_dict(weights.synthetic_state_dict(opt, seed=0))
    net.eval()
    ds = data.SyntheticDataset(opt, n=K, size=512)
    raws = [ds.get_raw_item(i) for i in range(K)]
    calib = train_util.gen_calib().to(dev)

    class Held:
        def get_raw_item(self, i):
            return raws[i]

    def sequential():
        for raw in raws:
            m = raw["mask"].astype(np.float32) / np.float32(255.0)
            x = (raw["rgb"].astype(np.float32) / np.float32(255.0) - np.float32(0.5)) / np.float32(0.5)
            img = torch.from_numpy(np.ascontiguousarray((m[None] * x.transpose(2, 0, 1))[None])).to(dev)
            _, f_lr, f_hr = net.super_res(img)
            net.filter_hr(f_hr)
            net.filter_lr(f_lr)
            mesh_util.reconstruction(opt, net, dev, calib, R, raw["b_min"], raw["b_max"], use_octree=False, want_normals=False)

    def pipelined():
        train_util.gen_mesh_pipelined(opt, net, dev, Held(), range(K), None, use_octree=False, write=False)

    for name, fn in (("sequential", sequential), ("pipelined", pipelined), ("sequential", sequential), ("pipelined", pipelined)):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print("%-10s R=%d %s: %d subjects in %.3f s = %.2f subjects/s (%.1f ms each)" % (name, R, prec, K, dt, K / dt, dt / K * 1e3), flush=True)


if __name__ == "__main__":
    main()
