"""Times the reference itself (imported from /root/reference through tools/ref_harness.py) on this container's CPU cores:
the numbers of BASELINE.md section 2.  Build-container only (the reference does not travel to the GPU box).

    python tools/ref_time.py [query] [encoder] [recon] [mc]        -> prints one JSON object, also written to profiles/
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_harness as rh  # noqa: E402
import gen_golden as gg  # noqa: E402
from surs_amd import weights  # noqa: E402


def median_time(fn, n):
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return float(np.median(ts))


def main():
    what = sys.argv[1:] or ["query", "encoder", "recon", "mc"]
    torch.set_num_threads(8)
    out = {"cores": torch.get_num_threads(), "host": "build container, 8 CPUs", "torch": torch.__version__}
    net, opt_ref, sd = gg.make_net()
    calib = torch.from_numpy(gg.CALIB[None].copy())
    if "encoder" in what:
        img = torch.from_numpy(weights.synthetic_image(512, seed=1).copy())
        st = {}
        with torch.no_grad(), rh.quiet():
            net.super_res(img)   # warm-up
            t = time.perf_counter(); img_sr, f_lr, f_hr = net.super_res(img); st["super_res_s"] = time.perf_counter() - t
            t = time.perf_counter(); net.filter_hr(f_hr); st["filter_hr_s"] = time.perf_counter() - t
            t = time.perf_counter(); net.filter_lr(f_lr); st["filter_lr_s"] = time.perf_counter() - t
        st["total_s"] = sum(st.values())
        out["encoder_512x512"] = st
    else:
        fl, fh = gg.synth_features(seed=7, hl=256, hh=1024)
        net.im_feat_list_lr = [torch.from_numpy(fl[None].copy())]
        net.im_feat_list_hr = [torch.from_numpy(fh[None].copy())]
    if "query" in what:
        pts = torch.from_numpy(weights.synthetic_points(50000, seed=2)[None].copy())

        def q():
            with torch.no_grad(), rh.quiet():
                net.query_mr(pts, calib)
                net.query_sr(pts, calib)
                net.get_preds()
        q()
        s = median_time(q, 5)
        out["query_50k"] = {"seconds_median_of_5": s, "points_per_s": 50000 / s,
                            "what": "query_mr + query_sr + get_preds, 50 000 points, fp32, feature maps of the 512x512 input"}
    if "recon" in what:
        ns = rh.load_reference()
        b_min, b_max = np.array([-0.5] * 3), np.array([0.5] * 3)
        R = 128
        t = time.perf_counter()
        with torch.no_grad(), rh.quiet():
            r = ns.mesh_util.reconstruction(opt_ref, net, torch.device("cpu"), calib, R, b_min, b_max, use_octree=False, num_samples=50000)
        s = time.perf_counter() - t
        out["reconstruction_dense_r128"] = {"seconds": s, "queries_per_s": R ** 3 / s, "verts_hr": int(len(r[0])),
                                            "what": "lib.mesh_util.reconstruction(use_octree=False), R=128, incl. 2x skimage marching cubes "
                                                    "(bridged to the conda interpreter through .npy files)",
                                            "extrapolated_512_seconds": 512 ** 3 / (R ** 3 / s)}
    if "mc" in what:
        import mc_volumes
        mc = {}
        for n in (128, 256, 512):
            vol = mc_volumes.blob(n).astype(np.float32)
            t = time.perf_counter()
            v, f, _, _ = rh.skimage_mc(vol, 0.5)
            mc[str(n)] = {"seconds_incl_npy_bridge": time.perf_counter() - t, "verts": int(len(v)), "faces": int(len(f))}
        out["skimage_marching_cubes_lewiner_blob"] = mc
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", "r02_reference_cpu_times.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
