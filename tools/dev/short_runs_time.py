"""Where do short runs stop paying?  50 000 points as runs of lo..hi points (the dirty lattice points of the reference's octree levels come
as ~ 25 per column) on the column kernels (surs_query_points_columns) against the layer kernels (surs_query_points)."""
import os, sys, time, numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common, gpu_common as g
from surs_amd import native
fl, fh = common.synth_features(hl=256, hh=1024)
Fl, Fh = g.upload_nhwc(fl), g.upload_nhwc(fh)
ws = native.Workspace(g.dev())
cal = common.CALIB.reshape(-1)[:12]
def pts_runs(lo, hi, n=50000, seed=1):
    rng = np.random.RandomState(seed); xs, ys, zs, tot = [], [], [], 0
    while tot < n:
        k = int(rng.randint(lo, hi + 1)); x, y = rng.uniform(-0.45, 0.45, 2)
        xs.append(np.full(k, x)); ys.append(np.full(k, y)); zs.append(np.sort(rng.uniform(-0.5, 0.5, k))); tot += k
    return np.stack([np.concatenate(xs), np.concatenate(ys), np.concatenate(zs)]).astype(np.float32)[:, :n]
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3, r
for prec in ("fp32", "bf16"):
    blob = g.blob("bf16")
    for lo, hi in ((4, 12), (8, 24), (12, 36), (16, 48), (24, 72)):
        p = torch.from_numpy(pts_runs(lo, hi)).to(g.dev())
        tc, rc = timed(lambda: native.query_points_columns(p, cal, 512, 200.0, Fl, Fh, blob, prec, ws))
        def layer():
            with native.reduced_point_operands(prec != "fp32"):
                return native.query_points(p, cal, 512, 200.0, Fl, Fh, blob, ws)
        tp, rp = timed(layer)
        print("%s runs of %d..%d points: column kernels %s, layer kernels %.3f ms" % (prec, lo, hi, "%.3f ms" % tc if rc is not None else "refused (%.3f ms)" % tc, tp))
