"""Where do v12's volumes differ from v10's? (noise field, bf16, a 64-plane slab of the 512^3 grid)"""
import os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import precision_report as pr
from surs_amd import native
dev = native.require_gpu()
sd, Fl, Fh, keep = pr.noise_inputs(dev)
R = 512
mlp = {k: v for k, v in sd.items() if k.startswith("mlp_")}
cal = pr.CALIB.reshape(-1)[:12]
m = np.eye(4); m[0, 0] = m[1, 1] = m[2, 2] = 1.0 / R; m[:3, 3] = -0.5
ws = native.Workspace(dev)
blob, _ = native.pack_mlp(mlp, "bf16", dev)
res = {}
for kv in (10, 12):
    vh = torch.empty((64, R, R), dtype=torch.float32, device=dev); vl = torch.empty_like(vh)
    native.query_grid(0, 64, R, R, m[:3].reshape(-1), cal, 512, 200.0, Fl, Fh, blob, "bf16", ws, vh, vl, kernel=kv)
    torch.cuda.synchronize()
    res[kv] = (vh.cpu().numpy(), vl.cpu().numpy())
if os.environ.get("V12_HALVES"):
    print("v12 halves: lanes n vs n+32 differ at", int((res[12][0] != res[12][1]).sum()), "voxels; lr-lane copy vs v10 hr:", int((res[12][1] != res[10][0]).sum()))
a, b = res[10][0], res[12][0]
bad = np.argwhere(a != b)
print("differing hr voxels", len(bad), "lr", int((res[10][1] != res[12][1]).sum()))
if len(bad):
    k = bad[:, 2]
    print("by tile (k // 128):", np.bincount(k // 128, minlength=4))
    print("by column tile ((k % 128) // 32):", np.bincount((k % 128) // 32, minlength=4))
    print("by lane (k % 32):", np.bincount(k % 32, minlength=32))
    cols = bad[:, 0] * R + bad[:, 1]
    uc, cnt = np.unique(cols * 4 + k // 128, return_counts=True)
    print("tiles touched", len(uc), "voxels per touched tile: min", cnt.min(), "max", cnt.max(), "hist", np.bincount(cnt)[:130].nonzero()[0][:20])
    for i, j, kk in bad[:12]:
        print((i, j, kk), a[i, j, kk], b[i, j, kk], "lr", res[10][1][i, j, kk])
    # are whole tiles shifted? compare tile of first bad voxel
    i, j, kk = bad[0]
    t = kk // 128
    if os.environ.get("V12_DBG"):
        ksv = res[12][1]
        kst = ksv[bad[:, 0], bad[:, 1], bad[:, 2]]
        print("ks (lr+hr) of differing voxels' tiles:", np.unique(kst, return_counts=True))
        print("ks distribution over all tiles:", np.unique(ksv[:, :, ::128], return_counts=True))
    print("v10 tile:", a[i, j, t * 128:(t + 1) * 128][:16])
    print("v12 tile:", b[i, j, t * 128:(t + 1) * 128][:16])
    print("lr  tile:", res[10][1][i, j, t * 128:(t + 1) * 128][:16])
