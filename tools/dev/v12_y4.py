"""Diagnostic for the y4 debug build: vol_lr = y4 from lanes n, vol_hr = y4 from lanes n + 32 (hr item)."""
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
vh = torch.empty((64, R, R), dtype=torch.float32, device=dev); vl = torch.empty_like(vh)
native.query_grid(0, 64, R, R, m[:3].reshape(-1), cal, 512, 200.0, Fl, Fh, blob, "bf16", ws, vh, vl, kernel=12)
torch.cuda.synchronize()
a = vl.cpu().numpy(); b = vh.cpu().numpy()   # values are cmask * sigmoid(y4): invert
def inv(x):
    x = np.clip(x.astype(np.float64), 1e-12, 1 - 1e-12)
    return np.log(x / (1 - x))
bad = np.argwhere(a != b)
print("differing", len(bad))
seen = set()
for i, j, k in bad:
    t = (i, j, k // 128)
    if t in seen: continue
    seen.add(t)
    if len(seen) > 6: break
    z = k // 128; k0 = z * 128 + (k % 128) // 32 * 32
    print("tile", t, "col tile", (k % 128) // 32)
    print("  y4 lanes n    :", np.round(inv(a[i, j, k0 + 14:k0 + 20]), 5))
    print("  y4 lanes n+32 :", np.round(inv(b[i, j, k0 + 14:k0 + 20]), 5))
    for dz in (-1, 1):
        if 0 <= z + dz < 4:
            kk = k0 + dz * 128
            print("  same column tile %+d lanes n:" % dz, np.round(inv(a[i, j, kk + 14:kk + 20]), 5))
    # search: does the wrong value appear as a correct value anywhere in the neighbourhood of columns?
    target = b[i, j, k0 + 16]
    hits = np.argwhere(a[:, :, (k0 % 128) + 16::128] == target)
    print("  wrong value of voxel 16 found as a correct lane-n value at (i, j, tile):", hits[:8].tolist())

# which p did the wrong lanes read?  (wrong - right) = w4p (p' - p): test candidates p' = p_lr at other voxels
vh10 = torch.empty((64, R, R), dtype=torch.float32, device=dev); vl10 = torch.empty_like(vh10)
native.query_grid(0, 64, R, R, m[:3].reshape(-1), cal, 512, 200.0, Fl, Fh, blob, "bf16", ws, vh10, vl10, kernel=10)
torch.cuda.synchronize()
plr = vl10.cpu().numpy().astype(np.float64)
ya, yb = inv(a), inv(b)
sel = bad[::97][:400]
for name, shift in (("k-128", -128), ("k+128", 128), ("k+32", 32), ("k-32", -32), ("k+16", 16), ("k-16", -16), ("k+64", 64), ("k-64", -64)):
    r = []
    for i, j, k in sel:
        kk = k + shift
        if 0 <= kk < R:
            dp = plr[i, j, kk] - plr[i, j, k]
            if abs(dp) > 1e-6:
                r.append((yb[i, j, k] - ya[i, j, k]) / dp)
    r = np.array(r)
    if len(r): print(name, "n", len(r), "ratio median %.4f  p10 %.4f p90 %.4f" % (np.median(r), np.percentile(r, 10), np.percentile(r, 90)))
w = sd["mlp_hr.conv4.weight"] if "mlp_hr.conv4.weight" in sd else None
if w is not None:
    w = np.asarray(w.cpu() if torch.is_tensor(w) else w).reshape(-1)
    print("w4 hr tail entries:", w[-4:], "len", len(w))
w4p = float(w[-1])
pp = []
for i, j, k in bad[::53][:2000]:
    pp.append(plr[i, j, k] + (yb[i, j, k] - ya[i, j, k]) / w4p)
pp = np.array(pp)
print("implied p' of the wrong lanes: min %.4f median %.4f max %.4f; |p'| < 0.01: %d of %d" % (pp.min(), np.median(pp), pp.max(), int((np.abs(pp) < 0.01).sum()), len(pp)))
