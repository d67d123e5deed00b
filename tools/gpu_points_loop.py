"""The reference's 50 000-point chunk loop (lib/sdf.py:32-45 around lib/mesh_util.py:20-28) on the facade, for profiling:
    python tools/gpu_points_loop.py [precision] [chunks] [grid|random]   (rocprofv3 --kernel-trace --stats -- python3 tools/gpu_points_loop.py bf16)
grid (default): consecutive points of a 512^3 grid from its middle, as the reference's loop passes them (runs of 512 points with one
image position: the facade takes them through the column kernels); random: uniform samples (the point kernels)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from surs_amd import model, options, weights, train_util
if os.environ.get("LOOP_TORCH_FINITE_CHECK") == "1":   # (A/B of surs_nonfinite against the two torch reductions it replaced, NOTES R6.8)
    from surs_amd import native
    native.any_nonfinite = lambda a, b: not bool(torch.isfinite(a.sum() + b.sum()).item())
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
net = model.SuRSNet(options.BaseOptions().parse(common.FLAGS + ["--precision", prec])).to(device=dev)
net.load_state_dict({k: torch.from_numpy(v) for k, v in common.state_dict().items()})
net.eval()
fl, fh = common.synth_features(hl=256, hh=1024)
net.im_feat_list_lr = [torch.from_numpy(fl[None]).to(dev)]
net.im_feat_list_hr = [torch.from_numpy(fh[None]).to(dev)]
calib = train_util.gen_calib().to(dev)
kind = sys.argv[3] if len(sys.argv) > 3 else "grid"
if kind == "grid":
    from surs_amd import sdf
    R = 512
    mat4 = sdf.create_grid(R, R, R, np.array([-0.5] * 3), np.array([0.5] * 3))[1]
    idx = np.arange(nch * 50000, dtype=np.int64) + (R // 2) * R * R - nch * 50000 // 2
    ijk = np.stack([idx // (R * R), (idx // R) % R, idx % R]).astype(np.float64)
    pts = np.matmul(mat4[:3, :3], ijk) + mat4[:3, 3:4]
else:
    pts = weights.synthetic_points(50000 * nch, seed=2).astype(np.float64)
def chunk(i):
    p = np.repeat(np.expand_dims(pts[:, i * 50000:(i + 1) * 50000], 0), 1, axis=0)
    s = torch.from_numpy(p).to(device=dev).float()
    net.query_mr(s, calib); net.query_sr(s, calib)
    return net.get_preds()[0][0].detach().cpu().numpy()
for i in range(3): chunk(i)
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(nch): chunk(i)
torch.cuda.synchronize(); dt = time.perf_counter() - t
print("%s, %s points: %.3f ms per 50k chunk, %.3e queries/s" % (prec, kind, dt / nch * 1e3, 50000 * nch / dt))
