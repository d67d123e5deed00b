"""Multi-view dense sweep (num_views = 2, R = 128): the one-call library sweep against the reference's batch loop through the facade."""
import os, sys, time, numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from surs_amd import mesh_util, model, options
V, R = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 128
opt = options.BaseOptions().parse(common.FLAGS + ["--num_views", str(V)])
net = model.SuRSNet(opt, "orthogonal").to(device=torch.device("cuda:0"))
net.load_state_dict({k: torch.from_numpy(v) for k, v in common.state_dict().items()}); net.eval()
g = np.load(os.path.join(ROOT, "tests", "golden", "query_views.npz"))
fl = np.stack([common.synth_features(seed=10 + v, hl=256, hh=1024)[0] for v in range(V)])
fh = np.stack([common.synth_features(seed=10 + v, hl=256, hh=1024)[1] for v in range(V)])
net.im_feat_list_lr = [torch.from_numpy(fl).to("cuda:0")]; net.im_feat_list_hr = [torch.from_numpy(fh).to("cuda:0")]
calibs = torch.from_numpy(g["o2_calibs"].copy())
b_min, b_max = np.array([-0.5] * 3), np.array([0.5] * 3)
for name, kw in (("one call", {}), ("loop, 50 000 per call", dict(loop=True, num_samples=50000)), ("loop, 262 144 per call", dict(loop=True))):
    mesh_util.eval_volumes_views(opt, net, calibs, 32, b_min, b_max, **kw); torch.cuda.synchronize()
    t = time.perf_counter(); mesh_util.eval_volumes_views(opt, net, calibs, R, b_min, b_max, **kw); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("%-24s R=%d V=%d: %.3f s, %.3e voxels/s" % (name, R, V, dt, R ** 3 / dt))
