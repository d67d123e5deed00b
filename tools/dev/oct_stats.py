import os, sys, numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import precision_report as pr
from surs_amd import mesh_util, model, native, options, weights, sdf
dev = native.require_gpu()
opt = options.BaseOptions().parse(pr.FLAGS + ["--precision", "fp32"])
net = model.SuRSNet(opt).to(device=dev)
full = weights.synthetic_state_dict(opt, seed=0); full.update(weights.body_state_dict(opt))
net.load_state_dict(full); net.eval()
fl, fh = weights.body_features(256, 1024)
feats = (pr._upload(fl, dev), pr._upload(fh, dev))
R = 512
mat = sdf.create_grid(R, R, R, np.array([-0.5] * 3), np.array([0.5] * 3))[1]
stats = []
zmul, zdiv = net._zscale()
cal = pr.CALIB.reshape(-1)[:12]
native.octree_volumes(R, mat[:3].reshape(-1), cal, zmul, zdiv, feats[0], feats[1], net._mlp_blob(), net._workspace(), 0.05, stats=stats)
for s in stats:
    print("reso %d: dirty points %d, columns %d, 64-point tiles %d, points per column %.1f, tiles per column %.2f, fill %.2f" % (s[0], s[1], s[2], s[3], s[1] / max(s[2], 1), s[3] / max(s[2], 1), s[1] / max(64 * s[3], 1)))
