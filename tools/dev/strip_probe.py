import os, sys, numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from surs_amd import encoder, model, weights, native, dist as sdist
from surs_amd.model import _as_img
from surs_amd.sdf import create_grid
dev = torch.device("cuda:0")
net = model.SuRSNet(common.opt()).to(device=dev)
net.load_state_dict({k: torch.from_numpy(v) for k, v in common.state_dict().items()}); net.eval()
H, world, R = 512, 4, 128
img = torch.from_numpy(weights.synthetic_image(H, seed=1)).to(dev)
W = net._encoder_weights(); x = _as_img(img)
full = encoder.super_res(W, x, want_image=False)
hwc = lambda t: t.buf.view(t.h, t.w, t.c)
wl, wh = x.w // 2, 2 * x.w
b_min, b_max = np.array([-0.5] * 3), np.array([0.5] * 3)
_, mat = create_grid(R, R, R, b_min, b_max)
parts = []
share = wl // world
for rank in range(world):
    i0, i1 = sdist.slab_range(R, rank, world)
    need_hr = sdist.slab_feature_columns(i0, i1, mat[:3], common.CALIB.reshape(-1)[:12].astype(np.float64), wh)
    a, b = rank * share, (rank + 1) * share
    a, b = min(a, need_hr[0] // 4 // 2 * 2), max(b, -(-need_hr[1] // 4) + (-(-need_hr[1] // 4)) % 2)
    b = min(b, wl)
    _, new2, new_fin = encoder.super_res_strip(W, x, a, b, want_image=False)
    ok2 = torch.equal(hwc(new2), hwc(full[1])[:, a:b, :]); okf = torch.equal(hwc(new_fin), hwc(full[2])[:, 4 * a:4 * b, :])
    print("rank %d strip [%d, %d): feature_lr %s, feature_hr %s" % (rank, a, b, ok2, okf))
    parts.append(hwc(new2)[:, rank * share - a:(rank + 1) * share - a, :].contiguous())
f_lr = torch.cat(parts, dim=1).contiguous()
print("gathered == full:", torch.equal(f_lr, hwc(full[1])))
flr = native.Img(full[1].h, wl, full[1].c, buf=f_lr.reshape(-1), device=dev)
o1 = encoder.filter_lr(W, flr)[-1]
o2 = encoder.filter_lr(W, full[1])[-1]
o3 = encoder.filter_lr(W, full[1])[-1]
print("filter_lr(gathered) == filter_lr(full):", torch.equal(hwc(o1), hwc(o2)), " filter_lr(full) twice:", torch.equal(hwc(o2), hwc(o3)), float((hwc(o1) - hwc(o2)).abs().max()))
# ---- closer to encode_sharded: a second network object, the persistent gather buffer, host-staged parts
from surs_amd import options
opt = options.BaseOptions().parse(common.FLAGS + ["--precision", "fp32"])
def make():
    n = model.SuRSNet(opt).to(device=dev); n.load_state_dict({k: torch.from_numpy(v) for k, v in common.state_dict().items()}); return n.eval()
rep, shd = make(), make()
_, f_lr_r, f_hr_r = rep.super_res(img); rep.filter_hr(f_hr_r); rep.filter_lr(f_lr_r)
W2 = shd._encoder_weights()
parts2 = [p.cpu().to(dev) for p in parts]
buf = torch.empty((full[1].h, wl, full[1].c), dtype=torch.float32, device=dev)
torch.cat(parts2, dim=1, out=buf)
fl2 = native.Img(full[1].h, wl, full[1].c, buf=buf.reshape(-1), device=dev)
o4 = encoder.filter_lr(W2, fl2)[-1]
from surs_amd.model import _as_nchw_view
print("second net, persistent buffer: equal to rep.filter_lr:", torch.equal(_as_nchw_view(o4), rep.im_feat_list_lr[-1]),
      float((_as_nchw_view(o4) - rep.im_feat_list_lr[-1]).abs().max()), " input equal:", torch.equal(buf.permute(2, 0, 1), rep.feature_lr[0]))
