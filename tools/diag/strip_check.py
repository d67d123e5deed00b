import os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from surs_amd import encoder, model, weights
from surs_amd.model import _as_img
dev = torch.device("cuda:0")
net = model.SuRSNet(common.opt()).to(device=dev)
net.load_state_dict({k: torch.from_numpy(v) for k, v in common.state_dict().items()})
net.eval()
for H in (256, 512):
    img = torch.from_numpy(weights.synthetic_image(H, seed=1)).to(dev)
    W = net._encoder_weights()
    x = _as_img(img)
    full = encoder.super_res(W, x)
    hwc = lambda t: t.buf.view(t.h, t.w, t.c)
    wl = H // 2
    for a, b in ((0, wl // 2), (wl // 2, wl), (wl // 4, wl // 2), (0, wl)):
        got = encoder.super_res_strip(W, x, a, b)
        for name, g, f, sc in zip(("img_sr", "new2", "new_fin"), got, full, (4, 1, 4)):
            d = (hwc(g) - hwc(f)[:, sc * a:sc * b, :]).abs()
            bad = (d > 0).any(dim=0).any(dim=1)   # per column
            cols = torch.nonzero(bad).flatten().cpu().numpy()
            print("H %d strip [%d,%d) %-8s max|diff| %.3e, differing columns: %s" % (H, a, b, name, d.max().item(),
                  ("none" if len(cols) == 0 else "%d..%d (%d)" % (cols.min(), cols.max(), len(cols)))), flush=True)
        fh_full = encoder.filter_hr(W, full[2])[0]
        fh_strip = encoder.filter_hr(W, got[2])[0]
        d = (hwc(fh_strip) - hwc(fh_full)[:, 4 * a:4 * b, :]).abs()
        print("   filter_hr strip vs full: max|diff| %.3e" % d.max().item(), flush=True)
