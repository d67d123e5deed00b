"""Relative error (to each tensor's range) of the encoder outputs against the reference goldens, for both conv paths."""
import os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from surs_amd import model, weights
for H in (64, 96):
    g = np.load(os.path.join(ROOT, "tests", "golden", "encoder_h%d.npz" % H))
    net = model.SuRSNet(common.opt()).to(device=torch.device("cuda:0"))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in common.state_dict().items()})
    net.eval()
    img = torch.from_numpy(weights.synthetic_image(H, seed=1)).to("cuda:0")
    _, f_lr, f_hr = net.super_res(img)
    net.filter_hr(f_hr); net.filter_lr(f_lr)
    im_lr = net.im_feat_list_lr[0][0].cpu().numpy(); f_lr = f_lr[0].cpu().numpy()
    s2 = lambda a: a[..., ::2, ::2]
    print("H=%d SURS_CONV_X3=%s: feature_lr %.2e  im_feat_lr %.2e" % (H, os.environ.get("SURS_CONV_X3", "1"),
          common.rel_err(f_lr if H == 64 else s2(f_lr), g["feature_lr"]), common.rel_err(im_lr if H == 64 else s2(im_lr), g["im_feat_lr"])))
