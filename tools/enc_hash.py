"""Prints a digest of the encoder outputs for a synthetic image (used to check that kernel re-tilings are bit-neutral)."""
import hashlib, os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from surs_amd import model, weights
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
net = model.SuRSNet(common.opt()).to(device=torch.device("cuda:0"))
net.load_state_dict({k: torch.from_numpy(v) for k, v in common.state_dict().items()})
net.eval()
img = torch.from_numpy(weights.synthetic_image(H, seed=1)).to("cuda:0")
_, f_lr, f_hr = net.super_res(img)
net.filter_hr(f_hr); net.filter_lr(f_lr)
h = hashlib.sha256()
for t in (f_lr, f_hr, net.im_feat_list_lr[-1], net.im_feat_list_hr[0]):
    h.update(t.contiguous().cpu().numpy().tobytes())
print("encoder digest H=%d" % H, h.hexdigest())
