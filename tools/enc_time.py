"""Times the image encoder (super_res + filter_hr + filter_lr) on a synthetic 512x512 image; prints a digest of the outputs
so that tile-configuration changes can be checked for bit neutrality."""
import hashlib, os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from surs_amd import model, weights
H = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ENC = sys.argv[2] if len(sys.argv) > 2 else "fp32"      # --encoder_precision
o = common.opt()
o.encoder_precision = ENC
net = model.SuRSNet(o).to(device=torch.device("cuda:0"))
net.load_state_dict({k: torch.from_numpy(v) for k, v in common.state_dict().items()})
net.eval()
img = torch.from_numpy(weights.synthetic_image(H, seed=1)).to("cuda:0")
def run():
    _, f_lr, f_hr = net.super_res(img)
    net.filter_hr(f_hr); net.filter_lr(f_lr)
    return f_lr, f_hr
run(); torch.cuda.synchronize()
t = time.time()
for _ in range(5): out = run()
torch.cuda.synchronize()
dt = (time.time() - t) / 5
h = hashlib.sha256()
for x in (out[0], out[1], net.im_feat_list_lr[-1], net.im_feat_list_hr[0]):
    h.update(x.contiguous().cpu().numpy().tobytes())
print("encoder H=%d (%s): %.3f ms  digest %s" % (H, ENC, dt * 1e3, h.hexdigest()[:16]))
# the three stages alone (a synchronisation between them: their sum exceeds the figure above by the drained pipelines)
def stage(name, fn, n=10):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): r = fn()
    torch.cuda.synchronize()
    print("  %-10s %.3f ms" % (name, (time.time() - t) / n * 1e3))
    return r
_, f_lr, f_hr = stage("super_res", lambda: net.super_res(img))
stage("filter_hr", lambda: net.filter_hr(f_hr))
stage("filter_lr", lambda: net.filter_lr(f_lr))
