"""Where do the milliseconds of one 50 000-point facade call go?  cProfile of the reference's eval_func around SuRSNet."""
import cProfile, pstats, io, os, sys, time
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from surs_amd import model, weights
dev = torch.device("cuda:0")
net = model.SuRSNet(common.opt()).to(device=dev)
net.load_state_dict({k: torch.from_numpy(v) for k, v in common.state_dict().items()})
net.eval()
_, f_lr, f_hr = net.super_res(torch.from_numpy(weights.synthetic_image(512, seed=1)).to(dev))
net.filter_hr(f_hr); net.filter_lr(f_lr)
calib = torch.from_numpy(common.CALIB[None]).to(dev)
pts_all = np.random.RandomState(0).uniform(-0.5, 0.5, (3, 20 * 50000))

def eval_func(points):
    points = np.expand_dims(points, axis=0)
    samples = torch.from_numpy(points).to(device=dev).float()
    net.query_mr(samples, calib)
    net.query_sr(samples, calib)
    return net.get_preds()[0][0].detach().cpu().numpy()

def loop():
    for i in range(20):
        eval_func(pts_all[:, i * 50000:(i + 1) * 50000])

loop(); torch.cuda.synchronize()
t = time.perf_counter(); loop(); torch.cuda.synchronize(); print("ms per chunk: %.2f" % ((time.perf_counter() - t) / 20 * 1e3))
pr = cProfile.Profile(); pr.enable(); loop(); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])

# ---- where is the time: device or host?
from surs_amd import native
samples = torch.from_numpy(pts_all[:, :50000].copy()).to(dev).float()
cal = common.CALIB.reshape(-1)[:12]
fl, fh = net.features()
ws, blob = net._workspace(), net._mlp_blob()
for name, fn in (("native.query_points", lambda: native.query_points(samples, cal, 512, 200.0, fl, fh, blob, ws)),
                 ("isfinite+all", lambda: bool(torch.isfinite(samples).all())),
                 ("stack", lambda: torch.stack([samples[0]])),
                 ("query_mr", lambda: net.query_mr(samples[None], calib)),
                 ("query_mr+sr+preds.cpu", lambda: (net.query_mr(samples[None], calib), net.query_sr(samples[None], calib), net.get_preds()[0][0].cpu()))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = time.perf_counter(); e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-28s host+device %.3f ms per call, device span %.3f ms" % (name, (time.perf_counter() - t) / 20 * 1e3, e0.elapsed_time(e1) / 20))
