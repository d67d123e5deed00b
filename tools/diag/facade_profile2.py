import os, sys, time
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

def timeit(name, fn, n=20):
    fn(0); torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    print("%-60s %.3f ms per chunk" % (name, (time.perf_counter() - t) / n * 1e3), flush=True)

sl = lambda i: pts_all[:, i * 50000:(i + 1) * 50000]
def full(i, prep):
    samples = prep(i)
    net.query_mr(samples, calib); net.query_sr(samples, calib)
    return net.get_preds()[0][0].detach().cpu().numpy()
timeit("reference eval_func (expand_dims, np.repeat, .to(dev).float())", lambda i: full(i, lambda i: torch.from_numpy(np.repeat(np.expand_dims(sl(i), 0), 1, axis=0)).to(device=dev).float()))
timeit("WITHOUT np.repeat (f64 strided view -> .to(dev).float())", lambda i: full(i, lambda i: torch.from_numpy(np.expand_dims(sl(i), 0)).to(device=dev).float()))
timeit("contiguous f64 -> .to(dev).float()", lambda i: full(i, lambda i: torch.from_numpy(np.ascontiguousarray(sl(i))[None]).to(device=dev).float()))
timeit("f32 host -> .to(dev)", lambda i: full(i, lambda i: torch.from_numpy(np.ascontiguousarray(sl(i), np.float32)[None]).to(device=dev)))
timeit("H2D only: strided f64 .to(dev).float()", lambda i: torch.from_numpy(np.expand_dims(sl(i), 0)).to(device=dev).float())
timeit("H2D only: contiguous f64 .to(dev)", lambda i: torch.from_numpy(np.ascontiguousarray(sl(i))[None]).to(device=dev))
dsamples = torch.from_numpy(np.ascontiguousarray(sl(0), np.float32)[None]).to(dev)
timeit("device samples: mr + sr + preds.cpu().numpy()", lambda i: full(i, lambda i: dsamples))
timeit("device samples + a 1.2 MB pageable H2D before", lambda i: (torch.from_numpy(np.ascontiguousarray(sl(i))).to(dev), full(i, lambda i: dsamples)))
os.environ["X"] = "1"
