"""Per-call table of the encoder (super_res + filter_hr + filter_lr) on a synthetic H x H image: every native.* call with its
shapes, its time (HIP events around the one call, synchronised: for reading, not for summing) and, for convolutions, the
arithmetic rate of the fp32-grade product (3 f16 MFMA products per multiply-add).  python tools/enc_layers.py [H]"""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from surs_amd import model, native, weights
H = int(sys.argv[1]) if len(sys.argv) > 1 else 512
net = model.SuRSNet(common.opt()).to(device=torch.device("cuda:0"))
net.load_state_dict({k: torch.from_numpy(v) for k, v in common.state_dict().items()})
net.eval()
img = torch.from_numpy(weights.synthetic_image(H, seed=1)).to("cuda:0")
def run():
    _, f_lr, f_hr = net.super_res(img)
    net.filter_hr(f_hr); net.filter_lr(f_lr)
run(); run(); torch.cuda.synchronize()
rows = []
def wrap(name):
    fn = getattr(native, name)
    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        r = fn(*a, **k)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        x = a[0]
        if name == "conv2d":
            cw, s = a[1], k.get("stride", 1)
            ho, wo = r.h, r.w
            fl = 2.0 * ho * wo * cw.k * cw.k * cw.cin * cw.cout
            rows.append((name, "%dx%d %d->%d k%d s%d%s" % (x.h, x.w, cw.cin, cw.cout, cw.k, s, " gn" if k.get("in_scale") is not None else ""), us, fl,
                         4.0 * (x.h * x.w * cw.cin + ho * wo * cw.cout)))
        else:
            rows.append((name, "%dx%d c%d" % (x.h, x.w, x.c), us, 0.0, 4.0 * x.h * x.w * x.c))
        return r
    setattr(native, name, w)
for n in ("conv2d", "groupnorm_coeffs", "avgpool2", "bicubic_up2", "pixel_shuffle2", "add3"):
    wrap(n)
run()
tot = sum(r[2] for r in rows)
agg = {}
for name, desc, us, fl, by in rows:
    k = (name, desc)
    a = agg.setdefault(k, [0, 0.0, fl, by])
    a[0] += 1; a[1] += us
print("%-18s %-34s %5s %9s %9s %8s %8s" % ("call", "shape", "n", "us each", "us total", "TF/s x3", "GB/s"))
for (name, desc), (n, us, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-18s %-34s %5d %9.1f %9.1f %8.1f %8.0f" % (name, desc, n, us / n, us, 3 * fl * n / us * 1e-6 if fl else 0.0, by * n / us * 1e-3))
print("sum of calls: %.1f us (%d calls); conv flops %.1f GF (x3 products: %.1f)" % (tot, len(rows), sum(r[3] for r in rows) * 1e-9, 3e-9 * sum(r[3] for r in rows)))
