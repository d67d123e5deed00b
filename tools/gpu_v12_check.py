"""Column kernel v12 (two workgroups per CU, layer 1 streamed into layer 2) against v10 on one GPU: volumes bit for bit on the
noise / body / gain-60 fields at 512^3 and on small ragged grids, and the sweep times of both.

    python tools/gpu_v12_check.py [fields ...]        # default: small noise body
"""
import os, sys, time, json
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import precision_report as pr
from surs_amd import native

dev = native.require_gpu()
fields = sys.argv[1:] or ["small", "noise", "body"]
out = {}


def cmp(tag, sd, Fl, Fh, R, precs):
    for prec in precs:
        a, ta, _ = pr.sweeps(sd, Fl, Fh, R, (prec,), dev, kernel=10)
        b, tb, _ = pr.sweeps(sd, Fl, Fh, R, (prec,), dev, kernel=12)
        same = [bool(torch.equal(a[prec][i], b[prec][i])) for i in range(2)]
        d = [float((a[prec][i] - b[prec][i]).abs().max()) for i in range(2)]
        nbad = [int((a[prec][i] != b[prec][i]).sum()) for i in range(2)]
        fin = bool(torch.isfinite(b[prec][0]).all() and torch.isfinite(b[prec][1]).all())
        # repeat for timing (second run of each: warm)
        _, ta2, _ = pr.sweeps(sd, Fl, Fh, R, (prec,), dev, kernel=10)
        _, tb2, _ = pr.sweeps(sd, Fl, Fh, R, (prec,), dev, kernel=12)
        r = {"bitwise_hr_lr": same, "max_abs_diff": d, "differing_voxels": nbad, "finite": fin,
             "v10_s": [ta[prec], ta2[prec]], "v12_s": [tb[prec], tb2[prec]]}
        out["%s_R%d_%s" % (tag, R, prec)] = r
        print(tag, R, prec, r, flush=True)
        del a, b


if "small" in fields:
    import common, gpu_common as g
    fl, fh = common.synth_features()
    sd = common.state_dict()
    Fl, Fh = g.upload_nhwc(fl), g.upload_nhwc(fh)
    for R in (40, 136):
        cmp("small", sd, Fl, Fh, R, ("bf16", "fp16"))
if "noise" in fields or "gain60" in fields or "gain16" in fields:
    sd, Fl, Fh, keep = pr.noise_inputs(dev)
    if "noise" in fields:
        cmp("noise", sd, Fl, Fh, 512, ("bf16", "fp16"))
    for gname, gain in (("gain16", 16.0), ("gain60", 60.0)):
        if gname in fields:
            sd2 = {k: (v.clone() if torch.is_tensor(v) else np.array(v, copy=True)) for k, v in sd.items()}
            for m in ("mlp_lr.", "mlp_hr."):
                sd2[m + "conv0.weight"][:, 320] *= gain
            cmp(gname, sd2, Fl, Fh, 512, ("bf16",))
if "body" in fields:
    sd, Fl, Fh = pr.body_inputs(dev)
    cmp("body", sd, Fl, Fh, 512, ("bf16",))
dump = os.environ.get("SURS_V12_JSON")
if dump:
    json.dump(out, open(dump, "w"), indent=1)
bad = [k for k, v in out.items() if not (all(v["bitwise_hr_lr"]) and v["finite"])]
print("MISMATCH" if bad else "ALL BITWISE EQUAL", bad)
sys.exit(1 if bad else 0)
