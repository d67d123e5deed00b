"""Is the encoder bound by the host's launch rate, and does a HIP graph of it replay correctly and faster?"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from surs_amd import encoder, model, options, weights
from surs_amd.model import _as_img
dev = torch.device("cuda:0")
for prec in ("fp32", "bf16"):
    opt = options.BaseOptions().parse(common.FLAGS + ["--precision", prec])
    net = model.SuRSNet(opt).to(device=dev)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in common.state_dict().items()})
    net.eval()
    img = torch.from_numpy(weights.synthetic_image(512, seed=1)).to(dev)
    W = net._encoder_weights()

    def run(x):
        _, f_lr, f_hr = encoder.super_res(W, x)
        return encoder.filter_lr(W, f_lr)[-1], encoder.filter_hr(W, f_hr)[0]

    x = _as_img(img)
    for _ in range(3): run(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): run(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s eager: host enqueue %.2f ms per encoder, wall %.2f ms" % (prec, (t1 - t0) / 10 * 1e3, (t2 - t0) / 10 * 1e3), flush=True)
    ref = [t.buf.clone() for t in run(x)]
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    static = img.clone()
    xs = _as_img(static)
    with torch.cuda.stream(s):
        for _ in range(2): run(xs)
    torch.cuda.synchronize()
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            outs = run(xs)
        torch.cuda.synchronize()
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        print("%s graph replay: %.2f ms per encoder; outputs %s" % (prec, (time.perf_counter() - t0) / 10 * 1e3,
              "identical" if all(torch.equal(a, o.buf) for a, o in zip(ref, outs)) else "DIFFERENT"), flush=True)
        # another image through the same graph
        img2 = torch.from_numpy(weights.synthetic_image(512, seed=2)).to(dev)
        want = [t.buf.clone() for t in run(_as_img(img2))]
        static.copy_(img2); g.replay(); torch.cuda.synchronize()
        print("   second image through the graph:", "identical" if all(torch.equal(a, o.buf) for a, o in zip(want, outs)) else "DIFFERENT", flush=True)
    except Exception as e:
        print("graph capture failed:", repr(e)[:400], flush=True)
