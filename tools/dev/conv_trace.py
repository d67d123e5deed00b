"""Phase stamps of conv_x3_kernel (library built with -DSURS_CONV_TRACE, SURS_CONV_TRACE=1): a few encoder shapes, one launch each."""
import os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from surs_amd import native, prng
dev = torch.device("cuda:0")
for (h, cin, cout) in ((256, 256, 128), (256, 128, 64), (256, 64, 64), (128, 256, 128), (64, 256, 128), (512, 64, 64), (1024, 64, 64)):
    x = native.Img(h, h, cin, device=dev); x.buf.normal_()
    cw = native.ConvWeights(prng.uniform("w", cin + cout, (cout, cin, 3, 3), -0.1, 0.1), None, dev)
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); native.conv2d(x, cw); e1.record(); torch.cuda.synchronize()
    print("  -> %dx%d %d->%d: %.1f us" % (h, h, cin, cout, e0.elapsed_time(e1) * 1e3), flush=True)
