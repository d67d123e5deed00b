"""Phase stamps of the split-f16 3x3 convolution kernel (library built with -DSURS_CONV_TRACE: tools/build_trace.sh -DSURS_CONV_TRACE):
    SURS_CONV_TRACE=1 SURS_LIB_PATH=abl/libsurs_trace.so python tools/diag/conv_trace.py
plus the wall time of each shape on the production kernel when run without the trace library."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from surs_amd import native
dev = native.require_gpu()
rng = np.random.RandomState(0)
shapes = [(256, 128, 64), (128, 64, 64), (64, 64, 64), (256, 128, 128), (128, 64, 128), (64, 64, 128), (256, 128, 256), (128, 64, 256), (64, 64, 256)]
for reduced in (True, False):
    for cin, cout, hw in shapes:
        x = native.Img(hw, hw, cin, device=dev)
        x.buf.copy_(torch.from_numpy(rng.uniform(-1, 1, x.buf.numel()).astype(np.float32)))
        cw = native.ConvWeights(rng.uniform(-0.1, 0.1, (cout, cin, 3, 3)).astype(np.float32), None, dev, reduced=reduced)
        out = native.Img(hw, hw, cout, device=dev)
        native.conv2d(x, cw, out=out)
        torch.cuda.synchronize()
        if os.environ.get("SURS_CONV_TRACE"):
            continue
        n = 50
        t = time.perf_counter()
        for _ in range(n):
            native.conv2d(x, cw, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n
        print("conv3x3 %s cin %3d cout %3d %3dx%-3d: %6.1f us  (%.0f TFLOP/s of products)" % ("f16x1" if reduced else "f16x2", cin, cout, hw, hw, dt * 1e6,
                                                                                   2.0 * hw * hw * 9 * cin * cout * (1 if reduced else 3) / dt / 1e12))
