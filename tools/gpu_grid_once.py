"""Runs the reduced-precision grid sweep a few times (profiling target for rocprofv3 --pmc)."""
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
import gpu_common as g  # noqa: E402
import oracle  # noqa: E402
from surs_amd import native  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
field = sys.argv[3] if len(sys.argv) > 3 else "bench"
if field == "bench":
    # the bench's field: seeded weights, the encoder's features of the synthetic 512 x 512 image (62 listed layer-0 channels per tile:
    # what bench.py sweeps; the PRNG feature maps of "prng" list 107 and belong to the eight-wave kernel by the host's rule)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import precision_report as pr
    sd, Fl, Fh, _keep = pr.noise_inputs(g.dev())
    b = native.pack_mlp({k: v for k, v in sd.items() if k.startswith("mlp_")}, dt, g.dev())[0]
else:
    fl, fh = common.synth_features(hl=256, hh=1024)
    Fl, Fh = g.upload_nhwc(fl), g.upload_nhwc(fh)
    b = g.blob("f16" if dt == "fp16" else "bf16")
ws = native.Workspace(g.dev())
mat = oracle.coords_matrix(R, [-0.5] * 3, [0.5] * 3)[:3].reshape(-1)
vh = torch.empty((R, R, R), dtype=torch.float32, device=g.dev())
vl = torch.empty_like(vh)
import ctypes as C  # noqa: E402
from surs_amd import _lib  # noqa: E402
L = _lib.lib()
L.surs_profile_enable(1)
for _ in range(3):
    native.query_grid(0, R, R, R, mat, common.CALIB.reshape(-1)[:12], 512, 200.0, Fl, Fh, b, dt, ws, vh, vl)
torch.cuda.synchronize()
tiles, ks = C.c_double(0), C.c_double(0)
L.surs_profile_read_ksteps(C.byref(tiles), C.byref(ks))   # column kernels v10 / v11: residual k-steps of layer 1 (data dependent)
L.surs_profile_enable(0)
print("done tile_mlps_per_sweep %.0f residual_ksteps_per_sweep %.0f" % (tiles.value / 3, ks.value / 3))
