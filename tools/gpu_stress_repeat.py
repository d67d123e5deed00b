"""Race screen of the default column kernels: the same sweep repeated N times must give the same bits every time (LDS-DMA
gather, dynamic column hand-out, list compaction).  python tools/gpu_stress_repeat.py [N]"""
import hashlib
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
import gpu_common as g  # noqa: E402
import oracle  # noqa: E402
from surs_amd import native  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
fl, fh = common.synth_features(hl=64, hh=256)
Fl, Fh = g.upload_nhwc(fl), g.upload_nhwc(fh)
ws = native.Workspace(g.dev())
bad = 0
for dt in ("bf16", "fp16", "fp32"):
    for R in (136, 200):
        mat = oracle.coords_matrix(R, [-0.5] * 3, [0.5] * 3)[:3].reshape(-1)
        b = g.blob("f16" if dt == "fp16" else "bf16")
        hs = set()
        for rep in range(N):
            vh, vl = native.query_grid(0, R, R, R, mat, common.CALIB.reshape(-1)[:12], 512, 200.0, Fl, Fh, b, dt, ws)
            hs.add(hashlib.sha256(vh.cpu().numpy().tobytes() + vl.cpu().numpy().tobytes()).hexdigest())
        print(dt, R, "stable" if len(hs) == 1 else "UNSTABLE (%d digests)" % len(hs), flush=True)
        bad += len(hs) != 1
sys.exit(1 if bad else 0)
