"""Prints sha256 digests of the column-kernel volumes for a few grid sizes, three launches each (race screen; compare two
builds of one kernel bit for bit: SURS_GRID_KERNEL / SURS_LIB_PATH select kernel and library)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
import gpu_common as g  # noqa: E402
import oracle  # noqa: E402
from surs_amd import native  # noqa: E402


def main():
    fl, fh = common.synth_features()
    Fl, Fh = g.upload_nhwc(fl), g.upload_nhwc(fh)
    ws = native.Workspace(g.dev())
    for dt in ("bf16", "fp16"):
        for R in (40, 136):
            mat = oracle.coords_matrix(R, [-0.5] * 3, [0.5] * 3)[:3].reshape(-1)
            b = g.blob("f16" if dt == "fp16" else "bf16")
            hs = set()
            for rep in range(3):   # repeated launches must agree with themselves (race screen)
                vh, vl = native.query_grid(0, R, R, R, mat, common.CALIB.reshape(-1)[:12], 512, 200.0, Fl, Fh, b, dt, ws)
                hs.add(hashlib.sha256(vh.cpu().numpy().tobytes() + vl.cpu().numpy().tobytes()).hexdigest())
            print(dt, R, "stable" if len(hs) == 1 else "UNSTABLE", sorted(hs)[0])


if __name__ == "__main__":
    main()
