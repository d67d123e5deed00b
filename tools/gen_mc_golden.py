"""Generate marching-cubes golden vectors with the compiled scikit-image core.

Build-container only (needs /opt/conda's scikit-image 0.18.3, the only skimage
here; the reference pins 0.17.2, same Lewiner core).  Writes tests/golden/mc_*.
The input volumes are rebuilt from seeds by tests/mc_volumes.py, so only the
OUTPUTS are stored.

    python tools/gen_mc_golden.py
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import mc_volumes  # noqa: E402
import ref_harness as rh  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

_BATCH = r"""
import sys, warnings, numpy as np
warnings.simplefilter('ignore')
from skimage import measure
vols = np.load(sys.argv[1]); level = float(sys.argv[2])
nv, nf, V, F = [], [], [], []
for v in vols:
    try:
        ve, f, n, val = measure.marching_cubes_lewiner(v, level)
    except (ValueError, RuntimeError):
        ve, f = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32)
    nv.append(len(ve)); nf.append(len(f)); V.append(ve); F.append(f)
np.savez(sys.argv[3], nv=np.array(nv, np.int32), nf=np.array(nf, np.int32),
         verts=np.concatenate(V).astype(np.float32), faces=np.concatenate(F).astype(np.int32))
"""


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    os.makedirs(GOLD, exist_ok=True)
    for name, make in mc_volumes.CASES.items():
        vol, level = make()
        v, f, n, val = rh.skimage_mc(vol, level)
        np.savez_compressed(os.path.join(GOLD, "mc_%s.npz" % name), verts=v, faces=f, normals=n, values=val)
        print(name, v.shape, f.shape)

    # single cells: every MC33 case / sub-case incl. the test_internal fall-through
    cells = mc_volumes.cells(6000, 7)
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "c.npy"), cells)
        subprocess.run([rh.CONDA_PY, "-c", _BATCH, os.path.join(d, "c.npy"), "0.0", os.path.join(d, "o.npz")],
                       env=env, check=True, capture_output=True)
        o = np.load(os.path.join(d, "o.npz"))
        np.savez_compressed(os.path.join(GOLD, "mc_cells.npz"), nv=o["nv"], nf=o["nf"], verts=o["verts"],
                            faces=o["faces"])
    print("cells", int(o["nf"].sum()), "faces")

    # large volumes: digests only
    meta = {}
    for n in (256, 512):
        vol = mc_volumes.blob(n)
        v, f, nr, val = rh.skimage_mc(vol, 0.5)
        meta["blob%d" % n] = {"nverts": int(len(v)), "nfaces": int(len(f)), "verts_sha256": sha(v),
                              "faces_sha256": sha(f), "values_sha256": sha(val)}
        print("blob", n, v.shape, f.shape)
    # edge cases
    vol = mc_volumes.noise((6, 6, 6), 3)
    q = np.round(vol * 4) / 4  # many voxels exactly equal to the level 0.5
    v, f, nr, val = rh.skimage_mc(q.astype(np.float32), 0.5)
    np.savez_compressed(os.path.join(GOLD, "mc_equal_level.npz"), verts=v, faces=f, normals=nr, values=val)
    for tag, level in (("above", 2.0), ("below", -1.0)):
        try:
            rh.skimage_mc(vol, level)
            meta["level_" + tag] = "ok"
        except ValueError as e:
            meta["level_" + tag] = "ValueError: " + str(e).split(": ")[-1]
    flat = np.full((4, 4, 4), 0.5, np.float32)
    try:
        rh.skimage_mc(flat, 0.5)
        meta["flat"] = "ok"
    except RuntimeError as e:
        meta["flat"] = "RuntimeError: " + str(e).split(": ")[-1]
    with open(os.path.join(GOLD, "mc_meta.json"), "w") as fjs:
        json.dump(meta, fjs, indent=1, sort_keys=True)
    print(json.dumps(meta, indent=1))


if __name__ == "__main__":
    main()
