"""Pins the marching-cubes oracle (oracle/mc_oracle.c) against outputs of the compiled
scikit-image Lewiner core captured by tools/gen_mc_golden.py (tests/golden/mc_*).
Bar: faces (vertex indices, order) bit-exact, vertex positions bit-exact, values exact,
normals within 1e-5 (fp32 accumulation order)."""
import hashlib
import json
import os

import numpy as np
import pytest

import mc_volumes
import oracle


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", list(mc_volumes.CASES))
def test_volume_matches_skimage(name, golden_dir):
    vol, level = mc_volumes.CASES[name]()
    g = np.load(os.path.join(golden_dir, "mc_%s.npz" % name))
    v, f, n, val = oracle.marching_cubes_lewiner(vol, level)
    assert f.dtype == np.int32 and v.dtype == np.float32
    assert np.array_equal(f, g["faces"])
    assert np.array_equal(v, g["verts"])
    assert np.array_equal(val, g["values"])
    assert np.abs(n - g["normals"]).max() < 1e-5


def test_single_cells_cover_all_cases(golden_dir):
    cells = mc_volumes.cells(6000, 7)
    g = np.load(os.path.join(golden_dir, "mc_cells.npz"))
    fo = np.concatenate([[0], np.cumsum(g["nf"])])
    vo = np.concatenate([[0], np.cumsum(g["nv"])])
    for i, c in enumerate(cells):
        try:
            v, f, _, _ = oracle.marching_cubes_lewiner(c, 0.0)
        except (ValueError, RuntimeError):
            v, f = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32)
        assert np.array_equal(f, g["faces"][fo[i]:fo[i + 1]]), i
        assert np.array_equal(v, g["verts"][vo[i]:vo[i + 1]]), i


def test_value_equal_to_level_is_outside(golden_dir):
    vol = mc_volumes.noise((6, 6, 6), 3)
    q = (np.round(vol * 4) / 4).astype(np.float32)
    g = np.load(os.path.join(golden_dir, "mc_equal_level.npz"))
    v, f, n, val = oracle.marching_cubes_lewiner(q, 0.5)
    assert np.array_equal(f, g["faces"]) and np.array_equal(v, g["verts"])


def test_errors_like_skimage(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "mc_meta.json")))
    vol = mc_volumes.noise((6, 6, 6), 3)
    assert meta["level_above"].startswith("ValueError")
    with pytest.raises(ValueError, match="within volume data range"):
        oracle.marching_cubes_lewiner(vol, 2.0)
    with pytest.raises(ValueError, match="within volume data range"):
        oracle.marching_cubes_lewiner(vol, -1.0)
    assert meta["flat"].startswith("RuntimeError")
    with pytest.raises(RuntimeError, match="No surface found"):
        oracle.marching_cubes_lewiner(np.full((4, 4, 4), 0.5, np.float32), 0.5)
    with pytest.raises(ValueError):
        oracle.marching_cubes_lewiner(np.zeros((1, 4, 4), np.float32), 0.0)


def test_blob256_digest(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "mc_meta.json")))["blob256"]
    v, f, n, val = oracle.marching_cubes_lewiner(mc_volumes.blob(256), 0.5)
    assert (len(v), len(f)) == (meta["nverts"], meta["nfaces"])
    assert _sha(f) == meta["faces_sha256"]
    assert _sha(v) == meta["verts_sha256"]
    assert _sha(val) == meta["values_sha256"]
