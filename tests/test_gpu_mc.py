"""GPU marching cubes (through the C ABI) vs the scikit-image goldens: faces and vertex order bit-exact,
vertex positions bit-exact, values exact, normals within 1e-4 (float atomics reorder the sums)."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

import mc_volumes

pytestmark = pytest.mark.gpu


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def mc():
    from surs_amd import native
    dev = native.require_gpu()
    ws = native.Workspace(dev)

    def run(vol, level, normals=True):
        v, f, n, val = native.marching_cubes_lewiner(torch.from_numpy(np.ascontiguousarray(vol, np.float32)).to(dev), level, ws,
                                                     want_normals=normals)
        return v.cpu().numpy(), f.cpu().numpy(), (n.cpu().numpy() if n is not None else None), (val.cpu().numpy() if val is not None else None)

    return run


@pytest.mark.parametrize("name", list(mc_volumes.CASES))
def test_volume_matches_skimage(mc, name, golden_dir):
    vol, level = mc_volumes.CASES[name]()
    g = np.load(os.path.join(golden_dir, "mc_%s.npz" % name))
    v, f, n, val = mc(vol, level)
    assert f.shape == g["faces"].shape and v.shape == g["verts"].shape
    assert np.array_equal(f, g["faces"])
    assert np.array_equal(v, g["verts"])
    assert np.array_equal(val, g["values"])
    assert np.abs(n - g["normals"]).max() < 1e-4


def test_cells_batched(mc, golden_dir):
    """6000 single cells covering every MC33 sub-case, each through its own call (one volume of separated 2x2x2 islands is
    not equivalent to single-cell runs): all of them, faces and vertices bit-exact."""
    cells = mc_volumes.cells(6000, 7)
    g = np.load(os.path.join(golden_dir, "mc_cells.npz"))
    fo = np.concatenate([[0], np.cumsum(g["nf"])])
    vo = np.concatenate([[0], np.cumsum(g["nv"])])
    for i in range(6000):
        try:
            v, f, _, _ = mc(cells[i], 0.0, normals=False)
        except (ValueError, RuntimeError):
            v, f = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32)
        assert np.array_equal(f, g["faces"][fo[i]:fo[i + 1]]), i
        assert np.array_equal(v, g["verts"][vo[i]:vo[i + 1]]), i


def test_equal_level_and_errors(mc, golden_dir):
    vol = mc_volumes.noise((6, 6, 6), 3)
    q = (np.round(vol * 4) / 4).astype(np.float32)
    g = np.load(os.path.join(golden_dir, "mc_equal_level.npz"))
    v, f, n, val = mc(q, 0.5)
    assert np.array_equal(f, g["faces"]) and np.array_equal(v, g["verts"])
    with pytest.raises(ValueError, match="within volume data range"):
        mc(vol, 2.0)
    with pytest.raises(ValueError, match="within volume data range"):
        mc(vol, -1.0)
    with pytest.raises(RuntimeError, match="No surface found"):
        mc(np.full((4, 4, 4), 0.5, np.float32), 0.5)
    with pytest.raises(ValueError):
        mc(np.zeros((1, 4, 4), np.float32), 0.0)


@pytest.mark.parametrize("n", [256, 512])
def test_blob_digest_full_size(mc, n, golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "mc_meta.json")))["blob%d" % n]
    v, f, nr, val = mc(mc_volumes.blob(n), 0.5)
    assert (len(v), len(f)) == (meta["nverts"], meta["nfaces"])
    assert _sha(f) == meta["faces_sha256"]
    assert _sha(v) == meta["verts_sha256"]
    assert _sha(val) == meta["values_sha256"]


def test_nan_in_the_volume_is_reported():
    """fminf / fmaxf skip NaNs, so a volume with NaN voxels would pass the level-range check and give a garbage mesh; the count
    pass reports them (SURS_E_NONFINITE -> NonFiniteVolumeError).  That is how an overflow of the fp32-grade sweep's f16 range
    surfaces; reconstruction() answers it by repeating the sweep on the layer kernels."""
    import torch
    from surs_amd import native
    from surs_amd._lib import NonFiniteVolumeError
    dev = native.require_gpu()
    ws = native.Workspace(dev)
    ax = torch.linspace(-1, 1, 40, device=dev)
    z, y, x = torch.meshgrid(ax, ax, ax, indexing="ij")
    vol = (1.0 / (1.0 + torch.exp(8.0 * (torch.sqrt(x * x + y * y + z * z) - 0.6)))).contiguous()
    v, f, _, _ = native.marching_cubes_lewiner(vol, 0.5, ws)
    assert len(v) > 100
    for where in ((0, 0, 0), (39, 39, 39), (17, 5, 39), (20, 20, 20)):
        bad = vol.clone()
        bad[where] = float("nan")
        with pytest.raises(NonFiniteVolumeError):
            native.marching_cubes_lewiner(bad, 0.5, native.Workspace(dev))
    v2, f2, _, _ = native.marching_cubes_lewiner(vol, 0.5, ws)     # the workspace is usable afterwards
    assert torch.equal(v, v2) and torch.equal(f, f2)
