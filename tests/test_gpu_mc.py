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


@pytest.mark.parametrize("shape", [(40, 40, 40), (23, 36, 52), (20, 24, 30)])
def test_both_emit_paths_against_the_oracle(mc, shape):
    """The emit pass has two forms: it normally sorts the cell codes the count pass stored (smooth field below: a few per
    cent of the cells are active), and classifies the active cells again where more than half of the cells are active and the sorted
    list would overlap the stored codes (white noise: ~99 %).  Both against the oracle (the C restatement pinned to scikit-image),
    faces / vertices / values bit for bit; (.., 30): nx % 4 != 0, the scalar-row form of the count pass."""
    import oracle
    rng = np.random.RandomState(5)
    noise = rng.rand(*shape).astype(np.float32)
    ax = [np.linspace(-1, 1, n, dtype=np.float32) for n in shape]
    z, y, x = np.meshgrid(*ax, indexing="ij")
    smooth = (1.0 / (1.0 + np.exp(9.0 * (np.sqrt(x * x + 1.3 * y * y + 0.8 * z * z) - 0.55)))).astype(np.float32)
    for vol, min_active in ((noise, 0.5), (smooth, 0.0)):
        ov, of, on, oval = oracle.marching_cubes_lewiner(vol, 0.5)
        v, f, n, val = mc(vol, 0.5)
        ncells = (shape[0] - 1) * (shape[1] - 1) * (shape[2] - 1)
        assert len(of) > min_active * ncells       # (every active cell has at least one triangle)
        assert np.array_equal(f, of) and np.array_equal(v, ov) and np.array_equal(val, oval)
        assert np.abs(n - on).max() < 1e-4


def test_emit_reclassify_switch_gives_the_same_mesh():
    """SURS_MC_EMIT_RECLASSIFY=1 forces the re-classifying emit pass on a field that would take the stored codes: same mesh."""
    import subprocess
    import sys
    code = r"""
import hashlib, sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import mc_volumes
from surs_amd import native
dev = native.require_gpu()
v, f, n, val = native.marching_cubes_lewiner(torch.from_numpy(mc_volumes.blob(96)).to(dev), 0.5, native.Workspace(dev))
h = hashlib.sha256(); h.update(v.cpu().numpy().tobytes()); h.update(f.cpu().numpy().tobytes()); h.update(val.cpu().numpy().tobytes())
print("digest", h.hexdigest(), len(v), len(f))
""" % (os.path.abspath(os.path.join(os.path.dirname(__file__), "..")), os.path.abspath(os.path.dirname(__file__)))
    outs = []
    for flag in ("0", "1"):
        env = dict(os.environ, SURS_MC_EMIT_RECLASSIFY=flag)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("digest")][0])
    assert outs[0] == outs[1] and int(outs[0].split()[-1]) > 1000
