"""Import the upstream reference (read-only, /root/reference) in THIS container.

Test infrastructure only: used by tools/gen_golden.py to produce the fixtures
under tests/golden/.  Nothing here (or anything it imports from
/root/reference) travels to the GPU box; tests, bench.py and smoke() never
import this module.

The reference needs packages this image lacks (skimage, cv2, trimesh, imageio,
torchvision).  They are only imported, never used, on the path we exercise, so
empty stub modules are injected before `lib.*` is imported.
Marching cubes is bridged to /opt/conda/bin/python3.9 (scikit-image 0.18.3)
through .npy files because that is the only skimage in the image.
"""
import contextlib
import io
import os
import subprocess
import sys
import tempfile
import types

import numpy as np

REF_ROOT = "/root/reference"
CONDA_PY = "/opt/conda/bin/python3.9"

_MC_SCRIPT = r"""
import sys, warnings, numpy as np
warnings.simplefilter('ignore')
from skimage import measure
vol = np.load(sys.argv[1]); level = float(sys.argv[2])
v, f, n, val = measure.marching_cubes_lewiner(vol, level)
np.savez(sys.argv[3], verts=v, faces=f, normals=n, values=val)
"""


def skimage_mc(volume, level):
    """marching_cubes_lewiner(volume, level) run by the conda interpreter."""
    with tempfile.TemporaryDirectory() as d:
        vin, vout = os.path.join(d, "v.npy"), os.path.join(d, "o.npz")
        np.save(vin, np.asarray(volume))
        env = dict(os.environ)
        env.pop("PYTHONPATH", None)
        r = subprocess.run([CONDA_PY, "-c", _MC_SCRIPT, vin, str(level), vout],
                           capture_output=True, text=True, env=env)
        if r.returncode != 0:
            msg = r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "?"
            if "within volume data range" in msg:
                raise ValueError(msg)
            if "No surface found" in msg:
                raise RuntimeError(msg)
            raise RuntimeError("conda skimage failed: " + r.stderr)
        o = np.load(vout)
        return o["verts"], o["faces"], o["normals"], o["values"]


def _install_stubs():
    if not hasattr(np, "bool"):
        np.bool = bool  # lib/sdf.py:63-64 uses the removed alias

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    measure = mod("skimage.measure", marching_cubes_lewiner=skimage_mc)
    sk = mod("skimage", measure=measure)
    for sub in ("io", "color", "transform"):
        setattr(sk, sub, mod("skimage." + sub))
    for name in ("cv2", "trimesh", "imageio"):
        mod(name)
    tv_utils = mod("torchvision.utils", save_image=lambda *a, **k: None)
    tv_tf = mod("torchvision.transforms")
    tv_resnet = mod("torchvision.models.resnet")
    tv_vgg = mod("torchvision.models.vgg")
    tv_models = mod("torchvision.models", resnet=tv_resnet, vgg=tv_vgg)
    mod("torchvision", utils=tv_utils, transforms=tv_tf, models=tv_models)
    if "tqdm" not in sys.modules:
        try:
            import tqdm  # noqa: F401
        except Exception:
            mod("tqdm", tqdm=lambda x, *a, **k: x)


_loaded = {}


def load_reference():
    """Returns a namespace with the reference's modules imported."""
    if _loaded:
        return _loaded["ns"]
    _install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    with contextlib.redirect_stdout(io.StringIO()):
        from lib.options import BaseOptions
        from lib.model.SuRSNet import SuRSNet
        from lib import mesh_util, sdf, geometry, train_util
    ns = types.SimpleNamespace(BaseOptions=BaseOptions, SuRSNet=SuRSNet, mesh_util=mesh_util,
                               sdf=sdf, geometry=geometry, train_util=train_util)
    _loaded["ns"] = ns
    return ns


def parse_opt(argv):
    ns = load_reference()
    old = sys.argv
    sys.argv = ["ref"] + list(argv)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            opt = ns.BaseOptions().parse()
    finally:
        sys.argv = old
    return opt


@contextlib.contextmanager
def quiet():
    """The reference prints the whole z tensor on every query (DepthNormalizer.py:17)."""
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def build_net(opt):
    import torch
    ns = load_reference()
    with quiet():
        net = ns.SuRSNet(opt, "orthogonal").to(torch.device("cpu"))
    net.eval()
    return net
