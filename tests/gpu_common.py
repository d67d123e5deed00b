"""Helpers for the GPU parity tests (import only inside @pytest.mark.gpu tests)."""
import numpy as np
import torch

import common
from surs_amd import native


def dev():
    return native.require_gpu()


def upload_nhwc(feat_chw):
    """numpy [C,H,W] -> native.Img NHWC on the GPU (host-side transpose, plain copy)."""
    c, h, w = feat_chw.shape
    t = torch.from_numpy(np.ascontiguousarray(feat_chw.transpose(1, 2, 0))).to(dev())
    return native.Img(h, w, c, c, t.reshape(-1))


_blobs = {}


def blob(dtype="bf16"):
    if dtype not in _blobs:
        _blobs[dtype] = native.pack_mlp(common.state_dict(), dtype, dev())[0]
    return _blobs[dtype]
