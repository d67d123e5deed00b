"""Shared helpers for the tests: flags, synthetic inputs (all from seeds)."""
import numpy as np

from surs_amd import options, prng, weights

FLAGS = ["--loadSize", "1024", "--residual", "--b_min", "-0.5", "-0.5", "-0.5", "--b_max", "0.5", "0.5", "0.5",
         "--num_samples", "50000", "--z_size", "200"]
CALIB = np.diag([2.0, -2.0, 2.0, 1.0]).astype(np.float32)
_cache = {}


def opt():
    return options.BaseOptions().parse(FLAGS)


def state_dict(seed=0):
    if seed not in _cache:
        _cache[seed] = weights.synthetic_state_dict(opt(), seed=seed)
    return _cache[seed]


def synth_features(seed=3, hl=32, hh=128):
    """Same PRNG features tools/gen_golden.py fed to the reference's query."""
    fl = prng.uniform("feat_lr", seed, (256, hl, hl), -1.0, 1.0)
    fh = prng.uniform("feat_hr", seed, (64, hh, hh), -1.0, 1.0)
    return fl, fh


def rel_err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(1e-12, np.abs(b).max()))
