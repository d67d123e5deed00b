"""Every switch of the host mirror, in ONE table: name -> (default, what it does).  None of them needs setting - the defaults are the
product's configuration; they exist for A/B timing, for the tests that hold two forms of one computation to each other, and for
diagnosis.  `get(name)` is the only place the package reads the environment: an override made with `set(name, value)` wins, then the
environment variable of the same name, then the default.  (The library's own switches are its options: native.set_option /
include/surs.h surs_set_option.)"""
import os

SWITCHES = {
    "SURS_LIB_PATH": (None, "another build of libsurs_hip.so (timing experiments, diagnostic builds)"),
    "SURS_ENC_NATIVE": ("1", "0: the encoder's launches sequenced by encoder.py instead of inside the library"),
    "SURS_ENC_SEPARATE_SUM": ("0", "1: a ConvBlock's closing sum as a launch of its own (the form encoder.py's sequencing reproduces)"),
    "SURS_ENC_STREAMS": ("1", "0: the hourglass branches one behind the other on the caller's stream"),
    "SURS_ENC_STREAM_PRIORITY": ("1", "0: default priority for the hourglass's side streams"),
    "SURS_ENC_GRAPH": (None, "1 / 0: the encoder as captured HIP graphs whatever the network's --encoder_graph says"),
    "SURS_ENC_FUSED_GN": ("1", "0: GroupNorm coefficients by two launches per normalisation (rounds 1 - 3)"),
    "SURS_CONV_SPLIT": ("f16x2", "bf16x3: three bf16 parts per operand in the 3x3 convolutions"),
    "SURS_CONV_X3": ("1", "0: every convolution on the fp32 MFMA kernel"),
    "SURS_POINT_RUNS": ("1", "0: query_mr / query_sr never take the column kernels (point arrays that come as runs)"),
    "SURS_GRID_AUTO": ("1", "0: no probe - the library's default column kernel for every sweep"),
    "SURS_GRID_KERNEL": (None, "a reduced-precision column kernel for the process (3, 10, 12): the library option; its presence turns the probe off"),
    "SURS_GRID_F32_KERNEL": (None, "an fp32-grade column kernel for the process (5, 11): as above"),
    "SURS_OCTREE_COLUMNS": ("1", "0: the octree levels on the per-point kernels"),
    "SURS_SLAB_COLUMNS": (None, "columns per launch of a slab's sweep (tests: several launches per slab)"),
    "SURS_SLAB_P2P": ("0", "1: slab meshes point to point instead of through shared host memory"),
}
_overrides = {}


def get(name):
    if name not in SWITCHES:
        raise KeyError("unknown switch %s" % name)
    if name in _overrides:
        return _overrides[name]
    return os.environ.get(name, SWITCHES[name][0])


def is_set(name):
    """The switch has a value other than its built-in default's source (an override or an environment variable)."""
    if name not in SWITCHES:
        raise KeyError("unknown switch %s" % name)
    return name in _overrides or name in os.environ


def set(name, value):   # noqa: A001 - settings.set(...)
    """Override a switch for this process (None: back to the environment / default)."""
    if name not in SWITCHES:
        raise KeyError("unknown switch %s" % name)
    if value is None:
        _overrides.pop(name, None)
    else:
        _overrides[name] = str(value)
