"""ctypes binding of libsurs_hip.so (the C ABI of include/surs.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C csrc``.
There is no fallback: if it is missing, or a call fails, this raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
from . import settings  # noqa: E402

LIB_PATH = settings.get("SURS_LIB_PATH") or os.path.join(_HERE, "libsurs_hip.so")   # override: timing experiments only
_lib = None

F32, BF16, F16, F32_GEMM = 0, 1, 2, 3
DTYPES = {"fp32": F32, "bf16": BF16, "fp16": F16, "f16": F16, "fp32x": F32_GEMM}


class SursError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("surs error %d: %s" % (code, msg))
        self.code = code


class LevelRangeError(ValueError):
    pass


class NoSurfaceError(RuntimeError):
    pass


class NonFiniteVolumeError(FloatingPointError):
    """The occupancy volume holds NaN values (marching cubes saw them).  With `--precision fp32` that is how an activation
    beyond the f16 range of the fp32-grade column kernel surfaces; reconstruction() then repeats the sweep on the layer
    kernels (fp32's exponent range)."""


class GridOptions(C.Structure):
    """SursGridOptions of include/surs.h: per-call column kernel / operand split of surs_query_grid_opt."""
    _fields_ = [("kernel", C.c_int), ("operand_parts", C.c_int), ("reserved", C.c_int * 6)]


class Conv(C.Structure):
    """SursConv of include/surs.h."""
    _fields_ = [("w_split", C.c_void_p), ("w_packed", C.c_void_p), ("bias", C.c_void_p), ("cin", C.c_int), ("cout", C.c_int),
                ("ksize", C.c_int), ("reserved", C.c_int)]


class GroupNorm(C.Structure):
    _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p)]


class ConvBlock(C.Structure):
    _fields_ = [("conv", Conv * 3), ("bn", GroupNorm * 3)]


class EncoderNet(C.Structure):
    """SursEncoderNet of include/surs.h (device pointers of the packed weights; the arrays behind the pointer fields are kept alive by
    encoder.NativeNet)."""
    _fields_ = [("residual", C.c_int), ("n_block", C.c_int * 3), ("num_stack", C.c_int), ("hg_depth", C.c_int), ("parts", C.c_int),
                ("flags", C.c_int),
                ("head", Conv), ("down", Conv * 3), ("tail0", Conv * 3), ("tail2", Conv * 3), ("bottleneck", Conv), ("bott2", Conv),
                ("ups2", Conv), ("ups3", Conv), ("ups4", Conv), ("last0", Conv), ("last2", Conv),
                ("body", C.POINTER(Conv)), ("conv5", Conv), ("conv2", ConvBlock), ("hg", C.POINTER(ConvBlock)),
                ("top_m", C.POINTER(ConvBlock)), ("conv_last", C.POINTER(Conv)), ("l", C.POINTER(Conv)), ("next", C.POINTER(Conv)),
                ("bn_end", C.POINTER(GroupNorm))]


class EncoderStreams(C.Structure):
    _fields_ = [("side", C.c_void_p * 4)]


class GnStats(C.Structure):
    """SursGnStats of include/surs.h."""
    _fields_ = [("sums", C.c_void_p), ("pitch", C.c_int), ("g1", C.c_int), ("g2", C.c_int), ("slots", C.c_int * 3)]


class McCounts(C.Structure):
    _fields_ = [("n_verts", C.c_int32), ("n_faces", C.c_int32), ("vmin", C.c_float), ("vmax", C.c_float)]


_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_SIGS = {
    "surs_abi_version": (C.c_int, []),
    "surs_last_error": (C.c_char_p, []),
    "surs_device_info": (C.c_int, [C.POINTER(C.c_int), C.c_char_p]),
    "surs_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "surs_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "surs_option_name": (C.c_char_p, [C.c_int]),
    "surs_option_help": (C.c_char_p, [C.c_int]),
    "surs_conv2d_nhwc": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _f, _vp, _i, _vp]),
    "surs_conv_pack_weights": (_sz, [_vp, _i, _i, _i, _vp]),
    "surs_groupnorm_coeffs": (C.c_int, [_vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "surs_groupnorm_coeffs_ws": (C.c_int, [_vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "surs_groupnorm_scratch_bytes": (_sz, []),
    "surs_scale_shift_act": (C.c_int, [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "surs_avgpool2": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "surs_bicubic_up2": (C.c_int, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "surs_pixel_shuffle2": (C.c_int, [_vp, _i, _i, _i, _i, _f, _vp, _i, _vp]),
    "surs_add3": (C.c_int, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _i, _vp]),
    "surs_image_prepare": (C.c_int, [_vp, _vp, _i, _i, _vp, _i, _vp]),
    "surs_nchw_to_nhwc": (C.c_int, [_vp, _i, _i, _i, _vp, _i, _vp]),
    "surs_nhwc_to_nchw": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "surs_conv2d_nhwc_x3": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _f, _vp, _i, _vp]),
    "surs_conv_pack_weights_x3": (_sz, [_vp, _i, _i, _i, _vp]),
    "surs_conv2d_nhwc_x2": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _f, _vp, _i, _vp]),
    "surs_conv_pack_weights_x2": (_sz, [_vp, _i, _i, _i, _vp]),
    "surs_conv_tile_scale": (C.c_int, [_i, _i]),
    "surs_conv2d_nhwc_x1": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _f, _vp, _i, _vp]),
    "surs_conv2d_nhwc_gn": (C.c_int, [_i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _f, _i, _f, _vp, _i,
                                      _vp, _i, C.POINTER(C.c_int), _vp]),
    "surs_avgpool2_gn": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _i, C.POINTER(C.c_int), _vp]),
    "surs_bicubic_up2_gn": (C.c_int, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, C.POINTER(C.c_int), _vp]),
    "surs_add3_gn": (C.c_int, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _i, _vp, _i, C.POINTER(C.c_int), _vp]),
    "surs_conv2d_nhwc_gn_sum": (C.c_int, [_i, _vp, _i, _i, _i, _i, _vp, _vp, C.POINTER(GnStats), _vp, _vp, _vp, _vp, _f, _vp, _i, _i,
                                          C.POINTER(GnStats), _vp, _i, _vp, _i, _vp, _i, _i, _i, C.POINTER(C.c_int), _vp]),
    "surs_encoder_workspace_bytes": (_sz, [C.POINTER(EncoderNet), _i, _i]),
    "surs_encoder_super_res": (C.c_int, [C.POINTER(EncoderNet), _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "surs_encoder_filter_lr": (C.c_int, [C.POINTER(EncoderNet), _vp, _i, _i, _i, C.POINTER(_vp), _vp, _sz, C.POINTER(EncoderStreams), _vp]),
    "surs_encoder_filter_hr": (C.c_int, [C.POINTER(EncoderNet), _vp, _i, _i, _i, _vp, _vp]),
    "surs_encoder_forward": (C.c_int, [C.POINTER(EncoderNet), _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, C.POINTER(EncoderStreams), _vp]),
    "surs_mlp_pack": (_sz, [_vp, _vp, _vp, _vp, _i, _vp]),
    "surs_set_operand_split": (C.c_int, [_i]),
    "surs_set_operand_split_local": (C.c_int, [_i]),
    "surs_set_grid_kernel": (C.c_int, [_i]),
    "surs_query_workspace_bytes": (_sz, [_i]),
    "surs_query_points": (C.c_int, [_vp, _i, _vp, _f, _f, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "surs_query_points_columns": (C.c_int, [_vp, C.c_longlong, _i, _vp, _f, _f, _vp, _i, _i, _vp, _i, _i, _vp, _i, _vp, _sz, _vp, _vp,
                                            C.POINTER(C.c_int), _vp]),
    "surs_query_points_columns_workspace_bytes": (_sz, []),
    "surs_nonfinite": (C.c_int, [_vp, _vp, C.c_longlong, _vp, _vp]),
    "surs_point_runs": (C.c_int, [_vp, C.c_longlong, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "surs_query_points_hr": (C.c_int, [_vp, _i, _vp, _f, _f, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "surs_query_points_views": (C.c_int, [_vp, _i, _i, _i, _vp, _f, _f, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp,
                                          _vp, _vp]),
    "surs_query_views_workspace_bytes": (_sz, [_i, _i]),
    "surs_query_grid_views": (C.c_int, [_i, _i, _i, _i, _vp, _i, _i, _vp, _f, _f, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp]),
    "surs_query_grid_views_workspace_bytes": (_sz, [_i]),
    "surs_query_grid": (C.c_int, [_i, _i, _i, _i, _vp, _vp, _f, _f, _vp, _i, _i, _vp, _i, _i, _vp, _i, _vp, _sz, _vp, _vp, _vp]),
    "surs_query_grid_opt": (C.c_int, [_i, _i, _i, _i, _vp, _vp, _f, _f, _vp, _i, _i, _vp, _i, _i, _vp, _i, _vp, _sz, _vp, _vp,
                                      C.POINTER(GridOptions), _vp]),
    "surs_query_grid_workspace_bytes": (_sz, [_i, _i, _i]),
    "surs_query_grid_probe": (C.c_int, [_i, _i, _i, _i, _vp, _vp, _f, _f, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _sz, _vp, _vp]),
    "surs_octree_select": (C.c_int, [_vp, _i, _i, _vp, _i, _vp, C.POINTER(C.c_int), _vp]),
    "surs_query_grid_indexed": (C.c_int, [_vp, _i, _i, _i, _vp, _vp, _f, _f, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp]),
    "surs_octree_scatter": (C.c_int, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "surs_octree_level_columns": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _f, _f, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _sz,
                                            C.POINTER(C.c_longlong), _vp]),
    "surs_octree_level_columns_dt": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _f, _f, _vp, _i, _i, _vp, _i, _i, _vp, _i, _vp, _sz,
                                               C.POINTER(C.c_longlong), _vp]),
    "surs_octree_columns_workspace_bytes": (_sz, [_i]),
    "surs_octree_workspace_bytes": (_sz, [_i, _i]),
    "surs_octree_cells": (C.c_int, [_vp, _vp, _vp, _i, _i, C.c_double, _vp, _sz, _vp]),
    "surs_f64_to_f32": (C.c_int, [_vp, _vp, C.c_longlong, _vp]),
    "surs_save_obj_mesh": (C.c_int, [C.c_char_p, _vp, C.c_longlong, _vp, C.c_longlong, _i]),
    "surs_profile_enable": (C.c_int, [_i]),
    "surs_profile_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "surs_profile_read_ksteps": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "surs_mc_workspace_bytes": (_sz, [_i, _i, _i]),
    "surs_transform_points": (C.c_int, [_vp, _i, _vp, _vp, _vp]),
    "surs_mc_lewiner_range": (C.c_int, [_vp, _i, _i, _i, _i, _i, C.c_double, _vp, _sz, _vp, _vp, _vp, _i, _vp, _i,
                                        C.POINTER(McCounts), _vp]),
    "surs_mc_normalize": (C.c_int, [_vp, _i, _vp]),
    "surs_mc_lewiner_range_slab": (C.c_int, [_vp, _i, _i, _i, _i, _i, C.c_double, _vp, _sz, _vp, _i, _vp, _i, C.POINTER(McCounts), _i, _vp]),
    "surs_mc_slab_top_ids": (C.c_int, [_vp, _sz, _i, _i, _i, _vp, _vp]),
    "surs_mc_slab_fixup": (C.c_int, [_vp, C.c_longlong, _i, _vp, _i, _vp]),
    "surs_mc_lewiner": (C.c_int, [_vp, _i, _i, _i, C.c_double, _vp, _sz, _vp, _vp, _vp, _i, _vp, _i, C.POINTER(McCounts), _vp]),
}
EXPORTS = sorted(_SIGS)


def lib():
    """The loaded library.  Raises if it has not been built (never falls back to a CPU path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libsurs_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(or make -C %s/csrc)" % _HERE)
        # PyTorch first: its wheel carries its own HIP runtime (libamdhip64), and the library must bind to THAT copy - loaded
        # before torch, it would pull in /opt/rocm's, torch would then run on a second runtime in the same process, and the first
        # kernel launch fails with "no ROCm-capable device is detected" (seen with build() and smoke() in one process)
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)  # AttributeError if the library misses a declared entry point
            fn.restype = res
            fn.argtypes = args
        if l.surs_abi_version() != 1:
            raise ImportError("libsurs_hip.so has ABI version %d, expected 1" % l.surs_abi_version())
        _lib = l
    return _lib


def check(code):
    if code == 0:
        return
    msg = lib().surs_last_error().decode("utf-8", "replace")
    if code == -4:
        raise LevelRangeError("Surface level must be within volume data range.")
    if code == -5:
        raise NoSurfaceError("No surface found at the given iso value.")
    if code == -7:
        raise NonFiniteVolumeError("the occupancy volume contains NaN values")
    raise SursError(code, msg)
