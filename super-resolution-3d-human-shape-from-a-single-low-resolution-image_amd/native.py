"""Thin torch-tensor wrappers over the C ABI (include/surs.h).

PyTorch is used for device memory and streams only; every function here hands raw
device pointers to libsurs_hip.so.  Tensors must live on the current CUDA (HIP)
device.  Nothing here computes on the CPU and nothing falls back.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib, settings
from ._lib import BF16, DTYPES, F16, F32, check, lib  # noqa: F401


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda, "device tensor expected"
    return C.c_void_p(t.data_ptr())


def _f32c(t):
    assert t.dtype == torch.float32 and t.is_contiguous(), "contiguous float32 tensor expected"
    return t


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("surs_amd needs a HIP device (MI355X); there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def device_info():
    cu = C.c_int(0)
    arch = C.create_string_buffer(32)
    check(lib().surs_device_info(C.byref(cu), arch))
    return cu.value, arch.value.decode()


# ------------------------------------------------------------------ NHWC image tensors

class Img:
    """NHWC fp32 device image: a view (h, w, c) into a buffer with channel pitch ld."""
    __slots__ = ("buf", "h", "w", "c", "ld", "off", "stats")

    def __init__(self, h, w, c, ld=None, buf=None, off=0, device=None):
        ld = c if ld is None else ld
        if buf is None:
            buf = torch.empty(h * w * ld, dtype=torch.float32, device=device or require_gpu())
        self.buf, self.h, self.w, self.c, self.ld, self.off = buf, h, w, c, ld, off
        self.stats = None   # GnStats of exactly these values, if the kernel that wrote them left any (conv2d_gn, *_gn below)

    def ptr(self):
        return C.c_void_p(self.buf.data_ptr() + 4 * self.off)

    def slice(self, c0, c):
        """channels [c0, c0+c) of the same pixels (the reference's torch.cat, in place)."""
        return Img(self.h, self.w, c, self.ld, self.buf, self.off + c0)

    def to_nchw(self):
        out = torch.empty((1, self.c, self.h, self.w), dtype=torch.float32, device=self.buf.device)
        check(lib().surs_nhwc_to_nchw(self.ptr(), self.c, self.h, self.w, self.ld, _ptr(out), _stream()))
        return out

    @staticmethod
    def from_nchw(t, ld=None):
        t = _f32c(t.reshape(t.shape[-3:]).contiguous())
        c, h, w = t.shape
        img = Img(h, w, c, ld, device=t.device)
        check(lib().surs_nchw_to_nhwc(_ptr(t), c, h, w, img.ptr(), img.ld, _stream()))
        return img


import contextlib
import threading

_wide = threading.local()


@contextlib.contextmanager
def wide_operands():
    """Everything computed inside runs its fp32-grade matrix products on three bf16 parts per operand (fp32's exponent range)
    instead of two f16 parts (|x| < 65504): the point / octree / multi-view kernels and the GEMMs behind the sweep through the
    calling thread's operand split (surs_set_operand_split_local), the encoder's 3x3 convolutions through ConvWeights'
    bf16 x 3 image (packed on first use).  What reconstruction() and SuRSNet.query_* repeat a computation under after it produced
    non-finite values - the reference is plain fp32 and has no such range limit."""
    prev = getattr(_wide, "on", False)
    _wide.on = True
    check(lib().surs_set_operand_split_local(3))
    try:
        yield
    finally:
        _wide.on = prev
        check(lib().surs_set_operand_split_local(3 if prev else 0))


def wide_operands_active():
    return getattr(_wide, "on", False)


@contextlib.contextmanager
def reduced_point_operands(on=True):
    """Inside, surs_query_points / surs_query_points_hr of this thread run ONE f16 product per MAC (one f16 part per operand: 11
    significant bits, a third of the matrix work of the fp32-grade point path): the arbitrary-point evaluator of `--precision bf16 |
    fp16` (SuRSNet.query_mr / query_sr).  No effect inside wide_operands() (the retry after an overflow is fp32-grade) or with
    on=False."""
    if not on or wide_operands_active():
        yield
        return
    check(lib().surs_set_operand_split_local(1))
    try:
        yield
    finally:
        check(lib().surs_set_operand_split_local(3 if wide_operands_active() else 0))


class ConvWeights:
    """Packed conv weights ([tap][cin_pad][cout_pad]) + bias on the device."""

    def __init__(self, w, b, device, reduced=False):
        w = np.ascontiguousarray(w, np.float32)
        self._host_w, self._w3_wide = w, None
        self.reduced = bool(reduced)   # 3x3: one f16 product per MAC (surs_conv2d_nhwc_x1) - the encoder of --precision bf16 / fp16
        self.cout, self.cin, self.k = w.shape[0], w.shape[1], w.shape[2]
        n = lib().surs_conv_pack_weights(None, self.cout, self.cin, self.k, None)
        packed = np.empty(n, np.float32)
        lib().surs_conv_pack_weights(w.ctypes.data_as(C.c_void_p), self.cout, self.cin, self.k,
                                     packed.ctypes.data_as(C.c_void_p))
        self.w = torch.from_numpy(packed).to(device)
        self.b = torch.from_numpy(np.ascontiguousarray(b, np.float32)).to(device) if b is not None else None
        # split image for the 3x3 / stride-1 kernel: two f16 parts (surs_conv2d_nhwc_x2, default) or, with SURS_CONV_SPLIT=bf16x3,
        # three bf16 parts (surs_conv2d_nhwc_x3: fp32's exponent range); SURS_CONV_X3=0: neither (fp32 MFMA kernel).  1x1
        # convolutions have a two-part kernel only (conv1x1_x2_kernel); with three parts asked for they stay on the fp32 MFMA kernel
        self.w3, self.parts = None, 3 if settings.get("SURS_CONV_SPLIT").startswith("b") else 2
        if (self.k == 3 or (self.k == 1 and self.parts == 2)) and settings.get("SURS_CONV_X3") != "0":
            pack = lib().surs_conv_pack_weights_x3 if self.parts == 3 else lib().surs_conv_pack_weights_x2
            nb = pack(None, self.cout, self.cin, self.k, None)
            buf = np.empty(nb, np.uint8)
            pack(w.ctypes.data_as(C.c_void_p), self.cout, self.cin, self.k, buf.ctypes.data_as(C.c_void_p))
            self.w3 = torch.from_numpy(buf).to(device)

    def split_image(self):
        """(packed split weights, parts) for the 3x3 / stride-1 kernel; inside wide_operands() the bf16 x 3 image, packed on first use."""
        if self.w3 is None or self.parts == 3 or not wide_operands_active():
            return self.w3, self.parts
        if self.k == 1:
            return None, 0    # (1x1: the plain fp32 MFMA kernel is the wide form)
        if self._w3_wide is None:
            pack = lib().surs_conv_pack_weights_x3
            nb = pack(None, self.cout, self.cin, self.k, None)
            buf = np.empty(nb, np.uint8)
            pack(self._host_w.ctypes.data_as(C.c_void_p), self.cout, self.cin, self.k, buf.ctypes.data_as(C.c_void_p))
            self._w3_wide = torch.from_numpy(buf).to(self.w.device)
        return self._w3_wide, 3


def conv2d(x, cw, out=None, stride=1, in_scale=None, in_shift=None, act=0, slope=0.0, residual=None):
    pad = cw.k // 2
    ho, wo = (x.h + 2 * pad - cw.k) // stride + 1, (x.w + 2 * pad - cw.k) // stride + 1
    assert x.c == cw.cin
    if out is None:
        out = Img(ho, wo, cw.cout, device=x.buf.device)
    assert (out.h, out.w, out.c) == (ho, wo, cw.cout)
    # 3 output channels (the image head of the super-resolution net): the direct fp32 kernel behind surs_conv2d_nhwc, not a matrix tile
    thin = cw.k == 3 and stride == 1 and in_scale is None and cw.cout <= 4 and cw.cin == 32
    x3 = (cw.w3 is not None and not thin and (stride == 1 or (stride == 2 and cw.k == 3)) and x.c % 16 == 0 and x.ld % 4 == 0
          and (x.buf.data_ptr() + 4 * x.off) % 16 == 0)
    w3, parts = cw.split_image() if x3 else (None, 0)
    x3 = x3 and w3 is not None and (stride == 1 or parts == 2)
    fn, wt = ((lib().surs_conv2d_nhwc_x3 if parts == 3 else lib().surs_conv2d_nhwc_x2), w3) if x3 else (lib().surs_conv2d_nhwc, cw.w)
    if x3 and parts == 2 and cw.reduced and cw.k == 3 and not wide_operands_active():
        fn = lib().surs_conv2d_nhwc_x1
    check(fn(x.ptr(), x.h, x.w, x.c, x.ld, _ptr(wt), _ptr(cw.b), out.ptr(), cw.cout, out.ld, cw.k,
             stride, _ptr(in_scale), _ptr(in_shift), act, slope,
             residual.ptr() if residual is not None else None,
             residual.ld if residual is not None else 0, _stream()))
    out.stats = None
    return out


class GnStats:
    """GroupNorm(32) statistics of an image as the kernel that wrote it left them: partial sums [32 groups][slots][2] (sum, sum of
    squares; float64), folded in a fixed order by the 3x3 convolution that applies the normalisation (conv2d_gn)."""
    __slots__ = ("buf", "slots")

    def __init__(self, capacity, device):
        self.buf = torch.empty(32 * capacity * 2, dtype=torch.float64, device=device)
        self.slots = 0


def fused_groupnorm():
    """SURS_ENC_FUSED_GN=0: GroupNorm coefficients by surs_groupnorm_coeffs' two launches per normalisation (rounds 1 - 3)."""
    return settings.get("SURS_ENC_FUSED_GN") != "0"


def conv_gn_eligible(x, cw):
    """Can conv2d_gn run this 3x3 convolution (two-part f16 weight image, 32 | cin, 16-byte aligned pixels)?"""
    if cw.k not in (1, 3) or cw.w3 is None or x.c % 32 or x.ld % 4 or (x.buf.data_ptr() + 4 * x.off) % 16:
        return False
    w3, parts = cw.split_image()
    return w3 is not None and parts == 2


def conv2d_gn(x, cw, out=None, gn=None, want_stats=False, eps=1e-5, act=0, slope=0.0, residual=None):
    """3x3 / 1x1 convolution, stride 1, with GroupNorm(32) handed over between kernels: gn = (gamma, beta) applies GroupNorm + ReLU to
    x from x.stats (left by x's producer); want_stats leaves the statistics of the output (after activation and residual) in
    out.stats."""
    if out is None:
        out = Img(x.h, x.w, cw.cout, device=x.buf.device)
    assert x.c == cw.cin and (out.h, out.w, out.c) == (x.h, x.w, cw.cout)
    w3, parts = cw.split_image()
    assert parts == 2
    if gn is not None and x.stats is None:
        raise RuntimeError("conv2d_gn: the input carries no GroupNorm statistics")
    cap = (out.h * out.w + 127) // 128 if cw.k == 1 else ((out.w + 31) // 32) * ((out.h + 3) // 4)
    st = GnStats(cap, x.buf.device) if want_stats else None
    slots = C.c_int(0)
    check(lib().surs_conv2d_nhwc_gn(1 if cw.reduced else 2, x.ptr(), x.h, x.w, x.c, x.ld, _ptr(w3), _ptr(cw.b), out.ptr(), cw.cout, out.ld,
                                    cw.k, 1, _ptr(x.stats.buf) if gn is not None else None, x.stats.slots if gn is not None else 0,
                                    _ptr(gn[0]) if gn is not None else None, _ptr(gn[1]) if gn is not None else None, eps, act, slope,
                                    residual.ptr() if residual is not None else None, residual.ld if residual is not None else 0,
                                    _ptr(st.buf) if st is not None else None, st.buf.numel() // 64 if st is not None else 0,
                                    C.byref(slots), _stream()))
    if st is not None:
        st.slots = slots.value
    out.stats = st
    return out


def _ew_stats(n_items, device):
    return GnStats(min(512, (n_items + 1023) // 1024), device)   # (workgroups of 1024 threads, at most 512 of them)


def groupnorm_coeffs(x, gamma, beta, groups=32, eps=1e-5):
    scale = torch.empty(x.c, dtype=torch.float32, device=x.buf.device)
    shift = torch.empty_like(scale)
    # (the partial sums' scratch from the caching allocator: nothing is allocated inside the library, so the launches can be
    #  captured into a HIP graph - encoder.py)
    scratch = torch.empty(lib().surs_groupnorm_scratch_bytes(), dtype=torch.uint8, device=x.buf.device)
    check(lib().surs_groupnorm_coeffs_ws(x.ptr(), x.h * x.w, x.c, x.ld, groups, eps, _ptr(gamma), _ptr(beta), _ptr(scale),
                                         _ptr(shift), _ptr(scratch), _stream()))
    return scale, shift


def scale_shift_act(x, scale, shift, relu, out=None):
    out = out or Img(x.h, x.w, x.c, device=x.buf.device)
    check(lib().surs_scale_shift_act(x.ptr(), x.h * x.w, x.c, x.ld, _ptr(scale), _ptr(shift), int(relu), out.ptr(), out.ld,
                                     _stream()))
    out.stats = None   # (whatever GroupNorm statistics travelled with `out` described other values)
    return out


def avgpool2(x, out=None, want_stats=False):
    out = out or Img(x.h // 2, x.w // 2, x.c, device=x.buf.device)
    if want_stats:
        st, slots = _ew_stats(out.h * out.w * (x.c // 4), x.buf.device), C.c_int(0)
        check(lib().surs_avgpool2_gn(x.ptr(), x.h, x.w, x.c, x.ld, out.ptr(), out.ld, _ptr(st.buf), st.buf.numel() // 64, C.byref(slots),
                                     _stream()))
        st.slots = slots.value
        out.stats = st
        return out
    check(lib().surs_avgpool2(x.ptr(), x.h, x.w, x.c, x.ld, out.ptr(), out.ld, _stream()))
    out.stats = None
    return out


def bicubic_up2(x, align_corners, addend=None, out=None, want_stats=False):
    out = out or Img(2 * x.h, 2 * x.w, x.c, device=x.buf.device)
    if want_stats:
        st, slots = _ew_stats(out.h * out.w * (x.c // 4), x.buf.device), C.c_int(0)
        check(lib().surs_bicubic_up2_gn(x.ptr(), x.h, x.w, x.c, x.ld, int(bool(align_corners)),
                                        addend.ptr() if addend is not None else None, addend.ld if addend is not None else 0,
                                        out.ptr(), out.ld, _ptr(st.buf), st.buf.numel() // 64, C.byref(slots), _stream()))
        st.slots = slots.value
        out.stats = st
        return out
    out.stats = None
    check(lib().surs_bicubic_up2(x.ptr(), x.h, x.w, x.c, x.ld, int(bool(align_corners)),
                                 addend.ptr() if addend is not None else None, addend.ld if addend is not None else 0,
                                 out.ptr(), out.ld, _stream()))
    return out


def pixel_shuffle2(x, slope, out=None):
    out = out or Img(2 * x.h, 2 * x.w, x.c // 4, device=x.buf.device)
    check(lib().surs_pixel_shuffle2(x.ptr(), x.h, x.w, x.c, x.ld, slope, out.ptr(), out.ld, _stream()))
    out.stats = None
    return out


def add3(a, b, c=None, out=None, want_stats=False):
    out = out or Img(a.h, a.w, a.c, device=a.buf.device)
    if want_stats:
        st, slots = _ew_stats(a.h * a.w * (a.c // 4), a.buf.device), C.c_int(0)
        check(lib().surs_add3_gn(a.ptr(), a.ld, b.ptr(), b.ld, c.ptr() if c is not None else None, c.ld if c is not None else 0,
                                 a.h * a.w, a.c, out.ptr(), out.ld, _ptr(st.buf), st.buf.numel() // 64, C.byref(slots), _stream()))
        st.slots = slots.value
        out.stats = st
        return out
    check(lib().surs_add3(a.ptr(), a.ld, b.ptr(), b.ld, c.ptr() if c is not None else None, c.ld if c is not None else 0,
                          a.h * a.w, a.c, out.ptr(), out.ld, _stream()))
    out.stats = None
    return out


# ------------------------------------------------------------------ point evaluator

def pack_mlp(sd, dtype, device):
    """Pack mlp_lr / mlp_hr from a (numpy) state dict into the device blob of surs_mlp_pack."""
    keep = []

    def arrs(prefix):
        ws, bs = (C.c_void_p * 5)(), (C.c_void_p * 5)()
        for l in range(5):
            w = np.ascontiguousarray(np.asarray(sd[prefix + "conv%d.weight" % l], np.float32).reshape(
                np.asarray(sd[prefix + "conv%d.weight" % l]).shape[0], -1))
            b = np.ascontiguousarray(np.asarray(sd[prefix + "conv%d.bias" % l], np.float32))
            keep.extend([w, b])
            ws[l], bs[l] = w.ctypes.data, b.ctypes.data
        return ws, bs

    expect = {"mlp_lr.": [(1024, 321), (512, 1024), (256, 833), (128, 577), (1, 449)],
              "mlp_hr.": [(1024, 322), (512, 1024), (256, 834), (128, 578), (1, 450)]}
    for prefix, shapes in expect.items():
        for l, s in enumerate(shapes):
            got = tuple(np.asarray(sd[prefix + "conv%d.weight" % l]).shape[:2])
            if got != s:
                raise ValueError("unsupported SurfaceClassifier shape %s%d: %s (the kernels are built for the "
                                 "reference's default mlp_dim / res_layers)" % (prefix, l, got))
    wl, bl = arrs("mlp_lr.")
    wh, bh = arrs("mlp_hr.")
    code = DTYPES[dtype] if isinstance(dtype, str) else dtype
    core = BF16 if code in (F32, _lib.F32_GEMM) else code
    n = lib().surs_mlp_pack(wl, bl, wh, bh, core, None)
    host = np.zeros(n, np.uint8)
    lib().surs_mlp_pack(wl, bl, wh, bh, core, host.ctypes.data_as(C.c_void_p))
    return torch.from_numpy(host).to(device), core


class Workspace:
    """Grow-only device scratch buffer; remembers mesh capacities between extractions."""

    def __init__(self, device):
        self.device = device
        self.buf = None
        self.mc_capacity = {}     # key -> (verts, faces) of that field's last extraction: sizes the next one without a counting pass
        self.mesh_ws = {}         # key -> private marching-cubes workspace of an incremental extraction (MeshStream)

    def get(self, nbytes):
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = None
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self.buf

    def to_host(self, tensors):
        """device tensors -> numpy arrays backed by pinned host memory (one sync for the whole batch).  Each array owns
        its pinned block (torch's caching host allocator recycles it once the array is garbage collected), so the
        results stay valid like the reference's freshly allocated arrays, without a second host copy."""
        outs = []
        for t in tensors:
            if t is None:
                outs.append(None)
                continue
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            outs.append(h)
        torch.cuda.current_stream().synchronize()
        return [o.numpy() if o is not None else None for o in outs]

    def to_host_async(self, tensors):
        """Same copies on a side stream, ordered after the work enqueued so far; returns a HostCopy."""
        return _to_host_async(self, tensors)


class HostCopy:
    """Device -> pinned host copies in flight on the workspace's copy stream (Workspace.to_host_async); result() waits
    for them and returns the numpy arrays.  Lets the copy of one mesh run under the marching cubes of the next."""

    def __init__(self, hosts, done, keep):
        self._hosts, self._done, self._keep = hosts, done, keep

    def result(self):
        self._done.synchronize()
        self._keep = None
        return [h.numpy() if h is not None else None for h in self._hosts]


_shared_streams = {}


def shared_stream(device, name):
    """The process's side stream `name` on `device` (copy stream, one marching-cubes stream per field): ONE set per device, shared
    by every Workspace.  HIP maps streams onto a handful of hardware queues; a second SuRSNet object with side streams of its own
    (bench.py's fp32 leg) found its copy stream on the hardware queue of the sweep, and the 1 GB of mesh copies that should travel
    under the sweep followed it instead: + 19 ms per 512^3 reconstruction (round 4, tools/diag/second_net.py)."""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), name)
    st = _shared_streams.get(key)
    if st is None:
        st = _shared_streams[key] = torch.cuda.Stream(device=dev)
    return st


def _to_host_async(ws, tensors):
    cur = torch.cuda.current_stream()
    side = shared_stream(ws.device, "copy")
    ready = torch.cuda.Event()
    ready.record(cur)
    side.wait_event(ready)
    hosts = []
    with torch.cuda.stream(side):
        for t in tensors:
            if t is None:
                hosts.append(None)
                continue
            t.record_stream(side)   # the caching allocator must not hand the block out again before the copy ran
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            hosts.append(h)
        done = torch.cuda.Event()
        done.record(side)
    return HostCopy(hosts, done, list(tensors))


def query_points(points, calib, zmul, zdiv, feat_lr, feat_hr, blob, ws, want_logits=False):
    """points [3,N] f32 device tensor; calib: 12 floats (host); feat_*: Img with ld == c.  Returns pred_hr, pred_lr[, logits]."""
    points = _f32c(points)
    n = points.shape[1]
    dev = points.device
    outs = [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(4 if want_logits else 2)]
    cal = (C.c_float * 12)(*[float(v) for v in calib])
    need = lib().surs_query_workspace_bytes(n)
    w = ws.get(need)
    assert feat_lr.ld == feat_lr.c == 256 and feat_hr.ld == feat_hr.c == 64
    check(lib().surs_query_points(_ptr(points), n, cal, float(zmul), float(zdiv), feat_lr.ptr(), feat_lr.h, feat_lr.w,
                                  feat_hr.ptr(), feat_hr.h, feat_hr.w, _ptr(blob), _ptr(w), w.numel(), _ptr(outs[0]),
                                  _ptr(outs[1]), _ptr(outs[2]) if want_logits else None,
                                  _ptr(outs[3]) if want_logits else None, _stream()))
    return tuple(outs)


def set_option(name, value):
    """surs_set_option: a library option by name (include/surs.h; `options()` lists them)."""
    check(lib().surs_set_option(name.encode(), int(value)))


def get_option(name):
    v = C.c_int(0)
    check(lib().surs_get_option(name.encode(), C.byref(v)))
    return v.value


def options():
    """{name: (value, help)} of every library option."""
    out, i = {}, 0
    while True:
        name = lib().surs_option_name(i)
        if name is None:
            return out
        out[name.decode()] = (get_option(name.decode()), lib().surs_option_help(i).decode())
        i += 1


def any_nonfinite(a, b=None):
    """surs_nonfinite: True if the float32 device tensors a / b hold a NaN or an infinity (one small launch, one 4-byte read-back)."""
    a = _f32c(a.reshape(-1))
    if b is not None:
        b = _f32c(b.reshape(-1))
        assert b.numel() == a.numel()
    if a.numel() == 0:
        return False
    flag = torch.empty(1, dtype=torch.int32, device=a.device)
    check(lib().surs_nonfinite(_ptr(a), _ptr(b) if b is not None else None, a.numel(), _ptr(flag), _stream()))
    return bool(flag.item())


POINT_RUNS_CHUNK = 262144   # points per surs_query_points_columns call


def query_points_columns(points, calib, zmul, zdiv, feat_lr, feat_hr, blob, dtype, ws):
    """surs_query_points_columns: both classifiers on points [3,N] THROUGH THE COLUMN KERNELS where the points come as runs that share
    their (x, y) - what the reference's sweep loop (lib/sdf.py:32-45: 50 000 consecutive grid points per call, z fastest) hands the
    facade.  dtype: the blob's ("fp32": kernel v11, fp32-grade; "bf16" / "fp16": kernel v10).  Returns (pred_hr, pred_lr), or None
    when the array holds no such runs (random samples; a general calibration; fewer than 2048 points; inside wide_operands()) - the
    caller then takes query_points.  One host synchronisation (the run count)."""
    if wide_operands_active() or settings.get("SURS_POINT_RUNS") == "0":
        return None
    points = _f32c(points)
    n = points.shape[1]
    if n < 2048:
        return None
    dev = points.device
    phr = torch.empty(n, dtype=torch.float32, device=dev)
    plr = torch.empty(n, dtype=torch.float32, device=dev)
    cal = (C.c_float * 12)(*[float(v) for v in calib])
    code = DTYPES[dtype] if isinstance(dtype, str) else dtype
    w = ws.get(lib().surs_query_points_columns_workspace_bytes())
    assert feat_lr.ld == feat_lr.c == 256 and feat_hr.ld == feat_hr.c == 64
    ncols = C.c_int(0)
    for p0 in range(0, n, POINT_RUNS_CHUNK):
        nb = min(POINT_RUNS_CHUNK, n - p0)
        check(lib().surs_query_points_columns(C.c_void_p(points.data_ptr() + 4 * p0), n, nb, cal, float(zmul), float(zdiv),
                                              feat_lr.ptr(), feat_lr.h, feat_lr.w, feat_hr.ptr(), feat_hr.h, feat_hr.w, _ptr(blob),
                                              code, _ptr(w), w.numel(), C.c_void_p(phr.data_ptr() + 4 * p0),
                                              C.c_void_p(plr.data_ptr() + 4 * p0), C.byref(ncols), _stream()))
        if ncols.value == 0:
            if p0 == 0 and nb == n:
                return None
            # this piece holds no runs: the point kernels for it, in the arithmetic the rest of the array gets (one product per
            # MAC behind --precision bf16 | fp16: the column kernel's pieces are 16-bit there too)
            with reduced_point_operands(code != DTYPES["fp32"]):
                a, b = query_points(points[:, p0:p0 + nb], calib, zmul, zdiv, feat_lr, feat_hr, blob, ws)
            phr[p0:p0 + nb] = a
            plr[p0:p0 + nb] = b
    return phr, plr


def point_runs(points, tile=64):
    """surs_point_runs: the runs surs_query_points_columns would evaluate in points [3,N] (N <= POINT_RUNS_CHUNK) - numpy arrays
    (colstart, kcount, tiles [ntiles, 2], meta [4]); kcount / tiles are empty when the array holds more than one run per tile / 4 points."""
    points = _f32c(points)
    n = points.shape[1]
    dev = points.device
    cs, kc = torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)
    tl, meta = torch.empty(2 * n, dtype=torch.int32, device=dev), torch.zeros(4, dtype=torch.int32, device=dev)
    check(lib().surs_point_runs(_ptr(points), n, n, int(tile), _ptr(cs), _ptr(kc), _ptr(tl), _ptr(meta), _stream()))
    m = meta.cpu().numpy()
    nc, nt = int(m[0]), int(m[1])
    listed = nc * (int(tile) // 4) <= n
    return (cs[:nc].cpu().numpy(), kc[:nc].cpu().numpy() if listed else np.zeros(0, np.int32),
            tl[:2 * nt].cpu().numpy().reshape(-1, 2) if listed else np.zeros((0, 2), np.int32), m)


def query_points_hr(points, calib, zmul, zdiv, feat_lr, feat_hr, blob, ws, p_lr):
    """surs_query_points_hr: the hr classifier on points [3,N] with the lr occupancies p_lr [N] given (query_sr on points other than
    query_mr's).  Returns pred_hr [N]."""
    points = _f32c(points)
    n = points.shape[1]
    p_lr = _f32c(p_lr.reshape(-1))
    assert p_lr.numel() == n
    out = torch.empty(n, dtype=torch.float32, device=points.device)
    cal = (C.c_float * 12)(*[float(v) for v in calib])
    w = ws.get(lib().surs_query_workspace_bytes(n))
    check(lib().surs_query_points_hr(_ptr(points), n, cal, float(zmul), float(zdiv), feat_lr.ptr(), feat_lr.h, feat_lr.w,
                                     feat_hr.ptr(), feat_hr.h, feat_hr.w, _ptr(blob), _ptr(w), w.numel(), _ptr(p_lr), _ptr(out), None,
                                     _stream()))
    return out


def query_points_views(points, calibs, projection, zmul, zdiv, feat_lr, feat_hr, blob, ws, want_logits=False):
    """Multi-view / perspective query.  points [V,3,N] f32 device tensor; calibs [V,12] (host); feat_lr [V,hl,wl,256] and
    feat_hr [V,hh,wh,64] contiguous NHWC device tensors; projection 'orthogonal' | 'perspective'.
    Returns pred_hr [V,N], pred_lr [V,N][, logit_hr [N], logit_lr [N]]."""
    points = _f32c(points)
    V, _, n = points.shape
    dev = points.device
    feat_lr, feat_hr = _f32c(feat_lr), _f32c(feat_hr)
    assert feat_lr.shape[0] == V and feat_hr.shape[0] == V and feat_lr.shape[3] == 256 and feat_hr.shape[3] == 64
    phr = torch.empty((V, n), dtype=torch.float32, device=dev)
    plr = torch.empty_like(phr)
    lg = [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(2)] if want_logits else [None, None]
    cal = np.ascontiguousarray(np.asarray(calibs, np.float32).reshape(V, -1)[:, :12])
    cbuf = (C.c_float * (12 * V))(*[float(v) for v in cal.reshape(-1)])
    w = ws.get(lib().surs_query_views_workspace_bytes(n, V))
    check(lib().surs_query_points_views(_ptr(points), n, V, {"orthogonal": 0, "perspective": 1}[projection], cbuf, float(zmul),
                                        float(zdiv), _ptr(feat_lr), feat_lr.shape[1], feat_lr.shape[2], _ptr(feat_hr),
                                        feat_hr.shape[1], feat_hr.shape[2], _ptr(blob), _ptr(w), w.numel(), _ptr(phr), _ptr(plr),
                                        _ptr(lg[0]) if want_logits else None, _ptr(lg[1]) if want_logits else None, _stream()))
    return (phr, plr, lg[0], lg[1]) if want_logits else (phr, plr)


def query_grid_views(i0, i1, ry, rz, mat, calibs, projection, zmul, zdiv, feat_lr, feat_hr, blob, ws, vol_hr=None, vol_lr=None):
    """surs_query_grid_views: the dense sweep of a multi-view / perspective model over the grid slab [i0, i1) - every voxel seen by
    every view, view 0's predictions kept (lib/mesh_util.py:20-28).  calibs [V,12] (host); feat_lr [V,hl,wl,256], feat_hr [V,hh,wh,64]
    contiguous NHWC device tensors.  Returns (vol_hr, vol_lr) [(i1-i0), ry, rz]."""
    feat_lr, feat_hr = _f32c(feat_lr), _f32c(feat_hr)
    V = feat_lr.shape[0]
    dev = blob.device
    assert feat_hr.shape[0] == V and feat_lr.shape[3] == 256 and feat_hr.shape[3] == 64
    if vol_hr is None:
        vol_hr = torch.empty((i1 - i0, ry, rz), dtype=torch.float32, device=dev)
        vol_lr = torch.empty_like(vol_hr)
    m = (C.c_double * 12)(*[float(v) for v in np.asarray(mat, np.float64).reshape(-1)[:12]])
    cal = np.ascontiguousarray(np.asarray(calibs, np.float32).reshape(V, -1)[:, :12])
    cbuf = (C.c_float * (12 * V))(*[float(v) for v in cal.reshape(-1)])
    w = ws.get(lib().surs_query_grid_views_workspace_bytes(V))
    check(lib().surs_query_grid_views(i0, i1, ry, rz, m, V, {"orthogonal": 0, "perspective": 1}[projection], cbuf, float(zmul),
                                      float(zdiv), _ptr(feat_lr), feat_lr.shape[1], feat_lr.shape[2], _ptr(feat_hr), feat_hr.shape[1],
                                      feat_hr.shape[2], _ptr(blob), _ptr(w), w.numel(), _ptr(vol_hr), _ptr(vol_lr), _stream()))
    return vol_hr, vol_lr


def query_grid(i0, i1, ry, rz, mat, calib, zmul, zdiv, feat_lr, feat_hr, blob, dtype, ws, vol_hr=None, vol_lr=None, kernel=0,
               operand_parts=0):
    """Dense sweep of grid slab [i0, i1): returns (vol_hr, vol_lr) float32 device tensors [(i1-i0), ry, rz].
    kernel: column-kernel version for this call (grid_kernel_for's choice; 0 = the library's default / process setting);
    operand_parts: 2 / 3 = operand split of the fp32-grade GEMMs behind this sweep (0 = process setting).  Both travel in the
    call's SursGridOptions: no process-wide state is touched."""
    dev = blob.device
    if vol_hr is None:
        vol_hr = torch.empty((i1 - i0, ry, rz), dtype=torch.float32, device=dev)
        vol_lr = torch.empty_like(vol_hr)
    m = (C.c_double * 12)(*[float(v) for v in np.asarray(mat, np.float64).reshape(-1)[:12]])
    cal = (C.c_float * 12)(*[float(v) for v in calib])
    code = DTYPES[dtype] if isinstance(dtype, str) else dtype
    need = lib().surs_query_grid_workspace_bytes(ry, rz, code)
    w = ws.get(need)
    opt = _lib.GridOptions(int(kernel), int(operand_parts))
    check(lib().surs_query_grid_opt(i0, i1, ry, rz, m, cal, float(zmul), float(zdiv), feat_lr.ptr(), feat_lr.h, feat_lr.w,
                                    feat_hr.ptr(), feat_hr.h, feat_hr.w, _ptr(blob), code, _ptr(w), w.numel(), _ptr(vol_hr),
                                    _ptr(vol_lr), C.byref(opt), _stream()))
    return vol_hr, vol_lr


# mean listed channels per tile above which the dense column kernels are the faster ones (profiles/r03_listed_sensitivity.json:
# the eight-wave bf16 / fp16 kernel still wins at 490 listed - 254 against 290 ms at 512^3 - and gains 0.3 ms per listed channel)
LISTED_DENSE_THRESHOLD = 400.0
LISTED_DENSE_THRESHOLDS = {"fp32": 400.0, "bf16": 600.0, "fp16": 600.0}
# bf16 / fp16: mean listed channels per tile up to which the streamed two-workgroup kernel (12) is used; above it the eight-wave kernel
# (10), whose chunks of 96 channels loop: kernel 12 stages 144 channels per tile and hands fuller tiles to kernel 10 one by one, which
# pays only while they are rare (bench field: 70 listed, one tile in 10^4; layer-0 depth gain 16: 212 listed, 264 against 233 ms)
LISTED_STREAM_THRESHOLD = 96.0


def probe_listed(i_plane, ry, rz, tile, mat, calib, zmul, zdiv, feat_lr, feat_hr, blob, ws):
    """surs_query_grid_probe: (mean listed layer-0 channels per z tile for the lr classifier, upper bound for hr), or
    (-1, -1) where the column kernels do not apply.  Synchronises the stream."""
    m = (C.c_double * 12)(*[float(v) for v in np.asarray(mat, np.float64).reshape(-1)[:12]])
    cal = (C.c_float * 12)(*[float(v) for v in calib])
    need = lib().surs_query_grid_workspace_bytes(ry, rz, DTYPES["bf16"])
    w = ws.get(need)
    out = (C.c_float * 2)(-1.0, -1.0)
    check(lib().surs_query_grid_probe(int(i_plane), ry, rz, tile, m, cal, float(zmul), float(zdiv), feat_lr.ptr(), feat_lr.h,
                                      feat_lr.w, feat_hr.ptr(), feat_hr.h, feat_hr.w, _ptr(blob), _ptr(w), w.numel(), out, _stream()))
    return float(out[0]), float(out[1])


def grid_kernel_for(rx, ry, rz, mat, calib, zmul, zdiv, feat_lr, feat_hr, blob, dtype, ws):
    """Column-kernel version for a sweep of the rx x ry x rz grid `mat` describes.  fp32: 0 (the library's default, 11: layer 1
    restated along the column) unless the probe says the sweep lists so many channels per tile that the dense kernel (5) is
    faster.  bf16 / fp16: 12 (restated, layer 1 streamed into layer 2, two workgroups per CU) while the probe's mean stays under
    LISTED_STREAM_THRESHOLD, 10 (restated, eight waves) above it, 3 (dense) above LISTED_DENSE_THRESHOLDS.  A deterministic function of the grid, the calibration, the features and
    the weights - the middle axis-0 plane of the WHOLE grid is probed, so every slab and every rank of a sharded sweep makes
    the same choice.  Probed on every call (one plane of column constants, about 0.3 ms and a stream synchronisation): the
    feature buffers are written through raw pointers into recycled allocator blocks, so nothing the host can see tells one
    subject's features from the next one's.  SURS_GRID_AUTO=0 or an explicit SURS_GRID_KERNEL / SURS_GRID_F32_KERNEL turn it
    off.  The last decision is left in ws.kernel_choice = (kernel, listed channels per tile) for reports."""
    if settings.get("SURS_GRID_AUTO") == "0" or settings.is_set("SURS_GRID_KERNEL") or settings.is_set("SURS_GRID_F32_KERNEL"):
        return 0
    if dtype not in ("fp32", "bf16", "fp16") or ry > 16384:
        return 0
    lr, _ = probe_listed(rx // 2, ry, rz, 64 if dtype == "fp32" else 128, mat, calib, zmul, zdiv, feat_lr, feat_hr, blob, ws)
    kern = (5 if dtype == "fp32" else 3) if lr > LISTED_DENSE_THRESHOLDS[dtype] else 0
    if kern == 0 and dtype != "fp32":
        kern = 12 if lr <= LISTED_STREAM_THRESHOLD else 10
    ws.kernel_choice = (kern, lr)
    ws.probes = getattr(ws, "probes", 0) + 1
    return kern


# ------------------------------------------------------------------ marching cubes

def marching_cubes_lewiner(vol, level, ws, want_normals=True, key=None):
    """vol: float32 device tensor [n0,n1,n2].  Returns device tensors (verts [V,3] f32, faces [F,3] i32, normals, values).
    Raises ValueError / RuntimeError like skimage.measure.marching_cubes_lewiner."""
    if vol.dim() != 3:
        raise ValueError("Input volume should be a 3D numpy array.")
    if min(vol.shape) < 2:
        raise ValueError("Input array must be at least 2x2x2.")
    vol = _f32c(vol.to(torch.float32).contiguous())
    n0, n1, n2 = vol.shape
    w = ws.get(lib().surs_mc_workspace_bytes(n0, n1, n2))
    counts = _lib.McCounts()
    dev = vol.device

    def run(cap_v, cap_f):
        verts = torch.empty((cap_v, 3), dtype=torch.float32, device=dev)
        faces = torch.empty((cap_f, 3), dtype=torch.int32, device=dev)
        normals = torch.empty((cap_v, 3), dtype=torch.float32, device=dev) if want_normals else None
        values = torch.empty((cap_v,), dtype=torch.float32, device=dev) if want_normals else None
        rc = lib().surs_mc_lewiner(_ptr(vol), n0, n1, n2, float(level), _ptr(w), w.numel(), _ptr(verts), _ptr(normals),
                                   _ptr(values), cap_v, _ptr(faces), cap_f, C.byref(counts), _stream())
        return rc, verts, faces, normals, values

    if ws.mc_capacity.get(key) is None:
        # first extraction with this workspace: counting pass (no outputs) to size the buffers
        check(lib().surs_mc_lewiner(_ptr(vol), n0, n1, n2, float(level), _ptr(w), w.numel(), None, None, None, 0, None, 0,
                                    C.byref(counts), _stream()))
        cap = (counts.n_verts, counts.n_faces)
    else:
        cap = ws.mc_capacity[key]
    rc, verts, faces, normals, values = run(*cap)
    if rc == -6:   # SURS_E_CAPACITY: the counts are filled in, retry with exact sizes
        rc, verts, faces, normals, values = run(counts.n_verts, counts.n_faces)
    check(rc)
    nv, nf = counts.n_verts, counts.n_faces
    ws.mc_capacity[key] = mesh_capacity(nv, nf)
    _warm_stream_buffers(ws, key)
    if key is not None and (ws.mesh_ws.get(key) is None or ws.mesh_ws[key].numel() < w.numel()):
        ws.mesh_ws[key] = torch.empty(w.numel(), dtype=torch.uint8, device=dev)   # the field's own tables (MeshStream)
    return (verts[:nv], faces[:nf], normals[:nv] if normals is not None else None,
            values[:nv] if values is not None else None)


def mesh_capacity(nv, nf):
    """Buffer sizes for the next extraction of a field that last had nv vertices / nf faces: 12.5 % head-room, rounded up
    to 2^20 vertices / 2^21 faces so that meshes of similar size ask the caching allocators (device and pinned host) for
    blocks of the same size and get the previous reconstruction's blocks back."""
    gv, gf = 1 << 20, 1 << 21
    return (-(-(int(nv * 1.125) + 1024) // gv) * gv, -(-(int(nf * 1.125) + 2048) // gf) * gf)


def _warm_stream_buffers(ws, key):
    """The streamed extraction (MeshStream) keeps its results in pinned host buffers of the capacity sizes.  Allocating
    pinned memory is slow (17 - 140 ms for the bench's 1 GB, depending on the box) and torch's caching host allocator only
    recycles blocks of a matching size: allocate and release them here, in the one-piece extraction that sizes the
    buffers, so that the first streamed reconstruction already finds them cached."""
    cap = ws.mc_capacity[key]
    warm = getattr(ws, "_pinned_warm", None)
    if warm is None:
        warm = ws._pinned_warm = {}
    if warm.get(key) == cap:
        return
    a = torch.empty((cap[0], 3), dtype=torch.float64, pin_memory=True)
    b = torch.empty((cap[1], 3), dtype=torch.int32, pin_memory=True)
    # ... and the device buffers of a MeshStream (hipMalloc is slow too, and synchronises)
    d = [torch.empty((cap[0], 3), dtype=torch.float32, device=ws.device), torch.empty((cap[0], 3), dtype=torch.float64, device=ws.device),
         torch.empty((cap[1], 3), dtype=torch.int32, device=ws.device), torch.empty((cap[0], 3), dtype=torch.float32, device=ws.device),
         torch.empty((cap[0],), dtype=torch.float32, device=ws.device)]
    del a, b, d
    warm[key] = cap
    # ... and the field's extraction stream: the first use of a HIP stream creates its hardware queue (milliseconds)
    known = len(_shared_streams)
    st = shared_stream(ws.device, ("mc", key))
    if len(_shared_streams) > known:
        with torch.cuda.stream(st):
            torch.zeros(1, device=ws.device).cpu()   # (a pageable read-back like the extraction's counts)


class MeshStream:
    """Incremental Lewiner extraction of ONE field while the dense sweep is still writing it (surs_mc_lewiner_range): after
    every advance() the new vertices are transformed to world space and they and the new faces start their way to
    pinned host memory on the workspace's copy stream, under the sweep launches that follow.  Buffers are sized from the
    field's previous extraction (ws.mc_capacity[key]); `overflow` is set if that turns out too small - the caller then
    extracts the finished volume in one piece."""

    def __init__(self, ws, key, vol, mat, level=0.5, want_normals=True, zoff=None):
        """zoff (slab mode, dist.reconstruction_sharded): `vol` is one axis-0 slab of a larger grid whose plane 0 is the grid's
        plane zoff and whose last plane is the next slab's first (halo); vertices and faces stay on the device, numbered
        locally, faces of the first cell layer referring to the slab below as surs_mc_lewiner_range_slab describes;
        finish() then returns device tensors and leaves the level-range / no-surface checks to the caller."""
        self.ws, self.key, self.vol, self.level, self.want = ws, key, vol, float(level), want_normals
        self.zoff = zoff
        if zoff is not None and want_normals:
            raise NotImplementedError("slab mode extracts vertices and faces only (normals accumulate across slabs)")
        self.n0, self.n1, self.n2 = vol.shape
        dev = vol.device
        self.cap_v, self.cap_f = ws.mc_capacity[key]
        need = lib().surs_mc_workspace_bytes(self.n0, self.n1, self.n2)
        w = ws.mesh_ws.get(key)
        if w is None or w.numel() < need:
            w = ws.mesh_ws[key] = torch.empty(int(need), dtype=torch.uint8, device=dev)   # holds the edge tables between calls
        self.w = w
        self.verts = torch.empty((self.cap_v, 3), dtype=torch.float32, device=dev)
        self.world = torch.empty((self.cap_v, 3), dtype=torch.float64, device=dev)
        self.faces = torch.empty((self.cap_f, 3), dtype=torch.int32, device=dev)
        self.normals = torch.empty((self.cap_v, 3), dtype=torch.float32, device=dev) if want_normals else None
        self.values = torch.empty((self.cap_v,), dtype=torch.float32, device=dev) if want_normals else None
        if zoff is None:
            self.h_world = torch.empty((self.cap_v, 3), dtype=torch.float64, pin_memory=True)
            self.h_faces = torch.empty((self.cap_f, 3), dtype=torch.int32, pin_memory=True)
        self.mat = (C.c_double * 12)(*[float(v) for v in np.asarray(mat, np.float64).reshape(-1)[:12]])
        self.run = _lib.McCounts(0, 0, 3.4028234663852886e38, -3.4028234663852886e38)
        self.layers = 0          # cell layers (axis 0) extracted so far
        self.sent_v = self.sent_f = 0
        self.overflow = False
        self.side = shared_stream(dev, "copy")
        # the extraction runs on a stream of its own (one per field): its kernels and its count read-backs (host syncs)
        # then neither sit between two launches of the sweep nor keep the host from enqueueing the next launch
        self.mc = shared_stream(dev, ("mc", key))
        self.mc.wait_stream(torch.cuda.current_stream(dev))   # the buffers above were allocated on the caller's stream
        for t in (self.world, self.faces):
            t.record_stream(self.side)
        for t in (self.verts, self.world, self.faces, self.normals, self.values, vol):
            if t is not None:
                t.record_stream(self.mc)

    def advance(self, layer_end, after=None):
        """Extract the cell layers [self.layers, layer_end): the voxel planes up to layer_end must be final - or final once
        the event `after` (recorded on the sweep's stream) has happened."""
        with torch.cuda.stream(self.mc):
            if after is not None:
                self.mc.wait_event(after)
            self._advance(layer_end)

    def _advance(self, layer_end):
        layer_end = min(int(layer_end), self.n0 - 1)
        if self.overflow or layer_end <= self.layers:
            return
        if self.zoff is not None:
            rc = lib().surs_mc_lewiner_range_slab(_ptr(self.vol), self.n0, self.n1, self.n2, self.layers, layer_end, self.level,
                                                  _ptr(self.w), self.w.numel(), _ptr(self.verts), self.cap_v, _ptr(self.faces),
                                                  self.cap_f, C.byref(self.run), int(self.zoff), _stream())
        else:
            rc = lib().surs_mc_lewiner_range(_ptr(self.vol), self.n0, self.n1, self.n2, self.layers, layer_end, self.level,
                                             _ptr(self.w), self.w.numel(), _ptr(self.verts), _ptr(self.normals), _ptr(self.values),
                                             self.cap_v, _ptr(self.faces), self.cap_f, C.byref(self.run), _stream())
        if rc == -6:
            self.overflow = True
            return
        check(rc)
        self.layers = layer_end
        nv, nf = self.run.n_verts, self.run.n_faces
        if nv > self.sent_v:
            check(lib().surs_transform_points(self.verts[self.sent_v:].data_ptr(), nv - self.sent_v, self.mat,
                                              self.world[self.sent_v:].data_ptr(), _stream()))
        if self.zoff is None and (nv > self.sent_v or nf > self.sent_f):
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
            self.side.wait_event(ready)
            with torch.cuda.stream(self.side):
                if nv > self.sent_v:
                    self.h_world[self.sent_v:nv].copy_(self.world[self.sent_v:nv], non_blocking=True)
                if nf > self.sent_f:
                    self.h_faces[self.sent_f:nf].copy_(self.faces[self.sent_f:nf], non_blocking=True)
        self.sent_v, self.sent_f = nv, nf

    def finish(self, after=None):
        """Last layers, the checks of marching_cubes_lewiner, normals; -> (verts_world, faces, normals, values) numpy."""
        with torch.cuda.stream(self.mc):
            if after is not None:
                self.mc.wait_event(after)
            return self._finish()

    def _finish(self):
        self._advance(self.n0 - 1)
        if self.overflow:
            return None
        if self.zoff is not None:   # slab mode: device results, local numbering; the caller checks the whole grid's range
            nv, nf = self.run.n_verts, self.run.n_faces
            self.ws.mc_capacity[self.key] = mesh_capacity(nv, nf)
            torch.cuda.current_stream().synchronize()
            return self.world[:nv], self.faces[:nf]
        if self.level < self.run.vmin or self.level > self.run.vmax:
            raise ValueError("Surface level must be within volume data range.")
        nv, nf = self.run.n_verts, self.run.n_faces
        if nv == 0:
            raise RuntimeError("No surface found at the given iso value.")
        self.ws.mc_capacity[self.key] = mesh_capacity(nv, nf)
        extra = [None, None]
        if self.want:
            check(lib().surs_mc_normalize(_ptr(self.normals), nv, _stream()))
            extra = self.ws.to_host([self.normals[:nv], self.values[:nv]])
        self.side.synchronize()
        return self.h_world[:nv].numpy(), self.h_faces[:nf].numpy(), extra[0], extra[1]


def slab_mesh_one_piece(ws, key, vol, mat, level, zoff):
    """Slab-mode extraction of a finished slab in one piece (the first reconstruction of a workspace, or after a streamed
    extraction ran out of buffer): a counting call sizes the buffers.  Returns device tensors (verts_world float64 [V,3],
    faces int32 [F,3] in the slab's local numbering), the counts and the workspace that holds the edge tables."""
    n0, n1, n2 = vol.shape
    dev = vol.device
    need = lib().surs_mc_workspace_bytes(n0, n1, n2)
    w = ws.mesh_ws.get(key)
    if w is None or w.numel() < need:
        w = ws.mesh_ws[key] = torch.empty(int(need), dtype=torch.uint8, device=dev)
    m = (C.c_double * 12)(*[float(v) for v in np.asarray(mat, np.float64).reshape(-1)[:12]])

    def run(cap_v, cap_f):
        verts = torch.empty((max(cap_v, 1), 3), dtype=torch.float32, device=dev)
        faces = torch.empty((max(cap_f, 1), 3), dtype=torch.int32, device=dev)
        counts = _lib.McCounts(0, 0, 3.4028234663852886e38, -3.4028234663852886e38)
        rc = lib().surs_mc_lewiner_range_slab(_ptr(vol), n0, n1, n2, 0, n0 - 1, float(level), _ptr(w), w.numel(), _ptr(verts), cap_v,
                                              _ptr(faces), cap_f, C.byref(counts), int(zoff), _stream())
        return rc, verts, faces, counts

    cap = ws.mc_capacity.get(key) or (0, 0)
    rc, verts, faces, counts = run(*cap)
    if rc == -6:
        rc, verts, faces, counts = run(counts.n_verts, counts.n_faces)
    check(rc)
    nv, nf = counts.n_verts, counts.n_faces
    ws.mc_capacity[key] = mesh_capacity(nv, nf)
    world = torch.empty((nv, 3), dtype=torch.float64, device=dev)
    if nv:
        check(lib().surs_transform_points(_ptr(verts), nv, m, _ptr(world), _stream()))
    return world, faces[:nf], counts, w


def transform_points(verts, mat):
    """float64 [V,3] device tensor = mat[:3,:3] @ v + mat[:3,3] (the reference's np.matmul step, on the device)."""
    out = torch.empty((verts.shape[0], 3), dtype=torch.float64, device=verts.device)
    m = (C.c_double * 12)(*[float(v) for v in np.asarray(mat, np.float64).reshape(-1)[:12]])
    check(lib().surs_transform_points(_ptr(verts), verts.shape[0], m, _ptr(out), _stream()))
    return out


# ------------------------------------------------------------------ octree sweep (lib/sdf.py:55-120)

def octree_volumes(R, mat, calib, zmul, zdiv, feat_lr, feat_hr, blob, ws, threshold, init_resolution=64, num_samples=None,
                   evaluate=None, device=None, columns=None, stats=None, dtype="fp32"):
    """eval_grid_octree on the device: float64 volumes (sdf_hr, sdf_lr) [R,R,R] like the reference's arrays.
    Host code only walks the levels; selection, evaluation (fp32-grade kernels), scatter and the cell pass are kernels.
    Single view, axis-aligned sweep (gen_mesh's): every level runs on the sweep's COLUMN kernel (surs_octree_level_columns: the
    lattice points of a level form columns of constant image position; a column's dirty points are evaluated 64 at a time); general
    calibrations, `columns=False` and native.wide_operands() (the retry after an f16 overflow: the column kernel carries f16
    parts) take the per-point layer kernels (surs_octree_select + surs_query_grid_indexed + surs_octree_scatter).
    evaluate(idx int64 device tensor [n]) -> (pred_hr, pred_lr) float32 [n] replaces the single-view evaluator
    (mesh_util passes the multi-view / perspective query there).
    stats: a list that receives (reso, dirty lattice points evaluated, lattice columns, 64-point tiles) per level (None, None on
    the per-point path).  dtype: "fp32" (the fp32-grade column kernel v11) | "bf16" | "fp16" (the 16-bit column kernel v10 on a blob
    packed for that precision - the octree sweep of `--precision bf16 | fp16`); the per-point fallbacks are fp32-grade in every case."""
    dev = device if device is not None else blob.device
    n3 = R * R * R
    sdf_hr = torch.zeros(n3, dtype=torch.float64, device=dev)
    sdf_lr = torch.zeros(n3, dtype=torch.float64, device=dev)
    dirty = torch.ones(n3, dtype=torch.uint8, device=dev)
    cnt_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    m = (C.c_double * 12)(*[float(v) for v in np.asarray(mat, np.float64).reshape(-1)[:12]])
    cal = (C.c_float * 12)(*[float(v) for v in calib]) if evaluate is None else None
    reso = R // init_resolution
    batch = 262144
    use_cols = evaluate is None and columns is not False and not wide_operands_active() \
        and settings.get("SURS_OCTREE_COLUMNS") != "0" and (R + max(reso, 1) - 1) // max(reso, 1) <= 2048
    while reso > 0:
        if use_cols:
            w = ws.get(lib().surs_octree_columns_workspace_bytes(R))
            counts = (C.c_longlong * 3)(0, 0, 0)
            rc = lib().surs_octree_level_columns_dt(_ptr(sdf_hr), _ptr(sdf_lr), _ptr(dirty), R, reso, R // 2, m, cal, float(zmul),
                                                    float(zdiv), feat_lr.ptr(), feat_lr.h, feat_lr.w, feat_hr.ptr(), feat_hr.h,
                                                    feat_hr.w, _ptr(blob), DTYPES[dtype] if isinstance(dtype, str) else dtype, _ptr(w),
                                                    w.numel(), counts, _stream())
            if rc == -3 and columns is None:
                use_cols = False     # general calibration: the per-point kernels (nothing has been written yet)
                continue
            check(rc)
            if stats is not None:
                stats.append((reso, int(counts[0]), int(counts[1]), int(counts[2])))
        else:
            nl = (R + reso - 1) // reso
            cap = nl * nl * nl
            idx = torch.empty(cap, dtype=torch.int64, device=dev)
            cnt = C.c_int(0)
            check(lib().surs_octree_select(_ptr(dirty), R, reso, _ptr(idx), cap, _ptr(cnt_dev), C.byref(cnt), _stream()))
            n = cnt.value
            phr = torch.empty(max(n, 1), dtype=torch.float32, device=dev)
            plr = torch.empty(max(n, 1), dtype=torch.float32, device=dev)
            w = ws.get(lib().surs_query_workspace_bytes(min(n, batch))) if evaluate is None else None
            for b0 in range(0, n, batch):
                nb = min(batch, n - b0)
                if evaluate is not None:
                    a, b = evaluate(idx[b0:b0 + nb])
                    phr[b0:b0 + nb] = a
                    plr[b0:b0 + nb] = b
                    continue
                check(lib().surs_query_grid_indexed(C.c_void_p(idx.data_ptr() + 8 * b0), nb, R, R, m, cal, float(zmul), float(zdiv),
                                                    feat_lr.ptr(), feat_lr.h, feat_lr.w, feat_hr.ptr(), feat_hr.h, feat_hr.w,
                                                    _ptr(blob), _ptr(w), w.numel(), C.c_void_p(phr.data_ptr() + 4 * b0),
                                                    C.c_void_p(plr.data_ptr() + 4 * b0), _stream()))
            check(lib().surs_octree_scatter(_ptr(idx), n, _ptr(phr), _ptr(plr), _ptr(sdf_hr), _ptr(sdf_lr), _ptr(dirty), _stream()))
            if stats is not None:
                stats.append((reso, n, None, None))
        if reso <= 1:
            break
        w = ws.get(lib().surs_octree_workspace_bytes(R, reso))
        check(lib().surs_octree_cells(_ptr(sdf_hr), _ptr(sdf_lr), _ptr(dirty), R, reso, float(threshold), _ptr(w), w.numel(),
                                      _stream()))
        reso //= 2
    return sdf_hr.view(R, R, R), sdf_lr.view(R, R, R)


def octree_level_values(R, reso, idx, mat, calib, zmul, zdiv, feat_lr, feat_hr, blob, ws, columns=True, dtype="fp32"):
    """What the octree sweep assigns to the lattice points `idx` (flat voxel indices, int64 device tensor, all on the lattice of
    stride reso) at level `reso`: (pred_hr, pred_lr) float32.  columns=True: the column kernel of surs_octree_level_columns, run on
    a walk state in which exactly `idx` is dirty (a point's last bits there depend on which points of its column share its tile:
    ask for a level's whole dirty set to get the walk's bits); False: the per-point layer kernels.  The checker of
    tests/test_gpu_octree.py drives the oracle's restatement of lib/sdf.py:55-120 with it."""
    dev = idx.device
    n = idx.numel()
    m = (C.c_double * 12)(*[float(v) for v in np.asarray(mat, np.float64).reshape(-1)[:12]])
    cal = (C.c_float * 12)(*[float(v) for v in calib])
    if not columns:
        phr = torch.empty(max(n, 1), dtype=torch.float32, device=dev)
        plr = torch.empty_like(phr)
        batch = 262144
        w = ws.get(lib().surs_query_workspace_bytes(min(max(n, 1), batch)))
        for b0 in range(0, n, batch):
            nb = min(batch, n - b0)
            check(lib().surs_query_grid_indexed(C.c_void_p(idx.data_ptr() + 8 * b0), nb, R, R, m, cal, float(zmul), float(zdiv),
                                                feat_lr.ptr(), feat_lr.h, feat_lr.w, feat_hr.ptr(), feat_hr.h, feat_hr.w, _ptr(blob),
                                                _ptr(w), w.numel(), C.c_void_p(phr.data_ptr() + 4 * b0),
                                                C.c_void_p(plr.data_ptr() + 4 * b0), _stream()))
        return phr[:n], plr[:n]
    # a scratch walk state in which exactly the asked points are dirty: the level call then evaluates their tiles and writes them
    n3 = R * R * R
    hr = torch.zeros(n3, dtype=torch.float64, device=dev)
    lr = torch.zeros(n3, dtype=torch.float64, device=dev)
    dirty = torch.zeros(n3, dtype=torch.uint8, device=dev)
    dirty[idx] = 1
    w = ws.get(lib().surs_octree_columns_workspace_bytes(R))
    check(lib().surs_octree_level_columns_dt(_ptr(hr), _ptr(lr), _ptr(dirty), R, reso, R // 2, m, cal, float(zmul), float(zdiv),
                                             feat_lr.ptr(), feat_lr.h, feat_lr.w, feat_hr.ptr(), feat_hr.h, feat_hr.w, _ptr(blob),
                                             DTYPES[dtype] if isinstance(dtype, str) else dtype, _ptr(w), w.numel(), None, _stream()))
    return hr[idx].float(), lr[idx].float()


def f64_to_f32(a):
    out = torch.empty(a.shape, dtype=torch.float32, device=a.device)
    check(lib().surs_f64_to_f32(_ptr(a), _ptr(out), a.numel(), _stream()))
    return out
