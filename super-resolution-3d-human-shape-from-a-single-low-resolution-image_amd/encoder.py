"""Image encoder orchestration: which kernel runs on which tensor, in the reference's order.

Host code only sequences calls into the C ABI (native.py); all arithmetic is in csrc/surs_encoder.hip.
Mirrors, layer for layer:
  SuRSSR_v3.forward           /root/reference/lib/model/SuRSSR_v3.py:143-181   (ResBlock: lib/model/common.py:14-33)
  HGFilter.forward low_res    lib/model/HGFilters.py:183-206  (ConvBlock :57-74, HourGlass :96-117)
  HGFilter.forward high_res   lib/model/HGFilters.py:179-181
torch.cat is never materialised: producers write into channel slices of the concatenated tensor.
"""
import ctypes as C
import numpy as np
import os

import torch

from . import native, settings
from .native import Img


_weights_serial = iter(range(1, 1 << 62))


class EncoderWeights:
    """Device-side packed weights of every conv / GroupNorm that is live at eval time."""

    def __init__(self, sd, opt, device):
        self.serial = next(_weights_serial)   # the key of this object's captured graphs: never reused, unlike id()
        self.device = device
        self.conv = {}
        self.gn = {}
        self.opt = opt
        # --encoder_precision: "fp32" (= "auto", the default) = two f16 parts, three products per MAC: the parity-grade encoder, whatever
        # --precision says about the classifiers (BASELINE configs[2] / [4] name a bf16 / fp16 MLP, not a reduced encoder); "f16" = one
        # f16 product in the 3x3 convolutions (features within 1.8e-3 / 4e-4 of their range, 2 ms of 7 per 512^2 image): opt-in only.
        # Measured at 512^3 (tools/precision_report.py encoder, tests/test_gpu_precision.py) it adds a quarter to the bf16 sweep's own
        # error (mean |d logit| 1.0e-3 -> 1.3e-3) and would multiply the fp16 sweep's by nine (6e-5 -> 5.9e-4).  (Round 4 made it the
        # default of --precision bf16; round 5 took that back.)
        ep = getattr(opt, "encoder_precision", "auto")
        self.reduced = ep == "f16"

        def get(k):
            v = sd[k]
            return v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)

        def add_conv(name, bias=True):
            self.conv[name] = native.ConvWeights(get(name + ".weight"), get(name + ".bias") if bias else None, device,
                                                 reduced=self.reduced)

        def add_gn(name):
            self.gn[name] = (torch.from_numpy(np.ascontiguousarray(get(name + ".weight"), np.float32)).to(device),
                             torch.from_numpy(np.ascontiguousarray(get(name + ".bias"), np.float32)).to(device))

        def add_block(prefix):
            for c in ("conv1", "conv2", "conv3"):
                add_conv(prefix + c, bias=False)
            for g in ("bn1", "bn2", "bn3"):
                add_gn(prefix + g)

        S = "super_resolution."
        names = ["head.0", "bottleneck.0", "bott2.0", "ups2.0", "ups3.0", "ups4.0", "last.0", "last.2"]
        for i, nb in zip((1, 2, 3), opt.n_block):
            names += ["down%d.0" % i, "tail%d.0" % i, "tail%d.2" % i]
            for b in range(nb):
                names += ["body%d.%d.body.0" % (i, b), "body%d.%d.body.2" % (i, b)]
        for n in names:
            add_conv(S + n)
        add_conv("image_filter_hr.conv5")
        L = "image_filter_lr."
        add_block(L + "conv2.")
        for s in range(opt.num_stack_lr):
            def gen(level):
                add_block(L + "m%d.b1_%d." % (s, level))
                add_block(L + "m%d.b2_%d." % (s, level))
                if level > 1:
                    gen(level - 1)
                else:
                    add_block(L + "m%d.b2_plus_%d." % (s, level))
                add_block(L + "m%d.b3_%d." % (s, level))
            gen(opt.hg_depth)
            add_block(L + "top_m_%d." % s)
            add_conv(L + "conv_last%d" % s)
            add_gn(L + "bn_end%d" % s)
            add_conv(L + "l%d" % s)
            if s < opt.num_stack_lr - 1:
                add_conv(L + "bl%d" % s)
                add_conv(L + "al%d" % s)
                # previous + bl(t) + al(l(t)) (HGFilters.py:203-206) is ONE pointwise convolution of t - nothing non-linear sits
                # between l and al: W = W_bl + W_al W_l, b = b_bl + W_al b_l + b_al, formed in float64 (filter_lr)
                wl, bl_ = get(L + "l%d.weight" % s).astype(np.float64)[:, :, 0, 0], get(L + "l%d.bias" % s).astype(np.float64)
                wa, ba = get(L + "al%d.weight" % s).astype(np.float64)[:, :, 0, 0], get(L + "al%d.bias" % s).astype(np.float64)
                wb, bb = get(L + "bl%d.weight" % s).astype(np.float64)[:, :, 0, 0], get(L + "bl%d.bias" % s).astype(np.float64)
                self.conv[L + "next%d" % s] = native.ConvWeights((wb + wa @ wl).astype(np.float32)[:, :, None, None],
                                                                 (bb + wa @ bl_ + ba).astype(np.float32), device, reduced=self.reduced)


# ------------------------------------------------------------------ the networks sequenced inside the library (default)
# csrc/surs_encoder_net.cpp runs the launches of super_res / filter_lr / filter_hr below - the same kernels, order, tiles and bits -
# from ONE C call per network: no ctypes hop per launch (~ 160 per image), intermediates in one workspace tensor instead of a tensor
# per map.  The Python sequencing below stays as the readable mirror, as the reference the native one is held to bit for bit
# (tests/test_gpu_encoder_net.py) and as the path of the non-default forms: wide operands (the retry after an f16 overflow), the
# three-part bf16 split, SURS_ENC_FUSED_GN=0, captured graphs, SURS_ENC_NATIVE=0.
class NativeNet:
    """SursEncoderNet (include/surs.h) of an EncoderWeights: device pointers of its packed weights, in the header's order."""

    def __init__(self, W):
        from . import _lib
        opt = W.opt
        self.keep = []   # ctypes arrays behind the struct's pointer fields

        def conv(name):
            cw = W.conv[name]
            w3 = cw.w3 if cw.parts == 2 else None
            return _lib.Conv(w3.data_ptr() if w3 is not None else None, cw.w.data_ptr(), cw.b.data_ptr() if cw.b is not None else None,
                             cw.cin, cw.cout, cw.k, 0)

        def gn(name):
            g, b = W.gn[name]
            return _lib.GroupNorm(g.data_ptr(), b.data_ptr())

        def block(prefix):
            return _lib.ConvBlock((_lib.Conv * 3)(*[conv(prefix + "conv%d" % i) for i in (1, 2, 3)]),
                                  (_lib.GroupNorm * 3)(*[gn(prefix + "bn%d" % i) for i in (1, 2, 3)]))

        def array(ctype, items):
            a = (ctype * max(1, len(items)))(*items)
            self.keep.append(a)
            return a

        n = _lib.EncoderNet()
        n.residual = 1 if opt.residual else 0
        n.n_block = (C.c_int * 3)(*[int(v) for v in opt.n_block])
        n.num_stack, n.hg_depth, n.parts = int(opt.num_stack_lr), int(opt.hg_depth), 1 if W.reduced else 2
        # SURS_ENC_SEPARATE_SUM=1: a ConvBlock's closing sum as a launch of its own (the form whose bits the sequencing below
        # reproduces: tests/test_gpu_encoder_net.py); default: in the three convolutions' epilogues (surs_conv2d_nhwc_gn_sum)
        n.flags = 1 if settings.get("SURS_ENC_SEPARATE_SUM") != "0" else 0
        S = "super_resolution."
        n.head = conv(S + "head.0")
        n.down = (_lib.Conv * 3)(*[conv(S + "down%d.0" % i) for i in (1, 2, 3)])
        n.tail0 = (_lib.Conv * 3)(*[conv(S + "tail%d.0" % i) for i in (1, 2, 3)])
        n.tail2 = (_lib.Conv * 3)(*[conv(S + "tail%d.2" % i) for i in (1, 2, 3)])
        for f, k in (("bottleneck", "bottleneck.0"), ("bott2", "bott2.0"), ("ups2", "ups2.0"), ("ups3", "ups3.0"), ("ups4", "ups4.0"),
                     ("last0", "last.0"), ("last2", "last.2")):
            setattr(n, f, conv(S + k))
        body = []
        for i, nb in zip((1, 2, 3), opt.n_block):
            for b in range(nb):
                body += [conv(S + "body%d.%d.body.0" % (i, b)), conv(S + "body%d.%d.body.2" % (i, b))]
        n.body = array(_lib.Conv, body)
        n.conv5 = conv("image_filter_hr.conv5")
        L = "image_filter_lr."
        n.conv2 = block(L + "conv2.")
        hg = []
        for s in range(opt.num_stack_lr):
            def gen(level):
                hg.append(block(L + "m%d.b1_%d." % (s, level)))
                hg.append(block(L + "m%d.b2_%d." % (s, level)))
                if level > 1:
                    gen(level - 1)
                else:
                    hg.append(block(L + "m%d.b2_plus_%d." % (s, level)))
                hg.append(block(L + "m%d.b3_%d." % (s, level)))
            gen(opt.hg_depth)
        n.hg = array(_lib.ConvBlock, hg)
        S_ = range(opt.num_stack_lr)
        n.top_m = array(_lib.ConvBlock, [block(L + "top_m_%d." % s) for s in S_])
        n.conv_last = array(_lib.Conv, [conv(L + "conv_last%d" % s) for s in S_])
        n.l = array(_lib.Conv, [conv(L + "l%d" % s) for s in S_])
        n.next = array(_lib.Conv, [conv(L + "next%d" % s) if s < opt.num_stack_lr - 1 else _lib.Conv() for s in S_])
        n.bn_end = array(_lib.GroupNorm, [gn(L + "bn_end%d" % s) for s in S_])
        self.net = n
        self.last_ch = W.conv[L + "l0"].cout
        self.ws = None

    def workspace(self, h, w, device):
        """The calls' workspace: one tensor per stream the encoder runs on (two encoders side by side - gen_mesh_pipelined - must not
        share intermediates), grown on demand."""
        need = native.lib().surs_encoder_workspace_bytes(C.byref(self.net), h, w)
        if need == 0:
            raise ValueError("surs_encoder_workspace_bytes refused a %dx%d image" % (h, w))
        if self.ws is None:
            self.ws = {}
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        t = self.ws.get(key)
        if t is None or t.numel() < need:
            self.ws.pop(key, None)
            t = self.ws[key] = torch.empty(need, dtype=torch.uint8, device=device)
        return t


def native_enabled(W):
    """The library's own sequencing applies: default operand split (two f16 parts, or the one-part opt-in), statistics handed from kernel
    to kernel, not inside wide_operands(), no captured graphs, SURS_ENC_NATIVE != 0."""
    return (settings.get("SURS_ENC_NATIVE") != "0" and not native.wide_operands_active() and native.fused_groupnorm()
            and settings.get("SURS_CONV_SPLIT").startswith("f") and settings.get("SURS_CONV_X3") != "0"
            and not graphs_enabled(W) and not torch.cuda.is_current_stream_capturing())


def _native_net(W):
    nn = getattr(W, "_native", None)
    if nn is None:
        nn = W._native = NativeNet(W)
    return nn


def _lent_streams(depth):
    """SursEncoderStreams: the hourglass levels' side streams, lent only to an encoder on the device's default stream (hourglass());
    created in the order the Python sequencing creates them (the deepest level first: the stream -> hardware-queue mapping depends on
    creation order, NOTES R4.4)."""
    from . import _lib
    cur = torch.cuda.current_stream()
    if settings.get("SURS_ENC_STREAMS") == "0" or cur.cuda_stream != torch.cuda.default_stream(cur.device).cuda_stream:
        return None, None
    arr = (C.c_void_p * 4)()
    sides = []
    for level in range(min(depth, 4), 0, -1):
        st = _side_stream(level)
        arr[level - 1] = st.cuda_stream
        sides.append(st)
    return _lib.EncoderStreams(arr), sides


def super_res_native(W, x, want_image=True):
    """super_res() as one library call (surs_encoder_super_res): same outputs, bit for bit."""
    if x.h % 4 or x.w % 4:
        raise ValueError("input image height/width must be multiples of 4 (three stride-2 stages), got %dx%d" % (x.h, x.w))
    nn, dev = _native_net(W), x.buf.device
    ws = nn.workspace(x.h, x.w, dev)
    new2 = Img(x.h // 2, x.w // 2, 256, device=dev)
    new_fin = Img(2 * x.h, 2 * x.w, 64, device=dev)
    img_sr = Img(2 * x.h, 2 * x.w, 3, device=dev) if want_image else None
    native.check(native.lib().surs_encoder_super_res(C.byref(nn.net), x.ptr(), x.h, x.w, x.ld, 1 if want_image else 0,
                                                     img_sr.ptr() if want_image else None, new2.ptr(), new_fin.ptr(), native._ptr(ws),
                                                     ws.numel(), native._stream()))
    return img_sr, new2, new_fin


def filter_lr_native(W, feature_lr, keep_all=False):
    """filter_lr() as one library call (surs_encoder_filter_lr)."""
    opt = W.opt
    if feature_lr.h % (1 << opt.hg_depth) or feature_lr.w % (1 << opt.hg_depth):
        raise ValueError("feature_lr size must be a multiple of 2^hg_depth")
    nn, dev = _native_net(W), feature_lr.buf.device
    ws = nn.workspace(2 * feature_lr.h, 2 * feature_lr.w, dev)
    S = opt.num_stack_lr
    outs = [Img(feature_lr.h, feature_lr.w, nn.last_ch, device=dev) if (keep_all or s == S - 1) else None for s in range(S)]
    ptrs = (C.c_void_p * S)(*[o.ptr() if o is not None else None for o in outs])
    ss, sides = _lent_streams(opt.hg_depth)   # (forked from and joined to the call's stream by events inside the call)
    native.check(native.lib().surs_encoder_filter_lr(C.byref(nn.net), feature_lr.ptr(), feature_lr.h, feature_lr.w, feature_lr.ld, ptrs,
                                                     native._ptr(ws), ws.numel(), C.byref(ss) if ss is not None else None, native._stream()))
    return [o for o in outs if o is not None]


LRELU = dict(act=1, slope=0.2)
RELU = dict(act=1, slope=0.0)


def super_res(W, x, want_image=True):
    """x: Img [H,W,3].  Returns (img_SR [2H,2W,3], new2 = feature_lr [H/2,W/2,256], new_fin = feature_hr [2H,2W,64]).
    want_image=False: img_SR (two convolutions at 2H x 2W that nothing on the reconstruction path reads: lib/train_util.py:57 drops
    it) is not computed and None is returned in its place - for callers that do not hand it out."""
    opt, P, cv = W.opt, "super_resolution.", native.conv2d
    if x.h % 4 or x.w % 4:
        raise ValueError("input image height/width must be multiples of 4 (three stride-2 stages), got %dx%d" % (x.h, x.w))
    dev = x.buf.device
    H2, W2 = 2 * x.h, 2 * x.w
    fin = Img(H2, W2, 64, device=dev)             # cat(h, up3)
    new3 = Img(x.h, x.w, 128, device=dev)          # cat(d1_f, up2)
    new2 = Img(x.h // 2, x.w // 2, 256, device=dev)  # cat(d2_f, up1)   -> feature_lr
    new1 = Img(x.h // 4, x.w // 4, 512, device=dev)  # cat(d3_f, bo)
    up = native.bicubic_up2(x, False)
    h = cv(up, W.conv[P + "head.0"], out=fin.slice(0, 32), **LRELU)

    def stage(i, src, dst):
        d = cv(src, W.conv[P + "down%d.0" % i], stride=2, **LRELU)
        if opt.residual:
            for b in range(opt.n_block[i - 1]):
                t = cv(d, W.conv[P + "body%d.%d.body.0" % (i, b)], **RELU)
                d = cv(t, W.conv[P + "body%d.%d.body.2" % (i, b)], residual=d)
        d = cv(d, W.conv[P + "tail%d.0" % i], **LRELU)
        return cv(d, W.conv[P + "tail%d.2" % i], out=dst, **LRELU)

    d1_f = stage(1, h, new3.slice(0, 64))
    d2_f = stage(2, d1_f, new2.slice(0, 128))
    d3_f = stage(3, d2_f, new1.slice(0, 256))
    cv(d3_f, W.conv[P + "bottleneck.0"], out=new1.slice(256, 256), **LRELU)
    # conv -> LeakyReLU -> PixelShuffle -> LeakyReLU (the second LeakyReLU is fused into the shuffle)
    native.pixel_shuffle2(cv(new1, W.conv[P + "bott2.0"], **LRELU), 0.2, out=new2.slice(128, 128))
    native.pixel_shuffle2(cv(new2, W.conv[P + "ups2.0"], **LRELU), 0.2, out=new3.slice(64, 64))
    native.pixel_shuffle2(cv(new3, W.conv[P + "ups3.0"], **LRELU), 0.2, out=fin.slice(32, 32))
    new_fin = cv(fin, W.conv[P + "ups4.0"], **LRELU)
    img_sr = cv(cv(new_fin, W.conv[P + "last.0"], **LRELU), W.conv[P + "last.2"]) if want_image else None
    return img_sr, new2, new_fin


# The super-resolution net has no normalisation: an output column depends on input columns within its receptive field only -
# in columns of the 2W-wide maps: bicubic x2 (4) + head (1) + down1 (1) + 6 convolutions at W (12) + down2 (2) + 6 at W/2 (24) +
# down3 (4) + 8 at W/4 (64) + ups2 (4) + ups3 (2) + ups4 (1) = 119.  Rounded up to a multiple of 8 (the three stride-2 stages and
# the pixel shuffles keep their phase when a strip starts at a multiple of 8): 128 columns = 32 columns of feature_lr.
SR_HALO_LR = 32


def super_res_strip(W, x, a, b, want_image=True):
    """super_res restricted to the columns [a, b) of feature_lr (= columns [4a, 4b) of feature_hr / img_SR): runs the net on the
    image columns that range depends on (a halo of SR_HALO_LR feature_lr columns on each side, clipped at the image border, where
    the convolutions' own zero padding applies) and returns (img_sr, new2, new_fin) cropped to the range.  Every value is computed
    from the same inputs by the same instruction sequence as in super_res on the whole image: the strips are BIT-IDENTICAL to
    the corresponding columns of the full maps (tests/test_gpu_dist.py).  a, b even; what one rank of a sharded reconstruction
    computes (dist.encode_sharded)."""
    wl = x.w // 2
    if a % 2 or b % 2 or not (0 <= a < b <= wl):
        raise ValueError("strip [%d, %d) of %d feature_lr columns: bounds must be even and inside the map" % (a, b, wl))
    a0, b0 = max(0, a - SR_HALO_LR), min(wl, b + SR_HALO_LR)
    dev = x.buf.device
    hwc = lambda t: torch.as_strided(t.buf, (t.h, t.w, t.c), (t.w * t.ld, t.ld, 1), t.buf.storage_offset() + t.off)
    xs = Img(x.h, 2 * (b0 - a0), x.c, buf=hwc(x)[:, 2 * a0:2 * b0, :].contiguous().reshape(-1), device=dev)
    # (the 3x3 kernels pick their tile by the size of the map, and the two tiles sum in different orders: the strip runs the full
    #  image's tiles)
    native.check(native.lib().surs_conv_tile_scale(x.w, xs.w))
    try:
        # (the library's own sequencing where it applies: the tile scale is the calling thread's, whoever issues the launches)
        run = super_res_native if native_enabled(W) else super_res
        img_sr, new2, new_fin = run(W, xs, want_image=want_image)
    finally:
        native.check(native.lib().surs_conv_tile_scale(1, 1))

    def crop(t, scale):   # columns [scale * (a - a0), scale * (b - a0)) of a strip map with `scale` columns per feature_lr column
        v = hwc(t)[:, scale * (a - a0):scale * (b - a0), :]
        return Img(t.h, scale * (b - a), t.c, buf=v.contiguous().reshape(-1), device=dev)

    return (crop(img_sr, 4) if img_sr is not None else None), crop(new2, 1), crop(new_fin, 4)


def conv_block(W, prefix, x, want_stats=False):
    """ConvBlock with in_planes == out_planes: cat(o1, o2, o3) + x, GroupNorm+ReLU fused into each conv's staging.

    Four launches when x carries the statistics of its values (x.stats: the kernel that wrote x left them) and the block's
    convolutions run on the split-f16 3x3 kernel: each convolution folds its input's statistics into the GroupNorm coefficients
    itself and leaves its output's for the next one (native.conv2d_gn); want_stats: the closing sum leaves the block output's for
    the ConvBlock that follows.  Otherwise (SURS_ENC_FUSED_GN=0, wide operands, a first block fed by the super-resolution net)
    ten: two statistics launches in front of each convolution."""
    c = x.c
    out = Img(x.h, x.w, c, device=x.buf.device)
    o1, o2, o3 = out.slice(0, c // 2), out.slice(c // 2, c // 4), out.slice(3 * c // 4, c // 4)
    cw = [W.conv[prefix + "conv%d" % i] for i in (1, 2, 3)]
    fused = native.fused_groupnorm() and c % 128 == 0 and all(native.conv_gn_eligible(t, w) for t, w in zip((x, o1, o2), cw))
    if fused and x.stats is not None:
        native.conv2d_gn(x, cw[0], o1, gn=W.gn[prefix + "bn1"], want_stats=True)
        native.conv2d_gn(o1, cw[1], o2, gn=W.gn[prefix + "bn2"], want_stats=True)
        native.conv2d_gn(o2, cw[2], o3, gn=W.gn[prefix + "bn3"])
        return native.add3(out, x, out=out, want_stats=want_stats)
    sc, sh = native.groupnorm_coeffs(x, *W.gn[prefix + "bn1"])
    native.conv2d(x, W.conv[prefix + "conv1"], out=o1, in_scale=sc, in_shift=sh)
    sc, sh = native.groupnorm_coeffs(o1, *W.gn[prefix + "bn2"])
    native.conv2d(o1, W.conv[prefix + "conv2"], out=o2, in_scale=sc, in_shift=sh)
    sc, sh = native.groupnorm_coeffs(o2, *W.gn[prefix + "bn3"])
    native.conv2d(o2, W.conv[prefix + "conv3"], out=o3, in_scale=sc, in_shift=sh)
    return native.add3(out, x, out=out, want_stats=want_stats and fused)


_side_streams = {}


def _side_stream(level):
    """The stream the low-resolution branch of hourglass level `level` runs on, one per (device, calling stream, level): two
    encoders running side by side (gen_mesh_pipelined) do not share them."""
    cur = torch.cuda.current_stream()
    key = (cur.device.index, cur.cuda_stream, level)
    st = _side_streams.get(key)
    if st is None:
        # (high priority: the low-resolution branch is the longer one - chains of small kernels - and the full-resolution block beside
        #  it would otherwise take the CUs first; SURS_ENC_STREAM_PRIORITY=0: default priority)
        prio = -1 if settings.get("SURS_ENC_STREAM_PRIORITY") != "0" else 0
        st = _side_streams[key] = torch.cuda.Stream(device=cur.device, priority=prio)
    return st


def hourglass(W, prefix, depth, x):
    """HourGlass._forward (HGFilters.py:29-74).  The two branches of a level are independent until their sum: the low-resolution
    one (pool -> ConvBlock -> [next level] -> ConvBlock; maps of 128^2 and 64^2 whose kernels fill a fraction of the chip) runs on
    a side stream beside the full-resolution ConvBlock.  Buffers cross streams only at the fork and the join, both ordered by
    events; a side stream's next use starts by waiting for the calling stream, i.e. after every reader of what it freed."""
    # (only for an encoder that runs on the device's default stream: gen_mesh_pipelined runs the next subject's encoder on a second
    #  stream under the current sweep, where two more streams of small kernels cost the sweep more than they save - 6.6 against 7.4
    #  subjects/s at 512^3)
    cur0 = torch.cuda.current_stream()
    # (... or one that is being captured into a HIP graph - graphed() below -, where the fork and the join become graph edges)
    fork = settings.get("SURS_ENC_STREAMS") != "0" and (
        cur0.cuda_stream == torch.cuda.default_stream(cur0.device).cuda_stream or torch.cuda.is_current_stream_capturing())

    st = native.fused_groupnorm()   # every map a ConvBlock reads is written with its GroupNorm statistics (conv_block)

    capturing = torch.cuda.is_current_stream_capturing()

    def fwd(level, inp):
        if fork and capturing:
            # Captured into a HIP graph (graphed forms below): every fork leaves and rejoins the CAPTURE stream - the full-resolution
            # ConvBlock of a level goes to the side stream, the low-resolution chain (with the next level's fork inside it) stays.
            # A fork off a forked stream - the eager form's level 1 inside level 2's side stream - ends ROCm 7.2's
            # hipStreamEndCapture in a segmentation fault (tools/dev/graph_probe.py nested).
            cur, side = torch.cuda.current_stream(), _side_stream(level)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                up1 = conv_block(W, prefix + "b1_%d." % level, inp)
            low1 = conv_block(W, prefix + "b2_%d." % level, native.avgpool2(inp, want_stats=st), want_stats=True)
            low2 = fwd(level - 1, low1) if level > 1 else conv_block(W, prefix + "b2_plus_%d." % level, low1, want_stats=True)
            low3 = conv_block(W, prefix + "b3_%d." % level, low2)
            cur.wait_stream(side)
            return native.bicubic_up2(low3, True, addend=up1, want_stats=st)

        def low_branch():
            low1 = conv_block(W, prefix + "b2_%d." % level, native.avgpool2(inp, want_stats=st), want_stats=True)
            low2 = fwd(level - 1, low1) if level > 1 else conv_block(W, prefix + "b2_plus_%d." % level, low1, want_stats=True)
            return conv_block(W, prefix + "b3_%d." % level, low2)
        if fork:
            cur, side = torch.cuda.current_stream(), _side_stream(level)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                low3 = low_branch()
            up1 = conv_block(W, prefix + "b1_%d." % level, inp)
            cur.wait_stream(side)
        else:
            up1 = conv_block(W, prefix + "b1_%d." % level, inp)
            low3 = low_branch()
        return native.bicubic_up2(low3, True, addend=up1, want_stats=st)   # up1 + up2
    return fwd(depth, x)


def filter_lr(W, feature_lr, keep_all=False):
    """Returns the list of stack outputs (only the last one unless keep_all, as SuRSNet.filter_lr does in eval)."""
    opt, P = W.opt, "image_filter_lr."
    if feature_lr.h % (1 << opt.hg_depth) or feature_lr.w % (1 << opt.hg_depth):
        raise ValueError("feature_lr size must be a multiple of 2^hg_depth")
    previous = conv_block(W, P + "conv2.", feature_lr, want_stats=True)
    outs = []
    for i in range(opt.num_stack_lr):
        hg = hourglass(W, P + "m%d." % i, opt.hg_depth, previous)
        ll = conv_block(W, P + "top_m_%d." % i, hg)
        last = i == opt.num_stack_lr - 1
        if native.fused_groupnorm() and all(native.conv_gn_eligible(ll, W.conv[P + k % i]) for k in ("conv_last%d", "l%d")):
            # the tail of a stack in two launches (three where the stack's output is wanted): conv_last leaves bn_end's statistics,
            # the pointwise convolutions behind it fold them; the next stack's input from the merged convolution, its sum with
            # `previous` in the epilogue, the statistics for the next hourglass with it
            t = native.conv2d_gn(ll, W.conv[P + "conv_last%d" % i], want_stats=True)
            if last or keep_all:
                outs.append(native.conv2d_gn(t, W.conv[P + "l%d" % i], gn=W.gn[P + "bn_end%d" % i]))
            if not last:
                previous = native.conv2d_gn(t, W.conv[P + "next%d" % i], gn=W.gn[P + "bn_end%d" % i], residual=previous, want_stats=True)
            continue
        t = native.conv2d(ll, W.conv[P + "conv_last%d" % i])
        sc, sh = native.groupnorm_coeffs(t, *W.gn[P + "bn_end%d" % i])
        # ll = relu(bn_end(conv_last(ll))) is consumed only by 1x1 convs: fused into their staging
        tmp_out = native.conv2d(t, W.conv[P + "l%d" % i], in_scale=sc, in_shift=sh)
        outs.append(tmp_out)
        if i < opt.num_stack_lr - 1:
            bl = native.conv2d(t, W.conv[P + "bl%d" % i], in_scale=sc, in_shift=sh)
            al = native.conv2d(tmp_out, W.conv[P + "al%d" % i])
            previous = native.add3(previous, bl, al, want_stats=native.fused_groupnorm())
    return outs if keep_all else outs[-1:]


def filter_hr(W, feature_hr):
    return [native.conv2d(feature_hr, W.conv["image_filter_hr.conv5"])]


# ------------------------------------------------------------------ HIP graphs (opt-in: --encoder_graph 1 / SURS_ENC_GRAPH=1)
# The encoder is ~ 160 launches of fixed shapes per image, a third of them a few microseconds long on the low-resolution branch of
# the hourglasses, issued by Python through ctypes.  Captured ONCE per (weights, image size, options) into a HIP graph and replayed,
# the chain does not depend on the host between kernels and the fork / join of the hourglass levels are graph edges.
# MEASURED (round 5, one MI355X, 512^2 image, fp32-grade): 7.33 ms replayed against 7.17 ms eager (filter_lr 4.12 / 3.99) - the
# eager launches are asynchronous and already GPU-bound (back-to-back kernels on each stream in the rocprofv3 trace), the graph only
# removes the ~ 12 us event hand-overs between streams and pays for them with ROCm's own graph scheduling.  Hence opt-in.
# What a replay hands out are the graph's own buffers: the tensors of the previous call with the same key are OVERWRITTEN (the
# reference returns fresh tensors; everything on the reconstruction path consumes the features before the next image is encoded);
# gen_mesh_pipelined's second encoder (off the default stream) and multi-view batches always run eagerly.
_graphs = {}
_stable = {}   # input addresses filter_lr_g may capture on: the super-resolution graphs' feature_lr buffers and addresses seen before
GRAPH_CACHE = 8   # captured graphs kept per process (each holds the encoder's intermediates: ~ 1.5 GB at 512^2)


def graphs_enabled(W=None):
    """--encoder_graph 1 switches the graphs on for a network, SURS_ENC_GRAPH=1 / 0 for the process whatever the networks say."""
    env = settings.get("SURS_ENC_GRAPH")
    if env is not None:
        return env != "0"
    return W is not None and str(getattr(W.opt, "encoder_graph", "0")) != "0"


def drop_graphs(W=None):
    """Forget the captured graphs (of the weights W; all of them without an argument) and free their buffers."""
    for k in [k for k in _graphs if W is None or k[1] == W.serial]:
        del _graphs[k]
    _stable.clear()


def _graph_ok(W):
    cur = torch.cuda.current_stream()
    return (graphs_enabled(W) and not torch.cuda.is_current_stream_capturing()
            and cur.cuda_stream == torch.cuda.default_stream(cur.device).cuda_stream)


def _flags():
    return (native.wide_operands_active(), native.fused_groupnorm(), settings.get("SURS_ENC_STREAMS"))


def _hwc(t):
    return torch.as_strided(t.buf, (t.h, t.w, t.c), (t.w * t.ld, t.ld, 1), t.buf.storage_offset() + t.off)


def _run_graphed(key, build, static_in=None, x=None):
    """build(xin) -> outputs, captured on first use and replayed afterwards.  static_in: x is copied into a buffer of the graph's
    own first (inputs that arrive at a new address every call); otherwise the key holds x's address and the graph reads x in place."""
    ent = _graphs.get(key)
    if ent is None:
        if len(_graphs) >= GRAPH_CACHE:
            _graphs.pop(next(iter(_graphs)))
        xin = x
        if static_in:
            xin = Img(x.h, x.w, x.c, device=x.buf.device)
            _hwc(xin).copy_(_hwc(x))
        build(xin)                      # eager once: kernel attributes, lazily packed weights, side streams
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream(device=x.buf.device)
        with torch.cuda.stream(cap):       # (the hourglass levels' side streams of the capture stream exist before the capture starts)
            for level in range(1, 8):
                _side_stream(level)
        with torch.cuda.graph(g, stream=cap):
            outs = build(xin)
        ent = _graphs[key] = (g, xin, outs)
    g, xin, outs = ent
    if static_in:
        _hwc(xin).copy_(_hwc(x))
    g.replay()
    return outs


def super_res_g(W, x, want_image=True):
    """super_res through a captured graph where that is allowed (see above), eagerly otherwise."""
    if native_enabled(W):
        return super_res_native(W, x, want_image=want_image)
    if not _graph_ok(W):
        return super_res(W, x, want_image=want_image)
    key = ("sr", W.serial, x.h, x.w, x.c, want_image, _flags())
    outs = _run_graphed(key, lambda xin: super_res(W, xin, want_image=want_image), static_in=True, x=x)
    _stable[_addr_key(outs[1])] = outs[1].buf
    return outs


def super_res_strip_g(W, x, a, b, want_image=True):
    if not _graph_ok(W):
        return super_res_strip(W, x, a, b, want_image=want_image)
    key = ("srs", W.serial, x.h, x.w, x.c, a, b, want_image, _flags())
    return _run_graphed(key, lambda xin: super_res_strip(W, xin, a, b, want_image=want_image), static_in=True, x=x)


def _addr_key(t):
    return (t.buf.data_ptr() + 4 * t.off, t.h, t.w, t.c, t.ld)


def persistent(img):
    """Tells filter_lr_g that `img` lives at an address its caller keeps across calls (a gather buffer): captured at first sight."""
    _stable[_addr_key(img)] = img.buf


def filter_lr_g(W, feature_lr, keep_all=False):
    """filter_lr through a graph captured on the ADDRESS of its input (the super-resolution graph's feature_lr buffer, or a
    caller's persistent one): an input at a new address is captured anew, at most GRAPH_CACHE graphs are kept."""
    if native_enabled(W):
        return filter_lr_native(W, feature_lr, keep_all=keep_all)
    if not _graph_ok(W):
        return filter_lr(W, feature_lr, keep_all=keep_all)
    key = ("lr", W.serial, _addr_key(feature_lr), keep_all, _flags())
    if key not in _graphs and _addr_key(feature_lr) not in _stable:
        # an address seen for the first time and not a graph's own buffer: a caller that allocates per call would have every call
        # captured (an eager run + a capture each); the second sighting is taken as "persistent"
        _stable[_addr_key(feature_lr)] = feature_lr.buf
        if len(_stable) > 64:
            _stable.pop(next(iter(_stable)))
        return filter_lr(W, feature_lr, keep_all=keep_all)

    def build(xin):
        xin.stats = None
        return filter_lr(W, xin, keep_all=keep_all)
    return _run_graphed(key, build, static_in=False, x=feature_lr)


def filter_hr_g(W, feature_hr):
    return filter_hr(W, feature_hr)   # one launch: nothing to capture
