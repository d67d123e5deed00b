"""Parameter inventory of SuRSNet and deterministic synthetic weights.

`state_dict_spec(opt)` restates the reference's module tree as an ordered list of
(key, shape, kind) so that `load_state_dict` can be strict about all 553 keys
 without importing the reference:

  SuRSNet.__init__           /root/reference/lib/model/SuRSNet.py:44-99
  HGFilter / ConvBlock / HourGlass   lib/model/HGFilters.py:29-174
  SuRSSR_v3 / ResBlock / MeanShift   lib/model/SuRSSR_v3.py:30-141, lib/model/common.py:14-42
  SurfaceClassifier          lib/model/SurfaceClassifier.py:30-43

`synthetic_state_dict` fills it from the counter-based PRNG (prng.py): the
reference's init is normal(0, 0.02) on conv weights, zero bias, GroupNorm
weight 1 / bias 0 (lib/net_util.py:99-132).  With that init the occupancy field
sits in [0.447, 0.570] and barely crosses the 0.5 level and
activations shrink layer by layer, so we use a zero-mean uniform with a
fan-in scaled standard deviation instead; GroupNorm affine and biases get
non-trivial values so that parity tests exercise them.
"""
from collections import OrderedDict

import numpy as np

from . import prng


def _convblock(prefix, cin, cout):
    c2, c4 = cout // 2, cout // 4
    out = [
        (prefix + "conv1.weight", (c2, cin, 3, 3), "conv"),
        (prefix + "conv2.weight", (c4, c2, 3, 3), "conv"),
        (prefix + "conv3.weight", (c4, c4, 3, 3), "conv"),
    ]
    for name, c in (("bn1", cin), ("bn2", c2), ("bn3", c4), ("bn4", cin)):
        out += [(prefix + name + ".weight", (c,), "gn_w"), (prefix + name + ".bias", (c,), "gn_b")]
    if cin != cout:
        # nn.Sequential(self.bn4, ReLU, Conv2d 1x1 no bias): bn4 shows up twice in the state dict
        out += [(prefix + "downsample.0.weight", (cin,), "alias:" + prefix + "bn4.weight"),
                (prefix + "downsample.0.bias", (cin,), "alias:" + prefix + "bn4.bias"),
                (prefix + "downsample.2.weight", (cout, cin, 1, 1), "conv")]
    return out


def _hourglass(prefix, depth, feat):
    out = []

    def gen(level):
        out.extend(_convblock(prefix + "b1_%d." % level, feat, feat))
        out.extend(_convblock(prefix + "b2_%d." % level, feat, feat))
        if level > 1:
            gen(level - 1)
        else:
            out.extend(_convblock(prefix + "b2_plus_%d." % level, feat, feat))
        out.extend(_convblock(prefix + "b3_%d." % level, feat, feat))

    gen(depth)
    return out


def _conv(prefix, cout, cin, k, bias=True):
    out = [(prefix + ".weight", (cout, cin, k, k), "conv")]
    if bias:
        out.append((prefix + ".bias", (cout,), "bias"))
    return out


def _gn(prefix, c):
    return [(prefix + ".weight", (c,), "gn_w"), (prefix + ".bias", (c,), "gn_b")]


def _hgfilter(prefix, n_stack, depth, in_ch, last_ch, down_type):
    out = _conv(prefix + "conv1", 64, in_ch, 7) + _gn(prefix + "bn1", 64)
    if down_type == "low_res":
        out += _convblock(prefix + "conv2.", 256, 256)
    elif down_type == "high_res":
        out += _convblock(prefix + "conv2.", 64, 128)
    else:
        raise ValueError(down_type)
    out += _convblock(prefix + "conv3.", 128, 128)
    out += _convblock(prefix + "conv4.", 128, 256)
    out += _conv(prefix + "conv5", 64, 64, 1)
    for s in range(n_stack):
        out += _hourglass(prefix + "m%d." % s, depth, 256)
        out += _convblock(prefix + "top_m_%d." % s, 256, 256)
        out += _conv(prefix + "conv_last%d" % s, 256, 256, 1)
        out += _gn(prefix + "bn_end%d" % s, 256)
        out += _conv(prefix + "l%d" % s, last_ch, 256, 1)
        if s < n_stack - 1:
            out += _conv(prefix + "bl%d" % s, 256, 256, 1)
            out += _conv(prefix + "al%d" % s, 256, last_ch, 1)
    return out


def _sr(prefix, n_block):
    out = [(prefix + "sub_mean.weight", (3, 3, 1, 1), "meanshift_w"),
           (prefix + "sub_mean.bias", (3,), "meanshift_b-"),
           (prefix + "add_mean.weight", (3, 3, 1, 1), "meanshift_w"),
           (prefix + "add_mean.bias", (3,), "meanshift_b+")]
    out += _conv(prefix + "head.0", 32, 3, 3)

    def stage(i, cin, cout, nb):
        o = _conv(prefix + "down%d.0" % i, cin, cin, 3)
        for b in range(nb):
            o += _conv(prefix + "body%d.%d.body.0" % (i, b), cin, cin, 3)
            o += _conv(prefix + "body%d.%d.body.2" % (i, b), cin, cin, 3)
        o += _conv(prefix + "tail%d.0" % i, cin, cin, 3)
        o += _conv(prefix + "tail%d.2" % i, cout, cin, 3)
        return o

    out += stage(1, 32, 64, n_block[0])
    out += stage(2, 64, 128, n_block[1])
    out += stage(3, 128, 256, n_block[2])
    out += _conv(prefix + "bottleneck.0", 256, 256, 3)
    out += _conv(prefix + "bott2.0", 512, 512, 3)
    out += _conv(prefix + "ups2.0", 256, 256, 3)
    out += _conv(prefix + "ups3.0", 128, 128, 3)
    out += _conv(prefix + "ups4.0", 64, 64, 3)
    out += _conv(prefix + "last.0", 32, 64, 3)
    out += _conv(prefix + "last.2", 3, 32, 3)
    return out


def _mlp(prefix, dims, res_layers, no_residual):
    out = []
    for l in range(len(dims) - 1):
        cin = dims[l] + (dims[0] if (not no_residual and l in res_layers) else 0)
        out.append((prefix + "conv%d.weight" % l, (dims[l + 1], cin, 1), "mlp"))
        out.append((prefix + "conv%d.bias" % l, (dims[l + 1],), "mlp_bias"))
    return out


def state_dict_spec(opt):
    """Ordered [(key, shape, kind)] identical to reference SuRSNet(opt).state_dict()."""
    spec = []
    spec += _hgfilter("image_filter_lr.", opt.num_stack_lr, opt.hg_depth, 256, opt.hg_dim, "low_res")
    spec += _hgfilter("image_filter_hr.", opt.num_stack_hr, opt.hg_depth, 64, opt.hg_dim, "high_res")
    spec += _sr("super_resolution.", list(opt.n_block))
    spec += _mlp("mlp_lr.", list(opt.mlp_dim_lr), list(opt.mlp_res_layers_lr), opt.no_residual)
    spec += _mlp("mlp_hr.", list(opt.mlp_dim_hr), list(opt.mlp_res_layers_hr), opt.no_residual)
    return spec


_SQRT3 = 3.0 ** 0.5


def synthetic_state_dict(opt, seed=0, enc_gain=0.6, mlp_gain=1.0):
    """Deterministic weights as numpy float32 arrays, keyed like the reference.

    Conv / MLP weights are zero-mean uniform with std = gain * sqrt(2 / fan_in) (gain 0.6 for the
    encoder convolutions, whose residual/concat structure otherwise doubles the scale per stage) so that every
    layer's output is O(1) (parity tests then see every layer at full sensitivity and the
    occupancy field straddles the 0.5 level)."""
    sd = OrderedDict()
    for key, shape, kind in state_dict_spec(opt):
        if kind.startswith("alias:"):
            sd[key] = sd[kind[6:]]
            continue
        if kind in ("conv", "mlp"):
            fan_in = int(np.prod(shape[1:]))
            a = (enc_gain if kind == "conv" else mlp_gain) * (2.0 / fan_in) ** 0.5 * _SQRT3
            v = prng.uniform(key, seed, shape, -a, a)
        elif kind == "bias":
            v = prng.uniform(key, seed, shape, -0.05, 0.05)
        elif kind == "gn_w":
            v = prng.uniform(key, seed, shape, 0.7, 1.3)
        elif kind == "gn_b":
            v = prng.uniform(key, seed, shape, -0.1, 0.1)
        elif kind == "mlp_bias":
            v = prng.uniform(key, seed, shape, -0.1, 0.1)
        elif kind == "meanshift_w":
            v = np.eye(3, dtype=np.float32).reshape(3, 3, 1, 1)
        elif kind in ("meanshift_b-", "meanshift_b+"):
            sign = -1.0 if kind.endswith("-") else 1.0
            v = (sign * opt.rgb_range * np.array([0.4488, 0.4371, 0.4040], np.float32)).astype(np.float32)
        else:
            raise ValueError(kind)
        sd[key] = np.ascontiguousarray(v, dtype=np.float32)
    return sd


def synthetic_image(h, w=None, seed=1):
    """[1,3,H,W] float32 in [-1,1] times a centred rectangular mask
    (rows H/8..7H/8, cols W/4..3W/4): the output contract of
    EvalDataset_LR_v2.get_render (/root/reference/lib/data/EvalDataset_LR_v2.py:239-243)."""
    w = h if w is None else w
    img = prng.uniform("synthetic_image", seed, (1, 3, h, w), -1.0, 1.0)
    mask = np.zeros((1, 1, h, w), np.float32)
    mask[:, :, h // 8: 7 * h // 8, w // 4: 3 * w // 4] = 1.0
    return (img * mask).astype(np.float32)


def smooth_image(h, w=None, seed=1, waves=6):
    """[1,3,H,W] float32 in [-1,1] times the same mask as synthetic_image, but band-limited (a few low-frequency
    cosines with seeded phases) like a photograph rather than white noise: the occupancy field it induces has a
    surface of body-like extent instead of one crossing in nearly every voxel column.  Used by bench.py."""
    w = h if w is None else w
    p = prng.uniform("smooth_image", seed, (3, waves, 4), 0.0, 1.0).astype(np.float64)
    yy, xx = np.mgrid[:h, :w].astype(np.float64)
    yy, xx = yy / h, xx / w
    img = np.zeros((3, h, w))
    for c in range(3):
        for k in range(waves):
            fx, fy = 1 + 5 * p[c, k, 0], 1 + 5 * p[c, k, 1]
            img[c] += np.cos(2 * np.pi * (fx * xx + fy * yy + p[c, k, 2])) * (0.4 + 0.6 * p[c, k, 3])
    img = img / np.abs(img).max()
    mask = np.zeros((1, h, w))
    mask[:, h // 8: 7 * h // 8, w // 4: 3 * w // 4] = 1.0
    return (img * mask)[None].astype(np.float32)


def synthetic_points(n, seed=2, lo=-0.55, hi=0.55):
    """[3,N] float32 points; about 17 % fall outside the image for +-0.55."""
    return prng.uniform("synthetic_points", seed, (3, n), lo, hi)


# ------------------------------------------------------------------ a smooth, closed, body-sized occupancy field
# The seeded random weights above give a noise-like field (a level crossing in nearly every voxel column: ~100x the
# surface of a body).  Acceptance tests of the reduced-precision sweeps need the other regime too: a smooth field whose
# 0.5 level set is ONE closed blob of body-like extent, so that a shift of the level set shows up as a mesh distance.
# Both MLPs keep seeded random weights (at a reduced gain: a smooth perturbation of about +-1 in the logit) and get a
# hand-routed path on top:
#     logit = A * G(x, y)  -  B * |z_feat|  +  (random network)
# G = 1 - (x/ax)^2 - (y/ay)^2 is channel 0 of the low-resolution feature map and enters through the last layer's skip
# connection; |z_feat| is built by two layer-0 units, lrelu(z_feat) and lrelu(-z_feat), and carried to the last layer by
# one unit in each of layers 1-3.  The level set is a closed double cone over an ellipse, wrinkled by the random part.

BODY_AX, BODY_AY, BODY_Z = 0.30, 0.42, 0.20     # half extents (world units; the grid is [-0.5, 0.5]^3)


def body_state_dict(opt, seed=0, gain=0.35, A=8.0):
    sd = synthetic_state_dict(opt, seed=seed, mlp_gain=gain)
    zscale = float(opt.loadSize // 2) / float(opt.z_size)       # z_feat = 2 z * zscale
    B = A / (0.99 * 2.0 * BODY_Z * zscale)                       # logit(0, 0, +-BODY_Z) = 0 without the random part
    for m, c0 in (("mlp_lr.", 321), ("mlp_hr.", 322)):
        w = [sd[m + "conv%d.weight" % l] for l in range(5)]
        b = [sd[m + "conv%d.bias" % l] for l in range(5)]
        w[0][0:2] = 0.0
        w[0][0, 320, 0], w[0][1, 320, 0] = 1.0, -1.0             # y0[0] = lrelu(z_feat), y0[1] = lrelu(-z_feat)
        b[0][0:2] = 0.0
        for l in (1, 2, 3):                                       # unit 0 of layers 1-3 carries 0.99 |z_feat|
            w[l][0] = 0.0
            b[l][0] = 0.0
        w[1][0, 0, 0] = w[1][0, 1, 0] = 1.0
        w[2][0, 0, 0] = 1.0
        w[3][0, 0, 0] = 1.0
        w[4][0, 0, 0] = -B                                        # ... and enters the logit
        w[4][0, 128 + 0, 0] = A                                   # skip connection: feature channel 0 = G(x, y)
        b[4][0] = 0.0
    return sd


def body_features(hl, hh, seed=4, waves=5):
    """Feature maps for body_state_dict: ([256,hl,hl], [64,hh,hh]) float32, channel 0 of the first = G(x, y), every other
    channel a band-limited field (a few low-frequency cosines with seeded phases) of amplitude about 1."""
    def maps(name, c, h):
        p = prng.uniform(name, seed, (c, waves, 4), 0.0, 1.0).astype(np.float64)
        v, u = np.mgrid[:h, :h].astype(np.float64) / max(h - 1, 1)       # rows <-> image Y, columns <-> image X, in [0, 1]
        out = np.zeros((c, h, h))
        for k in range(waves):
            fx, fy = (0.5 + 3.5 * p[:, k, 0])[:, None, None], (0.5 + 3.5 * p[:, k, 1])[:, None, None]
            out += np.cos(2 * np.pi * (fx * u[None] + fy * v[None] + p[:, k, 2][:, None, None])) * (0.3 + 0.4 * p[:, k, 3])[:, None, None]
        return out
    fl = maps("body_feat_lr", 256, hl)
    fh = maps("body_feat_hr", 64, hh)
    v, u = np.mgrid[:hl, :hl].astype(np.float64) / max(hl - 1, 1)
    X, Y = 2.0 * u - 1.0, 2.0 * v - 1.0                                   # image coordinates; world x = X/2, y = -Y/2
    fl[0] = 1.0 - (X / (2.0 * BODY_AX)) ** 2 - (Y / (2.0 * BODY_AY)) ** 2
    return fl.astype(np.float32), fh.astype(np.float32)
