"""GPU parity of the encoder primitives (through the C ABI) against the C oracle on seeded inputs.
fp32 MFMA fmaf chains vs the oracle's fp32 sums: relative tolerance 2e-5 of the output range."""
import numpy as np
import pytest
import torch

import common
from surs_amd import prng

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import gpu_common as g
    import oracle
    from surs_amd import native
    return dict(g=g, native=native, oracle=oracle, dev=g.dev())


def _img(env, x):
    return env["g"].upload_nhwc(x)


def _chw(img):
    return img.to_nchw()[0].cpu().numpy()


@pytest.mark.parametrize("cin,cout,h,w,k,stride,bias", [
    (3, 32, 37, 45, 3, 1, True),      # head: cin not a multiple of 4, ragged tile
    (32, 32, 64, 64, 3, 2, True),     # strided down conv
    (64, 128, 33, 31, 3, 1, True),
    (256, 128, 16, 16, 3, 1, False),  # ConvBlock conv1 (bias free)
    (64, 64, 40, 24, 1, 1, True),     # 1x1
    (512, 512, 8, 8, 3, 1, True),     # bott2 at the smallest size
    (32, 3, 20, 20, 3, 1, True),      # last: cout 3
    (32, 128, 256, 256, 3, 1, True),  # 512 workgroups of 8 rows x 64 channels: the big tile of the split-bf16 kernel
    (32, 64, 33, 47, 3, 2, True),     # stride 2 on the split-f16 kernel, odd sizes (ragged tiles, the last row / column padded)
    (16, 32, 18, 22, 3, 2, False),
    (3, 32, 5, 130, 3, 1, False),     # direct 3 -> 32 kernel: eight lanes per pixel, pixel count not a multiple of 32
    (3, 32, 32, 96, 3, 1, True),      # ... its whole-tile form (16 rows x 32 pixels per workgroup, the window sliding down in registers)
    (32, 3, 7, 9, 3, 1, False),       # direct 32 -> 3 kernel
    (6, 32, 12, 12, 3, 1, True),      # neither: the fp32 MFMA kernel
])
def test_conv(env, cin, cout, h, w, k, stride, bias):
    nat, orc = env["native"], env["oracle"]
    x = prng.uniform("cx", cin * 7 + h, (cin, h, w), -1, 1)
    wt = prng.uniform("cw", cout * 3 + cin, (cout, cin, k, k), -0.2, 0.2)
    b = prng.uniform("cb", cout, (cout,), -0.5, 0.5) if bias else None
    ref = orc.conv2d(x, wt, b, stride)
    cw = nat.ConvWeights(wt, b, env["dev"])
    y = _chw(nat.conv2d(_img(env, x), cw, stride=stride))
    assert y.shape == ref.shape
    assert common.rel_err(y, ref) < 2e-5


def test_head_conv_whole_tile_kernel_equals_the_per_pixel_kernel(env):
    """conv3x3_3to32_rows_kernel (maps of whole 16 x 32 tiles: the super-resolution net's head at 2H x 2W) against
    conv3x3_3to32_kernel (any size) on the same data: the interior of a 48 x 64 map computed by the tile kernel equals, bit for
    bit, the same pixels computed by the per-pixel kernel on the map padded to 49 x 64 by a zero row (which zero padding supplies
    anyway) - LeakyReLU, bias, residual and a channel slice of a wider output included."""
    nat = env["native"]
    h, w = 48, 64
    x = prng.uniform("hx", 11, (3, h, w), -1, 1)
    wt = prng.uniform("hw", 12, (32, 3, 3, 3), -0.2, 0.2)
    b = prng.uniform("hb", 13, (32,), -0.5, 0.5)
    res = prng.uniform("hr", 14, (32, h, w), -1, 1)
    cw = nat.ConvWeights(wt, b, env["dev"])
    wide = nat.Img(h, w, 64, device=env["dev"])
    wide.buf.zero_()
    nat.conv2d(_img(env, x), cw, out=wide.slice(32, 32), act=1, slope=0.2, residual=_img(env, res))
    tile = _chw(wide)
    assert np.all(tile[:32] == 0)
    xp = np.concatenate([x, np.zeros((3, 1, w), np.float32)], 1)
    rp = np.concatenate([res, np.zeros((32, 1, w), np.float32)], 1)
    pix = _chw(nat.conv2d(_img(env, xp), cw, act=1, slope=0.2, residual=_img(env, rp)))
    assert np.array_equal(tile[32:], pix[:, :h])


def test_conv_fused_groupnorm_relu_lrelu_residual_slice(env):
    """ConvBlock-style use: GN coefficients -> conv with fused GN-apply+ReLU prologue, LeakyReLU epilogue,
    residual add, output written into a channel slice of a wider tensor (the reference's torch.cat)."""
    nat, orc = env["native"], env["oracle"]
    cin, cout, h, w = 64, 32, 24, 20
    x = prng.uniform("gx", 1, (cin, h, w), -2, 3)
    gamma, beta = prng.uniform("gg", 1, (cin,), 0.5, 1.5), prng.uniform("gb", 1, (cin,), -0.3, 0.3)
    wt = prng.uniform("gw", 1, (cout, cin, 3, 3), -0.1, 0.1)
    res = prng.uniform("gr", 1, (cout, h, w), -1, 1)
    ref = orc.lrelu(orc.conv2d(orc.relu(orc.group_norm(x, gamma, beta)), wt), 0.2) + res
    X = _img(env, x)
    sc, sh = nat.groupnorm_coeffs(X, torch.from_numpy(gamma).to(env["dev"]), torch.from_numpy(beta).to(env["dev"]))
    wide = nat.Img(h, w, 96, device=env["dev"])
    wide.buf.zero_()
    nat.conv2d(X, nat.ConvWeights(wt, None, env["dev"]), out=wide.slice(64, 32), in_scale=sc, in_shift=sh, act=1, slope=0.2,
               residual=_img(env, res))
    full = _chw(wide)
    assert common.rel_err(full[64:], ref) < 2e-5
    assert np.all(full[:64] == 0)
    # GroupNorm apply alone
    y = _chw(nat.scale_shift_act(X, sc, sh, True))
    assert common.rel_err(y, orc.relu(orc.group_norm(x, gamma, beta))) < 2e-5


def test_pool_bicubic_shuffle_add(env):
    nat, orc = env["native"], env["oracle"]
    x = prng.uniform("px", 2, (64, 18, 26), -1, 1)
    X = _img(env, x)
    assert np.array_equal(_chw(nat.avgpool2(X)), orc.avg_pool2(x))
    for ac in (True, False):
        assert common.rel_err(_chw(nat.bicubic_up2(X, ac)), orc.bicubic_up2(x, ac)) < 1e-6
    add = prng.uniform("pa", 3, (64, 36, 52), -1, 1)
    assert common.rel_err(_chw(nat.bicubic_up2(X, True, addend=_img(env, add))), add + orc.bicubic_up2(x, True)) < 1e-6
    assert np.array_equal(_chw(nat.pixel_shuffle2(X, 0.2)), orc.lrelu(orc.pixel_shuffle2(x), 0.2))
    y = prng.uniform("py", 4, (64, 18, 26), -1, 1)
    z = prng.uniform("pz", 5, (64, 18, 26), -1, 1)
    assert np.array_equal(_chw(nat.add3(X, _img(env, y), _img(env, z))), (x + y) + z)
    assert np.array_equal(_chw(nat.add3(X, _img(env, y))), x + y)
    t = torch.from_numpy(x[None]).to(env["dev"])
    assert np.array_equal(_chw(nat.Img.from_nchw(t)), x)
    # channel counts that are not multiples of 4 / 16 take the scalar forms of the same kernels: same results
    x3 = prng.uniform("p3", 6, (3, 10, 14), -1, 1)
    X3 = _img(env, x3)
    assert np.array_equal(_chw(nat.avgpool2(X3)), orc.avg_pool2(x3))
    assert common.rel_err(_chw(nat.bicubic_up2(X3, False)), orc.bicubic_up2(x3, False)) < 1e-6
    y3 = prng.uniform("q3", 7, (3, 10, 14), -1, 1)
    assert np.array_equal(_chw(nat.add3(X3, _img(env, y3))), x3 + y3)
    x8 = prng.uniform("p8", 8, (8, 6, 10), -1, 1)
    assert np.array_equal(_chw(nat.pixel_shuffle2(_img(env, x8), 0.2)), orc.lrelu(orc.pixel_shuffle2(x8), 0.2))


@pytest.mark.parametrize("parts", [2, 3])
def test_split_operand_conv_equals_fp32_mfma_conv(env, parts, monkeypatch):
    """surs_conv2d_nhwc_x2 (two f16 parts per operand, three partial products: the encoder's default) and surs_conv2d_nhwc_x3
    (three bf16 parts, six products) against surs_conv2d_nhwc (fp32 MFMA) on the same input, fused GroupNorm prologue and
    LeakyReLU + residual epilogue included: within fp32 round-off of each other (two parts carry 22 significant bits)."""
    import ctypes as C
    nat = env["native"]
    monkeypatch.setenv("SURS_CONV_SPLIT", "bf16x3" if parts == 3 else "f16x2")
    cin, cout, h, w = 128, 96, 50, 70
    x = prng.uniform("sx", 1, (cin, h, w), -2, 2)
    wt = prng.uniform("sw", 2, (cout, cin, 3, 3), -0.1, 0.1)
    b = prng.uniform("sb", 3, (cout,), -0.5, 0.5)
    sc = torch.from_numpy(prng.uniform("ss", 4, (cin,), 0.5, 1.5)).to(env["dev"])
    sh = torch.from_numpy(prng.uniform("sh", 5, (cin,), -0.3, 0.3)).to(env["dev"])
    res = _img(env, prng.uniform("sr", 6, (cout, h, w), -1, 1))
    X = _img(env, x)
    cw = nat.ConvWeights(wt, b, env["dev"])
    assert cw.w3 is not None and cw.parts == parts
    y3 = _chw(nat.conv2d(X, cw, in_scale=sc, in_shift=sh, act=1, slope=0.2, residual=res))
    w3, cw.w3 = cw.w3, None
    y1 = _chw(nat.conv2d(X, cw, in_scale=sc, in_shift=sh, act=1, slope=0.2, residual=res))
    cw.w3 = w3
    assert common.rel_err(y3, y1) < (2e-6 if parts == 3 else 4e-6)


def _fold(st):
    """The partial sums of a GnStats, folded on the host: [32][2] float64."""
    return st.buf[:64 * st.slots].view(32, st.slots, 2).sum(dim=1).cpu().numpy()


@pytest.mark.parametrize("reduced", [False, True])
def test_conv_block_with_groupnorm_statistics_handed_between_kernels(env, reduced):
    """ConvBlock (lib/model/HGFilters.py:57-74) in four launches - every kernel leaves the GroupNorm statistics of the map it
    writes, the next 3x3 convolution folds them itself (surs_conv2d_nhwc_gn, surs_add3_gn) - against the oracle's ConvBlock and
    against the ten-launch form (surs_groupnorm_coeffs in front of each convolution); the statistics themselves against float64
    sums of the stored values; two runs give the same bits (no atomics)."""
    nat, orc, dev = env["native"], env["oracle"], env["dev"]
    c, h, w = 256, 44, 72                     # ragged tiles in both directions (44 = 5.5 x 8 rows, 72 = 2.25 x 32 columns)
    x = prng.uniform("bx", 1, (c, h, w), -2, 3)
    wts = [prng.uniform("bw", i, s, -0.08, 0.08) for i, s in enumerate(((c // 2, c, 3, 3), (c // 4, c // 2, 3, 3), (c // 4, c // 4, 3, 3)))]
    gns = [(prng.uniform("bg", i, (n,), 0.5, 1.5), prng.uniform("bb", i, (n,), -0.3, 0.3)) for i, n in enumerate((c, c // 2, c // 4))]
    o1 = orc.conv2d(orc.relu(orc.group_norm(x, *gns[0])), wts[0])
    o2 = orc.conv2d(orc.relu(orc.group_norm(o1, *gns[1])), wts[1])
    o3 = orc.conv2d(orc.relu(orc.group_norm(o2, *gns[2])), wts[2])
    ref = np.concatenate([o1, o2, o3]) + x
    cws = [nat.ConvWeights(wt, None, dev, reduced=reduced) for wt in wts]
    G = [(torch.from_numpy(g).to(dev), torch.from_numpy(b).to(dev)) for g, b in gns]

    def fused():
        X = nat.add3(_img(env, x), _img(env, np.zeros_like(x)), want_stats=True)     # x + 0, with the statistics of x
        out = nat.Img(h, w, c, device=dev)
        a, b_, d = out.slice(0, c // 2), out.slice(c // 2, c // 4), out.slice(3 * c // 4, c // 4)
        nat.conv2d_gn(X, cws[0], a, gn=G[0], want_stats=True)
        nat.conv2d_gn(a, cws[1], b_, gn=G[1], want_stats=True)
        nat.conv2d_gn(b_, cws[2], d, gn=G[2])
        pre = _chw(out)
        return X, a, b_, pre, nat.add3(out, X, out=out, want_stats=True)

    X, a, b_, pre, out = fused()
    y = _chw(out)
    tol = 2e-3 if reduced else 2e-5          # (one f16 product per MAC: 11 significant bits)
    assert common.rel_err(y, ref) < tol
    # the statistics are those of the stored values
    for img, vals in ((X, x), (a, pre[:c // 2]), (b_, pre[c // 2:3 * c // 4]), (out, y)):
        v = vals.astype(np.float64).reshape(32, -1)
        want = np.stack([v.sum(1), (v * v).sum(1)], 1)
        got = _fold(img.stats)
        assert np.allclose(got, want, rtol=1e-12, atol=1e-9), (img.c, np.abs(got - want).max())
    # the ten-launch form on the same kernels: the coefficients differ by float rounding of differently ordered double sums at most
    Xl = _img(env, x)
    outl = nat.Img(h, w, c, device=dev)
    al, bl, dl = outl.slice(0, c // 2), outl.slice(c // 2, c // 4), outl.slice(3 * c // 4, c // 4)
    for src, dst, cw, g in ((Xl, al, cws[0], G[0]), (al, bl, cws[1], G[1]), (bl, dl, cws[2], G[2])):
        sc, sh = nat.groupnorm_coeffs(src, *g)
        nat.conv2d(src, cw, out=dst, in_scale=sc, in_shift=sh)
    yl = _chw(nat.add3(outl, Xl, out=outl))
    assert common.rel_err(y, yl) < 1e-6
    # deterministic
    assert np.array_equal(_chw(fused()[4]), y)


def test_pool_and_bicubic_leave_groupnorm_statistics(env):
    nat, dev = env["native"], env["dev"]
    x = prng.uniform("sx", 9, (128, 36, 52), -1, 2)
    add = prng.uniform("sa", 9, (128, 72, 104), -1, 1)
    X, A = _img(env, x), _img(env, add)
    for got, plain in ((nat.avgpool2(X, want_stats=True), nat.avgpool2(X)),
                       (nat.bicubic_up2(X, True, addend=A, want_stats=True), nat.bicubic_up2(X, True, addend=A)),
                       # (the statistics form computes 2 x 2 blocks of outputs from one 5 x 5 window: the same bits per output)
                       (nat.bicubic_up2(X, False, want_stats=True), nat.bicubic_up2(X, False))):
        assert plain.stats is None and got.stats is not None and 0 < got.stats.slots <= 512
        v = _chw(got)
        assert np.array_equal(v, _chw(plain))
        v = v.astype(np.float64).reshape(32, -1)
        assert np.allclose(_fold(got.stats), np.stack([v.sum(1), (v * v).sum(1)], 1), rtol=1e-12, atol=1e-9)


def test_pointwise_conv_with_groupnorm_statistics(env):
    """The tail of an hourglass stack (lib/model/HGFilters.py:196-206): conv_last leaves bn_end's statistics, the pointwise
    convolution behind it folds them, adds `previous` in its epilogue and leaves the statistics of the sum."""
    nat, orc, dev = env["native"], env["oracle"], env["dev"]
    c, h, w = 256, 21, 37                       # 777 pixels: a ragged last 128-pixel block
    x = prng.uniform("qx", 1, (c, h, w), -2, 2)
    prev = prng.uniform("qp", 2, (c, h, w), -1, 1)
    w1, b1 = prng.uniform("qw", 3, (c, c, 1, 1), -0.1, 0.1), prng.uniform("qb", 3, (c,), -0.2, 0.2)
    w2, b2 = prng.uniform("qw", 4, (c, c, 1, 1), -0.1, 0.1), prng.uniform("qb", 4, (c,), -0.2, 0.2)
    gamma, beta = prng.uniform("qg", 5, (c,), 0.5, 1.5), prng.uniform("qh", 5, (c,), -0.3, 0.3)
    t_ref = orc.conv2d(x, w1, b1)
    ref = orc.conv2d(orc.relu(orc.group_norm(t_ref, gamma, beta)), w2, b2) + prev
    G = (torch.from_numpy(gamma).to(dev), torch.from_numpy(beta).to(dev))
    t = nat.conv2d_gn(_img(env, x), nat.ConvWeights(w1, b1, dev), want_stats=True)
    assert t.stats.slots == (h * w + 127) // 128
    y = nat.conv2d_gn(t, nat.ConvWeights(w2, b2, dev), gn=G, residual=_img(env, prev), want_stats=True)
    assert common.rel_err(_chw(t), t_ref) < 2e-5 and common.rel_err(_chw(y), ref) < 2e-5
    for img in (t, y):
        v = _chw(img).astype(np.float64).reshape(32, -1)
        assert np.allclose(_fold(img.stats), np.stack([v.sum(1), (v * v).sum(1)], 1), rtol=1e-12, atol=1e-9)
    # against the unfused launches on the same kernels
    tl = nat.conv2d(_img(env, x), nat.ConvWeights(w1, b1, dev))
    sc, sh = nat.groupnorm_coeffs(tl, *G)
    yl = nat.conv2d(tl, nat.ConvWeights(w2, b2, dev), in_scale=sc, in_shift=sh, residual=_img(env, prev))
    assert np.array_equal(_chw(tl), _chw(t)) and common.rel_err(_chw(y), _chw(yl)) < 1e-6
