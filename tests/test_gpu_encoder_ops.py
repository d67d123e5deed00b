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
