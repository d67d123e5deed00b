"""The identity behind the restated column kernels v10 / v11, checked in float64 numpy on the oracle's inputs - no GPU,
no product code: along a grid column the layer-0 pre-activation of every channel is affine in (z_feat, p_lr), so

    W1 LeakyReLU(x(k)) + b1  ==  b1 + RA + RB z_feat(k) + RC p_lr(k) + sum over LISTED channels of W1[:, c] res_c(k)

with g_c the branch at mid column, R* = W1 (g . [a0, w0z, w0p]), res_c = LeakyReLU(x_c) - g_c x_c, and "listed" = the channels whose
box over the tile crosses zero or contradicts g_c (the kernel's classification, SurfaceClassifier.py:53-81 is what both sides state)."""
import numpy as np

import common
import oracle


def _lrelu(x):
    return np.maximum(x, 0.01 * x)


def test_layer1_restatement_is_exact_and_sparse():
    sd = common.state_dict()
    fl, fh = common.synth_features()
    rng = np.random.RandomState(5)
    R, tile = 512, 128
    z = -0.5 + np.arange(R) / R
    zf = (2.0 * z) * 512.0 / 200.0                      # calib z scale 2, DepthNormalizer: * (loadSize // 2) / z_size
    listed_total, tiles = 0, 0
    for _ in range(3):
        X, Y = rng.uniform(-0.9, 0.9, 2)
        g = np.concatenate([oracle._bilinear_np(fl, np.array([X], np.float32), np.array([Y], np.float32))[:, 0],
                            oracle._bilinear_np(fh, np.array([X], np.float32), np.array([Y], np.float32))[:, 0]]).astype(np.float64)
        p_lr = 1.0 / (1.0 + np.exp(-8.0 * (z + rng.uniform(-0.3, 0.3))))        # any per-voxel scalar in [0, 1]
        for m, pre in ((0, "mlp_lr."), (1, "mlp_hr.")):
            W0 = sd[pre + "conv0.weight"][:, :, 0].astype(np.float64)
            b0 = sd[pre + "conv0.bias"].astype(np.float64)
            W1 = sd[pre + "conv1.weight"][:, :, 0].astype(np.float64)
            b1 = sd[pre + "conv1.bias"].astype(np.float64)
            a0 = W0[:, :320] @ g + b0
            wz = W0[:, 320]
            wp = W0[:, 321] if m else np.zeros_like(wz)
            p = p_lr if m else np.zeros(R)
            x = a0[:, None] + wz[:, None] * zf[None, :] + wp[:, None] * p[None, :]          # [1024, R]
            dense = W1 @ _lrelu(x) + b1[:, None]
            gam = np.where(a0 + wz * zf[R // 2] + wp * 0.5 > 0, 1.0, 0.01)
            RA, RB, RC = W1 @ (gam * a0), W1 @ (gam * wz), W1 @ (gam * wp)
            for t in range(R // tile):
                sl = slice(t * tile, (t + 1) * tile)
                zlo, zhi, plo, phi = zf[sl].min(), zf[sl].max(), p[sl].min(), p[sl].max()
                mn = a0 + np.minimum(wz * zlo, wz * zhi) + np.minimum(wp * plo, wp * phi)
                mx = a0 + np.maximum(wz * zlo, wz * zhi) + np.maximum(wp * plo, wp * phi)
                constant = np.where(gam == 1.0, mn > 0, mx <= 0)
                listed = np.nonzero(~constant)[0]
                res = _lrelu(x[listed][:, sl]) - gam[listed, None] * x[listed][:, sl]
                # the unlisted channels' residual really is zero over the tile
                rest = np.nonzero(constant)[0]
                assert np.abs(_lrelu(x[rest][:, sl]) - gam[rest, None] * x[rest][:, sl]).max() < 1e-12
                restated = (b1 + RA)[:, None] + RB[:, None] * zf[None, sl] + RC[:, None] * p[None, sl] + W1[:, listed] @ res
                assert np.abs(restated - dense[:, sl]).max() < 1e-9 * max(1.0, np.abs(dense).max())
                listed_total += len(listed)
                tiles += 1
    # a small fraction of the 1024 channels per tile: what makes the restatement pay
    assert listed_total / tiles < 256, listed_total / tiles
