"""The encoder sequenced inside the library (csrc/surs_encoder_net.cpp: surs_encoder_super_res / _filter_lr / _filter_hr / _forward,
SURVEY 8b's proposed export) against the per-launch sequencing of the host mirror (encoder.py), which tests/test_gpu_model.py holds to
the reference's goldens.  With SURS_ENC_SEPARATE_SUM=1 (a ConvBlock's closing sum as a launch of its own, as encoder.py issues it)
these are the same launches in the same order on the same tiles, so every output must be equal BIT FOR BIT - eval and training mode
(every stack's output), with and without the lent side streams, the one-product opt-in included.  The default form (the sum in the
three convolutions' epilogues: one launch and one pass over the map fewer per ConvBlock) sums the GroupNorm statistics of a block's
output in another order: equal to 1e-5 of each map's range here, held to the reference's goldens in tests/test_gpu_model.py, and
deterministic (two runs, equal bits)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import common

pytestmark = pytest.mark.gpu


def _net(training=False, encoder_precision="auto"):
    from surs_amd import model
    o = common.opt()
    o.encoder_precision = encoder_precision
    net = model.SuRSNet(o).to(device=torch.device("cuda:0"))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in common.state_dict().items()})
    net.train(training)
    return net


def _hwc(t):
    return torch.as_strided(t.buf, (t.h, t.w, t.c), (t.w * t.ld, t.ld, 1), t.buf.storage_offset() + t.off)


@pytest.mark.parametrize("H,training,streams,prec", [(64, False, "1", "auto"), (96, True, "1", "auto"), (64, True, "0", "auto"),
                                                     (64, False, "1", "f16"), (512, False, "1", "auto")])
def test_native_sequencing_equals_host_sequencing_bit_for_bit(monkeypatch, H, training, streams, prec):
    from surs_amd import encoder, weights
    from surs_amd.model import _as_img
    monkeypatch.setenv("SURS_ENC_STREAMS", streams)
    monkeypatch.setenv("SURS_ENC_SEPARATE_SUM", "1")
    net = _net(training, prec)
    W = net._encoder_weights()
    assert encoder.native_enabled(W)
    x = _as_img(torch.from_numpy(weights.synthetic_image(H, seed=1)).to("cuda:0"))
    ref_sr = encoder.super_res(W, x)
    ref_lr = encoder.filter_lr(W, ref_sr[1], keep_all=training)
    ref_hr = encoder.filter_hr(W, ref_sr[2])[0]
    got_sr = encoder.super_res_native(W, x)
    for a, b, name in zip(got_sr, ref_sr, ("img_SR", "feature_lr", "feature_hr")):
        assert torch.equal(_hwc(a), _hwc(b)), name
    got_lr = encoder.filter_lr_native(W, got_sr[1], keep_all=training)
    assert len(got_lr) == len(ref_lr) == (W.opt.num_stack_lr if training else 1)
    for s, (a, b) in enumerate(zip(got_lr, ref_lr)):
        assert torch.equal(_hwc(a), _hwc(b)), "stack output %d" % s
    # want_image=False leaves img_SR out and changes nothing else; a second call reuses the workspace
    again = encoder.super_res_native(W, x, want_image=False)
    assert again[0] is None and torch.equal(_hwc(again[1]), _hwc(ref_sr[1])) and torch.equal(_hwc(again[2]), _hwc(ref_sr[2]))
    # ... and the whole encoder as ONE call (surs_encoder_forward)
    from surs_amd import native
    nn = encoder._native_net(W)
    ws = nn.workspace(H, H, x.buf.device)
    f_lr, f_hr = native.Img(H // 2, H // 2, 256), native.Img(2 * H, 2 * H, 64)
    im_lr, im_hr = native.Img(H // 2, H // 2, nn.last_ch), native.Img(2 * H, 2 * H, 64)
    ss, _ = encoder._lent_streams(W.opt.hg_depth)
    native.check(native.lib().surs_encoder_forward(C.byref(nn.net), x.ptr(), H, H, x.ld, f_lr.ptr(), f_hr.ptr(), im_lr.ptr(), im_hr.ptr(),
                                                   native._ptr(ws), ws.numel(), C.byref(ss) if ss is not None else None, native._stream()))
    assert torch.equal(_hwc(im_lr), _hwc(ref_lr[-1])) and torch.equal(_hwc(im_hr), _hwc(ref_hr))
    assert torch.equal(_hwc(f_lr), _hwc(ref_sr[1])) and torch.equal(_hwc(f_hr), _hwc(ref_sr[2]))


@pytest.mark.parametrize("H,training", [(64, True), (128, True), (512, False)])
def test_sum_in_the_convolutions_epilogues_against_the_separate_sum(monkeypatch, H, training):
    from surs_amd import encoder, weights
    from surs_amd.model import _as_img
    x = _as_img(torch.from_numpy(weights.synthetic_image(H, seed=1)).to("cuda:0"))
    outs = {}
    for mode in ("1", "0", "0"):
        monkeypatch.setenv("SURS_ENC_SEPARATE_SUM", mode)
        W = _net(training)._encoder_weights()
        assert encoder._native_net(W).net.flags == int(mode)
        sr = encoder.super_res_native(W, x, want_image=False)
        outs.setdefault(mode, []).append([_hwc(o).clone() for o in encoder.filter_lr_native(W, sr[1], keep_all=training)])
    sep, fused, again = outs["1"][0], outs["0"][0], outs["0"][1]
    assert len(sep) == len(fused) == (3 if training else 1)
    for a, b, c in zip(sep, fused, again):
        assert torch.equal(b, c)                                                   # deterministic
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()), float((a - b).abs().max())


def test_conv_with_the_sum_in_its_epilogue_against_conv_then_add(monkeypatch):
    """surs_conv2d_nhwc_gn_sum on its own: value, value + residual, both sets of statistics - against surs_conv2d_nhwc_gn followed by
    surs_add3_gn (the same tiles: equal bits for the maps; the statistics are sums in another order: 1e-12 relative)."""
    from surs_amd import native, prng
    from surs_amd import _lib
    import gpu_common as g
    h, w, cin, cout, ctot = 40, 96, 64, 32, 128    # whole tiles of 4 rows x 32 columns; channels [64, 96) of a 128-channel sum
    x = g.upload_nhwc(prng.uniform("sx", 1, (cin, h, w), -1, 1))
    xin = g.upload_nhwc(prng.uniform("sr", 2, (ctot, h, w), -1, 1))
    wt = prng.uniform("sw", 3, (cout, cin, 3, 3), -0.2, 0.2)
    gam, bet = [torch.from_numpy(prng.uniform(n, 4, (cin,), 0.5, 1.5)).to("cuda:0") for n in ("sg", "sb")]
    cw = native.ConvWeights(wt, None, x.buf.device)
    x0 = native.add3(x, x, want_stats=True)     # any producer that leaves statistics: x0 = 2 x
    ref_raw = native.conv2d_gn(x0, cw, gn=(gam, bet), want_stats=True)
    ref_sum = native.add3(ref_raw, xin.slice(64, cout), want_stats=False)
    cap = ((w + 31) // 32) * ((h + 3) // 4)
    raw, out = native.Img(h, w, cout), native.Img(h, w, ctot)
    s_in = _lib.GnStats(x0.stats.buf.data_ptr(), x0.stats.slots, 0, 0, (C.c_int * 3)(x0.stats.slots, 0, 0))
    sb1 = torch.zeros(32 * cap * 2, dtype=torch.float64, device="cuda:0")
    sb2 = torch.zeros(32 * cap * 2, dtype=torch.float64, device="cuda:0")
    s_out = _lib.GnStats(sb1.data_ptr(), cap, 0, 0, (C.c_int * 3)(0, 0, 0))
    slots = C.c_int(0)
    native.check(native.lib().surs_conv2d_nhwc_gn_sum(2, x0.ptr(), h, w, cin, x0.ld, native._ptr(cw.w3), None, C.byref(s_in), None, None,
                                                      native._ptr(gam), native._ptr(bet), 1e-5, raw.ptr(), cout, raw.ld, C.byref(s_out),
                                                      xin.slice(64, cout).ptr(), xin.ld, out.slice(64, cout).ptr(), out.ld, native._ptr(sb2), cap,
                                                      64 // 4, 4, C.byref(slots), native._stream()))
    assert torch.equal(_hwc(raw), _hwc(ref_raw))
    assert torch.equal(_hwc(out.slice(64, cout)), _hwc(ref_sum))
    n = slots.value
    assert n == s_out.slots[0] == ref_raw.stats.slots and s_out.pitch == n
    got1 = sb1[:32 * n * 2].view(32, n, 2).sum(1).cpu().numpy()
    ref1 = ref_raw.stats.buf[:32 * n * 2].view(32, n, 2).sum(1).cpu().numpy()
    assert np.allclose(got1, ref1, rtol=1e-12, atol=1e-9)
    # the sum's statistics: groups 16 .. 23 of the 128-channel map (4 channels each), rows `cap` slots apart
    v = _hwc(out.slice(64, cout)).double().reshape(-1, cout // 4, 4)
    want = torch.stack([v.sum((0, 2)), (v * v).sum((0, 2))], 1).cpu().numpy()
    got2 = sb2.view(32, cap, 2)[16:24, :n].sum(1).cpu().numpy()
    assert np.allclose(got2, want, rtol=1e-12, atol=1e-9)
    assert float(sb2.view(32, cap, 2)[:16].abs().max()) == 0.0 and float(sb2.view(32, cap, 2)[24:].abs().max()) == 0.0
    # ragged tiles are refused (the second output is made by the whole-tile epilogue): callers take conv + add there
    from surs_amd._lib import SursError
    with pytest.raises(SursError, match="whole tiles"):
        native.check(native.lib().surs_conv2d_nhwc_gn_sum(2, x0.ptr(), h - 1, w, cin, x0.ld, native._ptr(cw.w3), None, C.byref(s_in), None, None,
                                                          native._ptr(gam), native._ptr(bet), 1e-5, raw.ptr(), cout, raw.ld, C.byref(s_out),
                                                          xin.slice(64, cout).ptr(), xin.ld, out.slice(64, cout).ptr(), out.ld, native._ptr(sb2),
                                                          cap, 64 // 4, 4, C.byref(slots), native._stream()))


def test_facade_takes_the_native_encoder_and_falls_back_where_it_does_not_apply(monkeypatch):
    """SuRSNet.super_res / filter_lr / filter_hr go through the library's sequencing by default (one call per network) and through
    encoder.py's inside wide_operands() and with SURS_ENC_NATIVE=0 - the same features either way."""
    from surs_amd import encoder, native, weights
    net = _net()
    img = torch.from_numpy(weights.synthetic_image(64, seed=1)).to("cuda:0")
    calls = []
    real = encoder.filter_lr_native
    monkeypatch.setattr(encoder, "filter_lr_native", lambda *a, **k: calls.append(1) or real(*a, **k))

    def run():
        _, f_lr, f_hr = net.super_res(img)
        net.filter_hr(f_hr)
        net.filter_lr(f_lr)
        return net.im_feat_list_lr[-1].clone(), net.im_feat_list_hr[0].clone()
    a = run()
    assert calls == [1]
    monkeypatch.setenv("SURS_ENC_NATIVE", "0")
    b = run()
    assert calls == [1] and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    monkeypatch.delenv("SURS_ENC_NATIVE")
    with native.wide_operands():
        c = run()
    assert calls == [1] and float((a[0] - c[0]).abs().max()) < 1e-4 * float(a[0].abs().max())


def test_native_encoder_refuses_bad_arguments():
    from surs_amd import encoder, native
    from surs_amd._lib import SursError
    net = _net()
    W = net._encoder_weights()
    nn = encoder._native_net(W)
    assert native.lib().surs_encoder_workspace_bytes(C.byref(nn.net), 0, 64) == 0
    x = native.Img(66, 64, 3)
    with pytest.raises(ValueError):
        encoder.super_res_native(W, x)
    ws = torch.empty(1024, dtype=torch.uint8, device="cuda:0")
    x = native.Img(64, 64, 3)
    out = native.Img(32, 32, 256), native.Img(128, 128, 64)
    with pytest.raises(SursError, match="workspace too small"):
        native.check(native.lib().surs_encoder_super_res(C.byref(nn.net), x.ptr(), 64, 64, 3, 0, None, out[0].ptr(), out[1].ptr(),
                                                         native._ptr(ws), ws.numel(), native._stream()))
