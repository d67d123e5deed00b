"""CPU-only checks: the C ABI library loads and exports every symbol include/surs.h declares (no compute calls),
the host-side mirror of the reference interface (flags, state dict, grid matrix, OBJ writer) behaves like the
reference, and nothing in the product package imports the oracle."""
import os
import re

import numpy as np
import pytest

import common
from surs_amd import mesh_util, model, options, sdf, weights

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared():
    txt = open(os.path.join(ROOT, "include", "surs.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(surs_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from surs_amd import _lib
    names = _declared()
    assert len(names) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == names           # the ctypes binding covers the whole header
    assert _lib.lib().surs_abi_version() == 1


def test_switches_live_in_one_table():
    """The host mirror reads the environment in ONE place (settings.get: override, then environment, then default) and the library
    in one place (its option table behind surs_set_option): no other os.environ / getenv on the path."""
    import os
    import re
    from surs_amd import settings
    pkg = os.path.dirname(os.path.abspath(settings.__file__))
    for f in sorted(os.listdir(pkg)):
        if f.endswith(".py") and f != "settings.py":
            assert "os.environ" not in open(os.path.join(pkg, f)).read(), f
    csrc = os.path.join(pkg, "csrc")
    sites = [f for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".cpp", ".inc", ".h"))
             for line in open(os.path.join(csrc, f)) if re.search(r"\bgetenv\(", line)]
    assert sites == ["surs_api.cpp"], sites
    assert settings.get("SURS_ENC_NATIVE") == os.environ.get("SURS_ENC_NATIVE", "1")
    settings.set("SURS_ENC_NATIVE", 0)
    try:
        assert settings.get("SURS_ENC_NATIVE") == "0" and settings.is_set("SURS_ENC_NATIVE")
    finally:
        settings.set("SURS_ENC_NATIVE", None)
    with pytest.raises(KeyError):
        settings.get("SURS_NO_SUCH_SWITCH")


def test_library_options_by_name():
    """surs_set_option / surs_get_option / surs_option_name: no GPU needed."""
    from surs_amd import native
    opts = native.options()
    assert {"grid_kernel", "split_parts", "conv_tall_min_wg", "point_runs_speculate"} <= set(opts) and len(opts) >= 13
    assert opts["conv_tall_min_wg"][0] == int(__import__("os").environ.get("SURS_CONV_TALL_MIN_WG", 256))
    old = native.get_option("gemm_waves")
    native.set_option("gemm_waves", 16)
    try:
        assert native.get_option("gemm_waves") == 16 and native.get_option("SURS_GEMM_WAVES") == 16   # either spelling
    finally:
        native.set_option("gemm_waves", old)
    from surs_amd._lib import SursError
    with pytest.raises(SursError, match="unknown option"):
        native.set_option("no_such_option", 1)


def test_conv_weight_packing_layout():
    import ctypes as C
    from surs_amd import _lib
    w = np.arange(5 * 3 * 9, dtype=np.float32).reshape(5, 3, 3, 3)
    n = _lib.lib().surs_conv_pack_weights(None, 5, 3, 3, None)
    assert n == 9 * 16 * 64
    out = np.empty(n, np.float32)
    _lib.lib().surs_conv_pack_weights(w.ctypes.data_as(C.c_void_p), 5, 3, 3, out.ctypes.data_as(C.c_void_p))
    out = out.reshape(9, 16, 64)
    assert out[4, 2, 3] == w[3, 2, 1, 1] and out[8, 0, 4] == w[4, 0, 2, 2]
    assert np.all(out[:, 3:, :] == 0) and np.all(out[:, :, 5:] == 0)


def test_options_mirror_reference_defaults():
    opt = options.BaseOptions().parse([])
    assert (opt.loadSize, opt.resolution, opt.num_samples, opt.z_size, opt.threshold) == (512, 512, 50000, 200.0, 0.05)
    assert opt.mlp_dim_lr == [321, 1024, 512, 256, 128, 1] and opt.mlp_dim_hr[0] == 322
    assert opt.mlp_res_layers_lr == [2, 3, 4] and opt.n_block == [2, 2, 2] and not opt.residual and not opt.no_residual
    assert (opt.hg_depth, opt.hg_dim, opt.num_stack_lr, opt.num_stack_hr, opt.norm, opt.num_views) == (2, 256, 3, 1, "group", 1)
    # README test command of the reference parses unchanged
    o = options.BaseOptions().parse("--dataroot d --results_path r --loadSize 1024 --resolution 256 --load_netG_checkpoint_path "
                                    "w --name n --residual --b_min -0.5 -0.5 -0.5 --b_max 0.5 0.5 0.5".split())
    assert o.b_max == [0.5, 0.5, 0.5] and o.residual and o.loadSize == 1024


def test_state_dict_roundtrip_is_strict():
    net = model.SuRSNet(common.opt())
    sd = net.state_dict()
    assert len(sd) == 553 and net.name == "base" and net.num_views == 1
    net.load_state_dict(sd)
    bad = dict(sd)
    bad.pop("mlp_hr.conv4.bias")
    with pytest.raises(RuntimeError, match="missing"):
        net.load_state_dict(bad)
    bad = dict(sd)
    bad["extra.weight"] = sd["mlp_hr.conv4.bias"]
    with pytest.raises(RuntimeError, match="unexpected"):
        net.load_state_dict(bad)
    bad = dict(sd)
    bad["mlp_lr.conv0.weight"] = sd["mlp_lr.conv0.weight"][:, :300]
    with pytest.raises(RuntimeError, match="size mismatch"):
        net.load_state_dict(bad)
    with pytest.raises(RuntimeError, match="no CPU path"):
        net.features() if False else net._device()
    # the shared GroupNorm of a ConvBlock's downsample branch aliases bn4 in the reference's state dict
    assert np.array_equal(sd["image_filter_lr.conv4.downsample.0.weight"].numpy(), sd["image_filter_lr.conv4.bn4.weight"].numpy())


def test_synthetic_inputs_are_deterministic():
    a = weights.synthetic_state_dict(common.opt(), seed=0)
    b = common.state_dict(0)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    img = weights.synthetic_image(64, seed=1)
    assert img.shape == (1, 3, 64, 64) and img[0, 0, 0, 0] == 0 and abs(img).max() <= 1
    assert np.array_equal(img, weights.synthetic_image(64, seed=1))
    assert not np.array_equal(img, weights.synthetic_image(64, seed=2))


def test_create_grid_matrix_matches_reference_formula():
    import oracle
    _, m = sdf.create_grid(48, 48, 48, np.array([-0.5, -0.4, -0.3]), np.array([0.5, 0.6, 0.7]))
    assert np.array_equal(m, oracle.coords_matrix(48, [-0.5, -0.4, -0.3], [0.5, 0.6, 0.7]))
    t = np.eye(4)
    t[0, 3] = 2.0
    _, m2 = sdf.create_grid(8, 8, 8, np.zeros(3), np.ones(3), transform=t)
    assert m2[0, 3] == 2.0 and m2[0, 0] == 1 / 8


def test_obj_writer_format(tmp_path):
    v = np.array([[0.12345, -1.0, 2.00004], [1e-5, 3.14159, -0.00005]])
    f = np.array([[0, 1, 1], [1, 0, 0]], np.int32)
    p = tmp_path / "m.obj"
    mesh_util.save_obj_mesh(str(p), v, f)
    assert p.read_text() == "v 0.1235 -1.0000 2.0000\nv 0.0000 3.1416 -0.0001\nf 1 2 2\nf 2 1 1\n".replace("0.1235", "%.4f" % 0.12345)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "super-resolution-3d-human-shape-from-a-single-low-resolution-image_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h", ".inc")):
                txt = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, fn


def test_native_obj_writer_equals_python_format(tmp_path):
    rng = np.random.default_rng(3)
    v = np.concatenate([rng.normal(size=(70000, 3)) * 3, np.array([[0.00005, -0.00005, 1e-9], [-0.0, 0.12345, 2.5],
                                                                    [1234567.891, 0.00015, 0.00025], [0.5, -0.5, 0.49995]])])
    f = rng.integers(0, len(v), size=(90000, 3)).astype(np.int32)
    p = tmp_path / "big.obj"
    mesh_util.save_obj_mesh(str(p), v, f)      # >= 65536 items: the multi-threaded path
    assert p.read_text() == mesh_util._obj_text(v, f)
    mesh_util.save_obj_mesh(str(p), v[:5], f[:0])
    assert p.read_text() == mesh_util._obj_text(v[:5], f[:0])


def test_eval_dataset_contract(tmp_path):
    from PIL import Image
    from surs_amd import data
    os.makedirs(tmp_path / "image_final")
    os.makedirs(tmp_path / "mask_final")
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, (24, 40, 3), dtype=np.uint8)
    m = (rng.integers(0, 2, (24, 40)) * 255).astype(np.uint8)
    for name in ("b_subject", "a_subject"):
        Image.fromarray(rgb).save(tmp_path / "image_final" / (name + ".png"))
        Image.fromarray(m).save(tmp_path / "mask_final" / (name + ".png"))
    opt = options.BaseOptions().parse(["--dataroot", str(tmp_path), "--b_min", "-0.5", "-0.5", "-0.5", "--b_max", "0.5", "0.5", "0.5"])
    ds = data.EvalDataset(opt)
    assert len(ds) == 2 and ds[0]["name"] == ("a_subject", ".png")          # sorted listing, (stem, ext) tuple
    it = ds[1]
    assert tuple(it["img_LR"].shape) == (1, 3, 24, 40) and it["img_LR"].dtype.is_floating_point
    want = (m[None] / 255.0) * ((rgb.transpose(2, 0, 1) / 255.0 - 0.5) / 0.5)
    assert np.abs(it["img_LR"][0].numpy() - want).max() < 1e-6
    assert np.array_equal(it["calib"][0].numpy(), np.diag([2.0, -2.0, 2.0, 1.0]).astype(np.float32))
    assert np.array_equal(it["b_max"], [0.5, 0.5, 0.5])


def _pack_blob(dtype_code):
    import ctypes as C
    from surs_amd import _lib
    sd = common.state_dict()
    keep = []

    def arrs(prefix):
        ws, bs = (C.c_void_p * 5)(), (C.c_void_p * 5)()
        for l in range(5):
            w = np.ascontiguousarray(sd[prefix + "conv%d.weight" % l].reshape(sd[prefix + "conv%d.weight" % l].shape[0], -1))
            b = np.ascontiguousarray(sd[prefix + "conv%d.bias" % l])
            keep.extend([w, b])
            ws[l], bs[l] = w.ctypes.data, b.ctypes.data
        return ws, bs

    wl, bl = arrs("mlp_lr.")
    wh, bh = arrs("mlp_hr.")
    n = _lib.lib().surs_mlp_pack(wl, bl, wh, bh, dtype_code, None)
    blob = np.zeros(n, np.uint8)
    assert _lib.lib().surs_mlp_pack(wl, bl, wh, bh, dtype_code, blob.ctypes.data_as(C.c_void_p)) == n
    return blob, sd


def _bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def test_mlp_blob_fragment_images_and_bias_split():
    """surs_mlp_pack is host code: unpack what it wrote.  Header offsets (csrc/surs_mlp_layout.h: 14 uint32 before
    `core`), the 32x32x16 A-fragment image (v1-v3), the 16x16x32 image with the permuted k order of layers 2/3 (v4), and
    the three-part bias fragments whose parts must sum to the fp32 bias exactly in bf16."""
    blob, sd = _pack_blob(1)   # SURS_BF16
    hdr = blob[:256].view(np.uint32)
    assert hdr[0] == 0x53525553 and hdr[1] == 1
    # header: magic, dtype, wt[2][4], bias[2][4], w4[2], reserved[2], wc, bc, zvec, core, total, core16, b1frag
    core, total, core16, b1frag = int(hdr[25]), int(hdr[26]), int(hdr[27]), int(hdr[28])
    assert total == blob.size and core < core16 < b1frag < total
    per_mlp = 42 * 32768
    w1 = sd["mlp_hr.conv1.weight"].reshape(512, 1024)
    w2 = sd["mlp_hr.conv2.weight"].reshape(256, -1)
    rnd = lambda a: _bf16_to_f32(((a.view(np.uint32) + 0x7fff + ((a.view(np.uint32) >> 16) & 1)) >> 16).astype(np.uint16))
    # --- 32x32x16 image, hr MLP: layer 1 fragment (k-step s, row tile T): lane (r, h) element j = W1[32T + r][16s + 8h + j]
    img = blob[core + per_mlp: core + 2 * per_mlp].view(np.uint16)
    s_, T = 5, 3
    frag = _bf16_to_f32(img[(s_ * 16 + T) * 512:(s_ * 16 + T + 1) * 512]).reshape(64, 8)
    lane = np.arange(64)
    want = w1[32 * T + (lane & 31)][:, None, ...] if False else np.stack([w1[32 * T + (l & 31), 16 * s_ + 8 * (l >> 5): 16 * s_ + 8 * (l >> 5) + 8] for l in lane])
    assert np.array_equal(frag, rnd(np.ascontiguousarray(want)))
    # layer 2 (after 64*16 fragments): element j = W2[32T + r][16s + 8(j>>2) + 4h + (j&3)]
    s_, T = 7, 2
    base = (64 * 16 + s_ * 8 + T) * 512
    frag = _bf16_to_f32(img[base:base + 512]).reshape(64, 8)
    want = np.stack([[w2[32 * T + (l & 31), 16 * s_ + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3)] for j in range(8)] for l in lane]).astype(np.float32)
    assert np.array_equal(frag, rnd(want))
    # --- 16x16x32 image: layer 1 fragment (s, T of 32): lane (n, q) element j = W1[16T + n][32s + 8q + j];
    #     layer 2 (after 32*32 fragments, 16 row tiles): W2[16T + n][32s + 16(j>>2) + 4q + (j&3)]
    img16 = blob[core16 + per_mlp: core16 + 2 * per_mlp].view(np.uint16)
    s_, T = 9, 17
    frag = _bf16_to_f32(img16[(s_ * 32 + T) * 512:(s_ * 32 + T + 1) * 512]).reshape(64, 8)
    want = np.stack([w1[16 * T + (l & 15), 32 * s_ + 8 * (l >> 4): 32 * s_ + 8 * (l >> 4) + 8] for l in lane])
    assert np.array_equal(frag, rnd(np.ascontiguousarray(want)))
    s_, T = 4, 11
    base = (32 * 32 + s_ * 16 + T) * 512
    frag = _bf16_to_f32(img16[base:base + 512]).reshape(64, 8)
    want = np.stack([[w2[16 * T + (l & 15), 32 * s_ + 16 * (j >> 2) + 4 * (l >> 4) + (j & 3)] for j in range(8)] for l in lane]).astype(np.float32)
    assert np.array_equal(frag, rnd(want))
    # --- bias fragments [2][16][64][8]: lanes 0..31, elements 0..2 = three parts that sum to the bias exactly
    bf = _bf16_to_f32(blob[b1frag:b1frag + 2 * 16 * 1024].view(np.uint16)).reshape(2, 16, 64, 8)
    for m, key in enumerate(("mlp_lr.conv1.bias", "mlp_hr.conv1.bias")):
        b1 = sd[key]
        parts = bf[m, :, :32, :3].reshape(512, 3)
        assert np.array_equal((parts[:, 0] + parts[:, 1]) + parts[:, 2], b1)
        assert np.all(bf[m, :, 32:, :] == 0) and np.all(bf[m, :, :, 3:] == 0)
    # --- split-bf16 images of the fp32 layer matrices (header words 29..36 = wt3[2][4], 37 = wc3):
    #     [3 parts][K/16][M/32][2][32][8], parts sum to the fp32 weight exactly; hr MLP layer 1 (K = 1024, M = 512)
    K, M = 1024, 512
    off = int(hdr[29 + 4 * 1 + 1])
    img3 = _bf16_to_f32(blob[off:off + K * M * 6].view(np.uint16)).reshape(3, K // 16, M // 32, 2, 32, 8)
    s_, T = 37, 9
    got = (img3[0, s_, T] + img3[1, s_, T]) + img3[2, s_, T]            # [h][r][j] = W1[32T + r][16s + 8h + j]
    want = w1[32 * T:32 * T + 32, 16 * s_:16 * s_ + 16].reshape(32, 2, 8).transpose(1, 0, 2)
    assert np.array_equal(got, want)
    assert np.array_equal(img3[0, s_, T], rnd(np.ascontiguousarray(want)))
