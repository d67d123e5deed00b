"""GPU octree sweep (surs_octree_* + surs_octree_level_columns / surs_query_grid_indexed through the C ABI).
 (1) the cell pass against the oracle's level-by-level trace on an analytic field: bit-exact float64 arrays and masks;
 (2) end to end on the network at R=128 (levels 2, 1) against the reference's own eval_grid_octree output;
 (3) the whole walk at BASELINE's 512^3 (levels 8, 4, 2, 1) against the oracle's restatement of lib/sdf.py:55-120 driven by the
     product's own evaluator: the same lattice points evaluated at every level, the same float64 volumes, bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import common
from surs_amd import weights

pytestmark = pytest.mark.gpu

# tolerances of the end-to-end comparison with the REFERENCE's walk (different evaluator: a flat / not-flat decision flips where a
# cell's corner range is within the evaluators' 1e-5 of --threshold, and the flipped block is then interpolated instead of evaluated):
# round 4 measured on MI355X (printed by the test): 3.8e-6 of the voxels off by more than 1e-4, zero fractions within 3.4e-6,
# 12 of 421 031 / 365 321 vertices (3e-5); the bounds are five times that (until round 4: 1e-3, 2e-3, 2 %)
OCTREE_BAD_FRACTION, OCTREE_ZERO_DRIFT, OCTREE_VERT_DRIFT = 2e-5, 2e-5, 2e-4


def test_cell_pass_bitwise_vs_oracle_trace():
    import oracle
    from surs_amd import native
    from test_oracle_octree import field
    dev = native.require_gpu()
    lib = native.lib()
    ws = native.Workspace(dev)
    for R, thr, init in ((48, 0.05, 12), (40, 0.12, 20), (37, 0.08, 9)):
        trace = []
        oracle.eval_grid_octree(R, [-0.5] * 3, [0.5] * 3, field, thr, init, trace=trace)
        ev = {t[1]: t for t in trace if t[0] == "evaluated"}
        ce = {t[1]: t for t in trace if t[0] == "cells"}
        assert ce, "no cell level exercised"
        for reso, (_, _, hr1, lr1, d1) in ce.items():
            _, _, hr0, lr0, d0 = ev[reso]
            a = torch.from_numpy(hr0.reshape(-1).copy()).to(dev)
            b = torch.from_numpy(lr0.reshape(-1).copy()).to(dev)
            d = torch.from_numpy(d0.reshape(-1).astype(np.uint8)).to(dev)
            w = ws.get(lib.surs_octree_workspace_bytes(R, reso))
            native.check(lib.surs_octree_cells(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(d.data_ptr()), R, reso,
                                               float(thr), C.c_void_p(w.data_ptr()), w.numel(), None))
            torch.cuda.synchronize()
            assert np.array_equal(a.cpu().numpy().reshape(R, R, R), hr1), (R, reso)
            assert np.array_equal(b.cpu().numpy().reshape(R, R, R), lr1), (R, reso)
            assert np.array_equal(d.cpu().numpy().reshape(R, R, R).astype(bool), d1), (R, reso)
        # selection: lattice points that are dirty at the start of a level
        reso = max(ce)
        d0 = ev[reso][4] | True   # before evaluation everything is dirty at the first level
        dd = torch.from_numpy(np.ones(R ** 3, np.uint8)).to(dev)
        nl = (R + reso - 1) // reso
        idx = torch.empty(nl ** 3, dtype=torch.int64, device=dev)
        cnt_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        cnt = C.c_int(0)
        native.check(lib.surs_octree_select(C.c_void_p(dd.data_ptr()), R, reso, C.c_void_p(idx.data_ptr()), nl ** 3,
                                            C.c_void_p(cnt_dev.data_ptr()), C.byref(cnt), None))
        want = np.zeros((R, R, R), bool)
        want[0:R:reso, 0:R:reso, 0:R:reso] = True
        got = np.sort(idx[:cnt.value].cpu().numpy())
        assert np.array_equal(got, np.nonzero(want.reshape(-1))[0])


def test_octree_reconstruction_vs_reference(golden_dir):
    from surs_amd import mesh_util, model
    g = np.load(os.path.join(golden_dir, "octree_r128.npz"))
    dev = torch.device("cuda:0")
    net = model.SuRSNet(common.opt()).to(device=dev)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in common.state_dict().items()})
    net.eval()
    _, f_lr, f_hr = net.super_res(torch.from_numpy(weights.synthetic_image(64, seed=1)).to(dev))
    net.filter_hr(f_hr)
    net.filter_lr(f_lr)
    calib = torch.from_numpy(common.CALIB[None]).to(dev)
    b_min, b_max = np.array([-0.5] * 3), np.array([0.5] * 3)
    opt = common.opt()
    vh, vl, mat = mesh_util.eval_volumes_octree(opt, net, calib, 128, b_min, b_max)
    hr, lr = vh.cpu().numpy().astype(np.float32), vl.cpu().numpy().astype(np.float32)
    # a flat/not-flat decision can flip where the corner range is within ~1e-6 of the threshold: allow 0.1 % of voxels
    for got, want in ((hr[::2, ::2, ::2], g["hr_sub"]), (lr[1::2, ::2, 1::2], g["lr_sub"])):
        bad = np.abs(got - want) > 1e-4
        print("octree R=128 vs the reference's own walk: voxels off by > 1e-4: %.2e (max |d| where evaluated by both %.2e)" %
              (bad.mean(), np.abs(got - want)[~bad].max()))
        assert bad.mean() < OCTREE_BAD_FRACTION, bad.mean()
    print("zero fractions", (hr == 0).mean(), (lr == 0).mean(), "reference", g["zero_frac"])
    assert abs((hr == 0).mean() - g["zero_frac"][0]) < OCTREE_ZERO_DRIFT and abs((lr == 0).mean() - g["zero_frac"][1]) < OCTREE_ZERO_DRIFT
    out = mesh_util.reconstruction(opt, net, dev, calib, 128, b_min, b_max, use_octree=True)
    print("vertices", len(out[0]), len(out[4]), "reference", g["n_verts"])
    assert abs(len(out[0]) - g["n_verts"][0]) <= OCTREE_VERT_DRIFT * g["n_verts"][0]
    assert abs(len(out[4]) - g["n_verts"][1]) <= OCTREE_VERT_DRIFT * g["n_verts"][1]


def _walk_inputs(field, dev):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import precision_report as pr
    if field == "body":
        sd, Fl, Fh = pr.body_inputs(dev)
        return sd, Fl, Fh, None
    return pr.noise_inputs(dev, H=64 if field == "noise64" else 512)


@pytest.mark.parametrize("field,R,columns", [("body", 512, True), ("body", 512, False), ("noise64", 128, True), ("noise", 256, True),
                                             ("noise64", 74, True)])
def test_octree_walk_equals_oracle_walk_bit_for_bit(field, R, columns):
    """The reference's octree walk (lib/sdf.py:55-120: evaluate the dirty lattice points of a level, fill the blocks whose corner
    values span less than --threshold, halve the stride) restated in numpy by the oracle and DRIVEN BY THE PRODUCT'S EVALUATOR
    (native.octree_level_values: the column kernel of the level / the per-point layer kernels) against the device walk:
    per level the same number of evaluated lattice points, at the end the same float64 arrays (evaluated values, block fills and
    the zeros the shared-dirty artefact leaves), bit for bit.  R = 512 is BASELINE's grid (levels 8, 4, 2, 1); R = 74: ragged
    lattices (74 = 2 * 37: the last cells and tiles are partial)."""
    import oracle
    from surs_amd import native
    dev = native.require_gpu()
    sd, Fl, Fh, keep = _walk_inputs(field, dev)
    blob, _ = native.pack_mlp({k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in sd.items() if k.startswith("mlp_")},
                              "fp32", dev)
    ws = native.Workspace(dev)
    mat = oracle.coords_matrix(R, [-0.5] * 3, [0.5] * 3)[:3].reshape(-1)
    cal = common.CALIB.reshape(-1)[:12]
    thr, init = 0.05, (64 if R >= 128 else 37)
    stats = []
    vh, vl = native.octree_volumes(R, mat, cal, 512, 200.0, Fl, Fh, blob, ws, thr, init, columns=columns, stats=stats)
    got_hr, got_lr = vh.cpu().numpy(), vl.cpu().numpy()
    del vh, vl
    levels = []

    indep = []

    def index_func(reso, ii, jj, kk):
        idx = torch.from_numpy((ii.astype(np.int64) * R + jj) * R + kk).to(dev)
        a, b = native.octree_level_values(R, reso, idx, mat, cal, 512, 200.0, Fl, Fh, blob, ws, columns=columns)
        levels.append((reso, len(ii)))
        if columns:
            # the INDEPENDENT check of the column kernel's strided / item-list form: the same lattice points on
            # the per-point layer kernels (held to the reference's goldens at 1e-4), in logit space, every level, at this size
            pa, pb = native.octree_level_values(R, reso, idx, mat, cal, 512, 200.0, Fl, Fh, blob, ws, columns=False)
            for x, y in ((a, pa), (b, pb)):
                x64, y64 = x.double(), y.double()
                ok = (x64 > 0.0067) & (x64 < 0.9933) & (y64 > 0.0067) & (y64 < 0.9933)      # (|logit| < 5: see precision_report.field_stats)
                dl = (torch.log(x64 / (1 - x64)) - torch.log(y64 / (1 - y64))).abs()[ok]
                indep.append((reso, float(dl.max()) if dl.numel() else 0.0, float((x64 - y64).abs().max())))
        return a.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64)

    o_hr, o_lr = oracle.eval_grid_octree(R, [-0.5] * 3, [0.5] * 3, None, thr, init, index_func=index_func)
    print(field, R, "columns" if columns else "points", "evaluated per level:", levels, "of", R ** 3,
          "| zero voxels hr %.4f lr %.4f" % ((o_hr == 0).mean(), (o_lr == 0).mean()))
    assert [(r, n) for r, n, _, _ in stats] == levels
    print("   tiles per level:", [(r, t, "fill %.2f" % (n / (64.0 * t))) for r, n, _, t in stats if t])
    assert len(levels) >= 2 and sum(n for _, n in levels) < R ** 3
    assert np.array_equal(got_hr, o_hr)
    assert np.array_equal(got_lr, o_lr)
    if columns:
        print("   column kernel vs per-point kernels per level (reso, max |d logit|, max |d occupancy|):", indep)
        assert indep and all(dl < 1e-4 and dp < 3e-5 for _, dl, dp in indep), indep


@pytest.mark.parametrize("prec,tol", [("bf16", 3e-2), ("fp16", 4e-3)])
def test_octree_levels_in_reduced_precision(prec, tol):
    """`--precision bf16 | fp16 --octree_precision sweep` with use_octree=True: the levels run on the 16-bit column kernel
    (surs_octree_level_columns_dt, kernel v10 on lattice item lists; opt-in - the default keeps the levels fp32-grade).  (1) the level values of every level of the fp32 walk at 512^3, evaluated in the reduced precision,
    against the fp32-grade column kernel's: the bounds of the dense sweep's own test (test_grid_column_kernel_vs_fp32); (2) the whole
    reconstruction against the fp32-grade octree reconstruction: vertex counts within 1 %, 90 % (bf16) / 97 % (fp16) of the
    vertices of either mesh within half a voxel of the other - most of an octree mesh is the walk's artefact surfaces, which move where
    a flat / not-flat decision flips -, those 0.15 voxel apart on average."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import oracle
    import precision_report as pr
    from surs_amd import mesh_util, model, native, options
    dev = native.require_gpu()
    R = 512
    sd, Fl, Fh = pr.body_inputs(dev)
    mlp = {k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in sd.items() if k.startswith("mlp_")}
    b32, _ = native.pack_mlp(mlp, "fp32", dev)
    b16, _ = native.pack_mlp(mlp, prec, dev)
    ws = native.Workspace(dev)
    mat = oracle.coords_matrix(R, [-0.5] * 3, [0.5] * 3)[:3].reshape(-1)
    cal = common.CALIB.reshape(-1)[:12]
    worst = []

    def index_func(reso, ii, jj, kk):
        idx = torch.from_numpy((ii.astype(np.int64) * R + jj) * R + kk).to(dev)
        a, b = native.octree_level_values(R, reso, idx, mat, cal, 512, 200.0, Fl, Fh, b32, ws)
        c, d = native.octree_level_values(R, reso, idx, mat, cal, 512, 200.0, Fl, Fh, b16, ws, dtype=prec)
        worst.append((reso, len(ii), float((a - c).abs().max()), float((b - d).abs().max())))
        return a.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64)
    oracle.eval_grid_octree(R, [-0.5] * 3, [0.5] * 3, None, 0.05, 64, index_func=index_func)
    print("octree level values %s vs fp32-grade (reso, points, max |d| hr, lr):" % prec, worst)
    assert len(worst) == 4 and all(1e-7 < dh < tol and dl < tol for _, _, dh, dl in worst), worst
    # end to end
    calib = torch.from_numpy(pr.CALIB).to(dev)[None]
    outs = {}
    for p in ("fp32", prec):
        opt = options.BaseOptions().parse(pr.FLAGS + ["--precision", p, "--octree_precision", "sweep"])
        net = model.SuRSNet(opt).to(device=dev)
        net.load_state_dict(sd)
        net.eval()
        outs[p] = mesh_util.reconstruction(opt, net, dev, calib, R, np.array([-0.5] * 3), np.array([0.5] * 3), use_octree=True,
                                           features=(Fl, Fh), want_normals=False)
    for k in (0, 4):
        va, vb = outs["fp32"][k], outs[prec][k]
        assert abs(len(va) - len(vb)) <= 0.01 * len(va), (k, len(va), len(vb))
        a = torch.from_numpy(((np.asarray(va) + 0.5) * R).astype(np.float32)).to(dev)
        b = torch.from_numpy(((np.asarray(vb) + 0.5) * R).astype(np.float32)).to(dev)
        for x, y in ((a, b), (b, a)):
            d = pr.nearest_vertex_distance(x, y, R)
            near = d < 0.5
            print("   octree %s vs fp32-grade, field %d: %.5f of the vertices have a counterpart within half a voxel, those at %.3f voxel "
                  "on average (%d / %d vertices)" % (prec, k // 4, float(near.float().mean()), float(d[near].mean()), len(va), len(vb)))
            # (an octree volume holds blocks that were interpolated instead of evaluated and the zeros of the shared-dirty artefact;
            #  where a flat / not-flat decision flips between the two precisions a block's spurious surface appears or goes: a few
            #  vertices in 10^4 have no counterpart at all - the reference's own walk differs from ours the same way, see above)
            assert float(near.float().mean()) > (0.90 if prec == "bf16" else 0.97) and float(d[near].mean()) < 0.15
