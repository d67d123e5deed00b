"""Pins oracle.query / oracle encoder / oracle reconstruction against outputs of the reference
itself (tests/golden/*.npz, made by tools/gen_golden.py in the build container)."""
import json
import os

import numpy as np
import pytest

import common
import oracle
from surs_amd import weights


def test_state_dict_spec_matches_reference(golden_dir):
    keys = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    spec = weights.state_dict_spec(common.opt())
    assert [[k, list(s)] for k, s, _ in spec] == keys
    assert len(keys) == 553
    assert sum(int(np.prod(s)) for _, s, _ in spec) == 23674400


def test_query_random_points(golden_dir):
    g = np.load(os.path.join(golden_dir, "query.npz"))
    fl, fh = common.synth_features()
    pts = weights.synthetic_points(50000, seed=2)
    phr, plr, lhr, llr = oracle.query(common.state_dict(), pts, common.CALIB, fl, fh, 1024, 200.0, want_logits=True)
    assert np.abs(phr - g["a_pred_hr"]).max() < 1e-5
    assert np.abs(plr - g["a_pred_lr"]).max() < 1e-5
    assert np.abs(lhr - g["a_logit_hr"]).max() < 2e-5
    assert np.abs(llr - g["a_logit_lr"]).max() < 2e-5
    assert ((phr == 0) == (g["a_pred_hr"] == 0)).all()  # the in-image mask


def test_query_full_size_features(golden_dir):
    """The same 50 000 points over feature maps of BASELINE's sizes (256 x 256^2 / 64 x 1024^2): tests/golden/query_h512.npz."""
    g = np.load(os.path.join(golden_dir, "query_h512.npz"))
    fl, fh = common.synth_features(seed=7, hl=256, hh=1024)
    pts = weights.synthetic_points(50000, seed=2)
    phr, plr, lhr, llr = oracle.query(common.state_dict(), pts, common.CALIB, fl, fh, 1024, 200.0, want_logits=True)
    assert np.abs(phr - g["pred_hr"]).max() < 1e-5 and np.abs(plr - g["pred_lr"]).max() < 1e-5
    assert np.abs(lhr - g["logit_hr"]).max() < 2e-5 and np.abs(llr - g["logit_lr"]).max() < 2e-5
    assert ((phr == 0) == (g["pred_hr"] == 0)).all()


def test_query_general_calib_and_edges(golden_dir):
    g = np.load(os.path.join(golden_dir, "query.npz"))
    fl, fh = common.synth_features()
    pts = weights.synthetic_points(4099, seed=5)
    phr, plr, lhr, llr = oracle.query(common.state_dict(), pts, g["b_calib"], fl, fh, 1024, 200.0, want_logits=True)
    assert np.abs(phr - g["b_pred_hr"]).max() < 2e-5 and np.abs(plr - g["b_pred_lr"]).max() < 2e-5
    assert np.abs(lhr - g["b_logit_hr"]).max() < 1e-4
    phr, plr = oracle.query(common.state_dict(), g["c_points"], common.CALIB, fl, fh, 1024, 200.0)
    assert np.abs(phr - g["c_pred_hr"]).max() < 1e-5 and np.abs(plr - g["c_pred_lr"]).max() < 1e-5
    assert phr[5] == 0 and phr[6] == 0 and phr[0] > 0  # just outside -> 0, exactly on the border -> inside


@pytest.mark.parametrize("H", [64, 96])
def test_encoder(H, golden_dir):
    g = np.load(os.path.join(golden_dir, "encoder_h%d.npz" % H))
    sd = common.state_dict()
    img = weights.synthetic_image(H, seed=1)[0]
    img_sr, f_lr, f_hr = oracle.super_res(sd, img)
    im_hr = oracle.filter_hr(sd, f_hr)
    taps = {}
    im_lr = oracle.filter_lr(sd, f_lr, taps=taps)
    s2, s4 = (lambda a: a[..., ::2, ::2]), (lambda a: a[..., ::4, ::4])
    tol = 2e-5
    assert common.rel_err(s2(img_sr), g["img_sr_sub"]) < tol
    assert common.rel_err(f_lr if H == 64 else s2(f_lr), g["feature_lr"]) < tol
    assert common.rel_err(s4(f_hr), g["feature_hr_sub"]) < tol
    assert common.rel_err(f_hr.mean((1, 2)), g["feature_hr_mean"]) < tol
    assert common.rel_err(np.abs(f_hr).max((1, 2)), g["feature_hr_absmax"]) < tol
    assert common.rel_err(im_lr if H == 64 else s2(im_lr), g["im_feat_lr"]) < tol
    assert common.rel_err(s4(im_hr), g["im_feat_hr_sub"]) < tol
    for k, v in taps.items():
        assert common.rel_err(s2(v), g["tap_%s_sub" % k]) < tol, k


@pytest.mark.parametrize("R", [32, 48])
def test_reconstruction_dense(R, golden_dir):
    g = np.load(os.path.join(golden_dir, "recon_r%d.npz" % R))
    sd = common.state_dict()
    img = weights.synthetic_image(64, seed=1)[0]
    _, f_lr, f_hr = oracle.super_res(sd, img)
    im_hr, im_lr = oracle.filter_hr(sd, f_hr), oracle.filter_lr(sd, f_lr)
    out = oracle.reconstruction_dense(sd, im_lr, im_hr, common.CALIB, R, [-0.5] * 3, [0.5] * 3)
    assert np.abs(out["sdf_hr"] - g["sdf_hr"]).max() < 2e-5
    assert np.abs(out["sdf_lr"] - g["sdf_lr"]).max() < 2e-5
    # mesh stage on the reference's own field: indices bit-exact, world vertices exact
    mat = oracle.coords_matrix(R, [-0.5] * 3, [0.5] * 3)
    for tag in ("hr", "lr"):
        v, f, _, _ = oracle.marching_cubes_lewiner(g["sdf_" + tag].astype(np.float64), 0.5)
        vw = (np.matmul(mat[:3, :3], v.T) + mat[:3, 3:4]).T
        assert np.array_equal(f, g["faces_" + tag])
        assert np.array_equal(vw, g["verts_" + tag])


def test_multiview_oracle_matches_reference_goldens(golden_dir):
    """num_views = 2 (orthogonal) and 3 (perspective): the numpy restatement of the view-mean network against the
    outputs of the reference itself (tests/golden/query_views.npz, tools/gen_golden.py views)."""
    g = np.load(os.path.join(golden_dir, "query_views.npz"))
    sd = common.state_dict()
    for tag, V, proj, n in (("o2", 2, "orthogonal", 3001), ("p3", 3, "perspective", 2050)):
        fl = np.stack([common.synth_features(seed=10 + v)[0] for v in range(V)])
        fh = np.stack([common.synth_features(seed=10 + v)[1] for v in range(V)])
        pts = weights.synthetic_points(n, seed=20 + V)
        out = oracle.query_views(sd, np.repeat(pts[None], V, 0), g[tag + "_calibs"], fl, fh, proj)
        for name, o in zip(("pred_hr", "pred_lr", "logit_hr", "logit_lr"), out):
            assert o.shape == g[tag + "_" + name].shape
            assert np.abs(o - g[tag + "_" + name]).max() < 1e-5, (tag, name)
        # the masks differ between the views, the prediction under them does not
        assert (out[0] == 0).mean(1).std() > 0 or V == 1


def test_multiview_oracle_reduces_to_single_view():
    sd = common.state_dict()
    fl, fh = common.synth_features()
    pts = weights.synthetic_points(1500, seed=2)
    a = oracle.query(sd, pts, common.CALIB, fl, fh, want_logits=True)
    b = oracle.query_views(sd, pts[None], common.CALIB[None], fl[None], fh[None])
    for x, y in zip(a, (b[0][0], b[1][0], b[2], b[3])):
        assert np.abs(x - y).max() < 1e-5


def sr_other_case(g):
    """Inputs of tests/golden/query_sr_other.npz (tools/gen_golden.py query_sr): B = 2 subjects, query_mr and query_sr on
    different points and calibs."""
    n = int(g["n"])
    fs, ps = g["feat_seeds"], g["point_seeds"]
    feats = [common.synth_features(seed=int(k)) for k in fs]
    pts_mr = np.stack([weights.synthetic_points(n, seed=int(ps[0])), weights.synthetic_points(n, seed=int(ps[1]))])
    pts_sr = np.stack([weights.synthetic_points(n, seed=int(ps[2])), weights.synthetic_points(n, seed=int(ps[3]))])
    return feats, pts_mr, pts_sr


def test_query_sr_on_other_points_oracle_matches_reference_golden(golden_dir):
    """query_sr on a point set other than query_mr's, batch of two subjects: the oracle against the reference's own output."""
    g = np.load(os.path.join(golden_dir, "query_sr_other.npz"))
    feats, pts_mr, pts_sr = sr_other_case(g)
    sd = common.state_dict()
    for b in range(2):
        fl, fh = feats[b]
        hr, lr, _, _ = oracle.query_views(sd, pts_mr[b][None], g["cal_mr"][b][None], fl[None], fh[None], points_sr=pts_sr[b][None],
                                          calibs_sr=g["cal_sr"][b][None])
        assert np.abs(hr[0] - g["pred_hr"][b]).max() < 1e-5 and np.abs(lr[0] - g["pred_lr"][b]).max() < 1e-5
        # the hr mask is that of query_sr's points, not query_mr's
        same_pts, _, _, _ = oracle.query_views(sd, pts_mr[b][None], g["cal_mr"][b][None], fl[None], fh[None])
        assert not np.array_equal(same_pts[0] == 0, hr[0] == 0)


def test_point_runs_restatement_on_a_grid_chunk():
    """oracle.point_runs on a piece of a flattened grid (lib/sdf.py:4-29 order: z fastest) cut mid-column: the runs are the
    columns, the first and the last one partial; a run longer than the cap is cut; z order violations are reported per direction."""
    import oracle
    R = 20
    pts = oracle.grid_points(R, [-0.5] * 3, [0.5] * 3)
    a, n = 7 * R * R + 3 * R + 11, 1000
    cs, kc, tiles, viol = oracle.point_runs(pts[:, a:a + n], tile=8)
    assert cs[0] == 0 and kc[0] == R - 11 and (kc[1:-1] == R).all() and kc.sum() == n
    assert (np.diff(cs) == kc[:-1]).all() and viol == (False, True)          # z ascends inside every column
    assert len(tiles) == sum((k + 7) // 8 for k in kc) and tiles[0].tolist() == [0, 0] and tiles[-1][0] == len(kc) - 1
    line = np.stack([np.zeros(100, np.float32), np.zeros(100, np.float32), np.linspace(1, 0, 100, dtype=np.float32)])
    cs, kc, _, viol = oracle.point_runs(line, cap=32)
    assert cs.tolist() == [0, 32, 64, 96] and kc.tolist() == [32, 32, 32, 4] and viol == (True, False)
    line[2, 50] = np.nan
    assert oracle.point_runs(line, cap=32)[3] == (True, True)
    line[0, 10] = -0.0                                                        # -0.0 == 0.0: no new run
    assert oracle.point_runs(line, cap=1000)[0].tolist() == [0]
