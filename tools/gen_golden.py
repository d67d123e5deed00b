"""Generate golden vectors by running the upstream reference itself on CPU.

Build-container only: imports /root/reference through tools/ref_harness.py
(stub modules for the packages this image lacks) and writes small fixtures to
tests/golden/.  Inputs (weights, images, points, synthetic features) come from
the counter-based PRNG in surs_amd.prng, so the fixtures hold only the
reference's OUTPUTS plus the seeds/flags that produced them.

    python tools/gen_golden.py [query] [query_sr] [views] [encoder] [recon] [keys] [octree]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_harness as rh  # noqa: E402
from surs_amd import options, prng, weights  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
FLAGS = ["--loadSize", "1024", "--residual", "--b_min", "-0.5", "-0.5", "-0.5", "--b_max", "0.5", "0.5", "0.5",
         "--num_samples", "50000", "--z_size", "200"]
CALIB = np.diag([2.0, -2.0, 2.0, 1.0]).astype(np.float32)  # gen_mesh: lib/train_util.py:63-67


def make_net(seed=0):
    opt_ref = rh.parse_opt(FLAGS)
    net = rh.build_net(opt_ref)
    opt = options.BaseOptions().parse(FLAGS)
    sd = weights.synthetic_state_dict(opt, seed=seed)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    return net, opt_ref, sd


def run_query(net, pts, calib):
    """query_mr + query_sr + get_preds with logits captured at mlp_*.conv4."""
    cap = {}
    h1 = net.mlp_lr.conv4.register_forward_hook(lambda m, i, o: cap.__setitem__("lr", o.detach().clone()))
    h2 = net.mlp_hr.conv4.register_forward_hook(lambda m, i, o: cap.__setitem__("hr", o.detach().clone()))
    p = torch.from_numpy(pts[None].copy())
    c = torch.from_numpy(calib[None].copy())
    with torch.no_grad(), rh.quiet():
        net.query_mr(p, c)
        net.query_sr(p, c)
        phr, plr = net.get_preds()
    h1.remove(); h2.remove()
    return (phr[0, 0].numpy(), plr[0, 0].numpy(), cap["hr"][0, 0].numpy(), cap["lr"][0, 0].numpy())


def synth_features(seed=3, hl=32, hh=128):
    fl = prng.uniform("feat_lr", seed, (256, hl, hl), -1.0, 1.0)
    fh = prng.uniform("feat_hr", seed, (64, hh, hh), -1.0, 1.0)
    return fl, fh


def gen_keys():
    net, opt_ref, sd = make_net()
    ref_sd = net.state_dict()
    keys = [[k, list(v.shape)] for k, v in ref_sd.items()]
    with open(os.path.join(GOLD, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f)
    print("keys", len(keys))


def gen_query():
    net, opt_ref, sd = make_net()
    fl, fh = synth_features()
    net.im_feat_list_lr = [torch.from_numpy(fl[None].copy())]
    net.im_feat_list_hr = [torch.from_numpy(fh[None].copy())]
    out = {}
    # (a) 50k random points in [-0.55,0.55]^3 with gen_mesh's calib (about 17 % outside the image)
    pts = weights.synthetic_points(50000, seed=2)
    phr, plr, lhr, llr = run_query(net, pts, CALIB)
    out.update(a_pred_hr=phr, a_pred_lr=plr, a_logit_hr=lhr, a_logit_lr=llr)
    print("query a: pred_hr range", phr.min(), phr.max(), "pred_lr", plr.min(), plr.max(),
          "outside frac", float((phr == 0).mean()))
    # (b) general calib (rotation + translation, X/Y depend on z), 4099 points (ragged vs 64-point tiles)
    calib_b = np.array([[1.7, 0.3, -0.2, 0.05], [0.25, -1.8, 0.15, -0.04], [0.1, 0.2, 1.9, 0.02], [0, 0, 0, 1]],
                       np.float32)
    pts_b = weights.synthetic_points(4099, seed=5)
    phr, plr, lhr, llr = run_query(net, pts_b, calib_b)
    out.update(b_calib=calib_b, b_pred_hr=phr, b_pred_lr=plr, b_logit_hr=lhr, b_logit_lr=llr)
    # (c) edge cases: exactly on the image border (closed interval), just outside, centre, corners
    e = 0.5
    pts_c = np.array([[e, -e, 0, e, -e, 0.5000001, 0.0, 0.25, -0.5, 0.5],
                      [e, -e, 0, -e, e, 0.0, -0.5000001, 0.5, 0.5, -0.5],
                      [0.1, -0.2, 0, 0.3, -0.4, 0.0, 0.2, -0.5, 0.5, 0.0]], np.float32)
    phr, plr, lhr, llr = run_query(net, pts_c, CALIB)
    out.update(c_points=pts_c, c_pred_hr=phr, c_pred_lr=plr, c_logit_hr=lhr, c_logit_lr=llr)
    np.savez_compressed(os.path.join(GOLD, "query.npz"), **out)
    print("query done")


def gen_query_sr():
    """query_sr on OTHER points than the preceding query_mr, for a batch of two subjects (B = 2, num_views = 1): the reference
    itself (SuRSNet.py:131-187) - hr features / depth / in_img from query_sr's points and calibs, lr occupancies from query_mr's."""
    net, opt_ref, sd = make_net()
    fa, fb = synth_features(seed=3), synth_features(seed=4)
    net.im_feat_list_lr = [torch.from_numpy(np.stack([fa[0], fb[0]]))]
    net.im_feat_list_hr = [torch.from_numpy(np.stack([fa[1], fb[1]]))]
    calib_b = np.array([[1.7, 0.3, -0.2, 0.05], [0.25, -1.8, 0.15, -0.04], [0.1, 0.2, 1.9, 0.02], [0, 0, 0, 1]], np.float32)
    n = 4099
    pts_mr = np.stack([weights.synthetic_points(n, seed=11), weights.synthetic_points(n, seed=12)])
    pts_sr = np.stack([weights.synthetic_points(n, seed=13), weights.synthetic_points(n, seed=14)])
    cal_mr = np.stack([CALIB, calib_b])
    cal_sr = np.stack([calib_b, CALIB])
    with torch.no_grad(), rh.quiet():
        net.query_mr(torch.from_numpy(pts_mr.copy()), torch.from_numpy(cal_mr.copy()))
        net.query_sr(torch.from_numpy(pts_sr.copy()), torch.from_numpy(cal_sr.copy()))
        phr, plr = net.get_preds()
    np.savez_compressed(os.path.join(GOLD, "query_sr_other.npz"), feat_seeds=np.array([3, 4]), point_seeds=np.array([11, 12, 13, 14]),
                        n=np.array(n), cal_mr=cal_mr, cal_sr=cal_sr, pred_hr=phr[:, 0].numpy(), pred_lr=plr[:, 0].numpy())
    print("query_sr_other: pred_hr", phr.min().item(), phr.max().item(), "outside", float((phr == 0).float().mean()))


def gen_views():
    """num_views = 2 (orthogonal) and num_views = 3 (perspective): the reference's own multi-view path
    (SurfaceClassifier.py:70-76, train_util.py:40-51, geometry.py:34-48), one subject, per-view features and calibs."""
    ns = rh.load_reference()
    out = {}
    for tag, V, proj, n in (("o2", 2, "orthogonal", 3001), ("p3", 3, "perspective", 2050)):
        opt_ref = rh.parse_opt(FLAGS + ["--num_views", str(V)])
        with rh.quiet():
            net = ns.SuRSNet(opt_ref, proj).to(torch.device("cpu"))
        net.eval()
        sd = weights.synthetic_state_dict(options.BaseOptions().parse(FLAGS), seed=0)
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
        fl = np.stack([synth_features(seed=10 + v)[0] for v in range(V)])
        fh = np.stack([synth_features(seed=10 + v)[1] for v in range(V)])
        net.im_feat_list_lr = [torch.from_numpy(fl.copy())]
        net.im_feat_list_hr = [torch.from_numpy(fh.copy())]
        pts = weights.synthetic_points(n, seed=20 + V)
        if proj == "orthogonal":
            calibs = np.stack([np.array([[2.0 * np.cos(a), 0, 2.0 * np.sin(a), 0.02 * v], [0, -2.0, 0, -0.01 * v],
                                         [-2.0 * np.sin(a), 0, 2.0 * np.cos(a), 0], [0, 0, 0, 1]], np.float32)
                               for v, a in enumerate(np.linspace(0.0, 0.6, V))])
        else:   # projected z in about [1.5, 2.6]: x, y divided by it
            calibs = np.stack([np.array([[3.6 * np.cos(a), 0, 3.6 * np.sin(a), 0.03 * v], [0, -3.6, 0.2, 0.0],
                                         [0.15 * v, 0.1, 0.5, 2.0 + 0.05 * v], [0, 0, 0, 1]], np.float32)
                               for v, a in enumerate(np.linspace(0.0, 0.5, V))])
        cap = {}
        h1 = net.mlp_lr.conv4.register_forward_hook(lambda m, i, o: cap.__setitem__("lr", o.detach().clone()))
        h2 = net.mlp_hr.conv4.register_forward_hook(lambda m, i, o: cap.__setitem__("hr", o.detach().clone()))
        samples = ns.train_util.reshape_sample_tensor(torch.from_numpy(pts[None].copy()), V)   # [V,3,N]
        c = torch.from_numpy(calibs.copy())
        with torch.no_grad(), rh.quiet():
            net.query_mr(samples, c)
            net.query_sr(samples, c)
            phr, plr = net.get_preds()
        h1.remove(); h2.remove()
        assert tuple(phr.shape) == (V, 1, n) and tuple(cap["lr"].shape) == (1, 1, n)
        out.update({tag + "_calibs": calibs, tag + "_pred_hr": phr[:, 0].numpy(), tag + "_pred_lr": plr[:, 0].numpy(),
                    tag + "_logit_hr": cap["hr"][0, 0].numpy(), tag + "_logit_lr": cap["lr"][0, 0].numpy()})
        print("views", tag, "pred_hr range", float(phr.min()), float(phr.max()), "outside frac per view",
              [float((phr[v] == 0).float().mean()) for v in range(V)])
    np.savez_compressed(os.path.join(GOLD, "query_views.npz"), **out)


def _sub(a, step):
    return np.ascontiguousarray(a[..., ::step, ::step])


def gen_encoder():
    net, opt_ref, sd = make_net()
    for H in (64, 96):
        img = weights.synthetic_image(H, seed=1)
        taps = {}
        hooks = []

        def tap(name, mod):
            hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: taps.__setitem__(name, o.detach()[0].numpy().copy())))

        tap("conv2", net.image_filter_lr.conv2)
        for i in range(3):
            tap("hg%d" % i, getattr(net.image_filter_lr, "m%d" % i))
            tap("out%d" % i, getattr(net.image_filter_lr, "l%d" % i))
        t = time.time()
        with torch.no_grad(), rh.quiet():
            img_sr, f_lr, f_hr = net.super_res(torch.from_numpy(img.copy()))
            net.filter_hr(f_hr)
            net.filter_lr(f_lr)
        for h in hooks:
            h.remove()
        print("encoder H=%d: %.1fs" % (H, time.time() - t))
        im_lr = net.im_feat_list_lr[0][0].numpy()
        im_hr = net.im_feat_list_hr[0][0].numpy()
        out = dict(img_sr_sub=_sub(img_sr[0].numpy(), 2), feature_lr=f_lr[0].numpy() if H == 64 else _sub(f_lr[0].numpy(), 2),
                   feature_hr_sub=_sub(f_hr[0].numpy(), 4), im_feat_lr=im_lr if H == 64 else _sub(im_lr, 2),
                   im_feat_hr_sub=_sub(im_hr, 4),
                   feature_hr_mean=f_hr[0].numpy().mean((1, 2)), feature_hr_absmax=np.abs(f_hr[0].numpy()).max((1, 2)),
                   im_feat_hr_mean=im_hr.mean((1, 2)), im_feat_hr_absmax=np.abs(im_hr).max((1, 2)),
                   img_sr_mean=img_sr[0].numpy().mean((1, 2)))
        for k, v in taps.items():
            out["tap_" + k + "_sub"] = _sub(v, 2)
        np.savez_compressed(os.path.join(GOLD, "encoder_h%d.npz" % H), **out)
        for k, v in out.items():
            print("  ", k, v.shape, float(np.abs(v).max()))
        if H == 64:
            torch.save({"im_lr": im_lr, "im_hr": im_hr}, "/tmp/enc64_feats.pt")


def gen_encoder512():
    """BASELINE's image size: the reference encoder on the 512 x 512 synthetic image.  The tensors are 67 - 268 MB each, so
    the fixture holds strided sub-samples (every 8th / 16th pixel of every channel) plus per-channel means (float64 sums)
    and abs-maxima of the whole tensors."""
    net, opt_ref, sd = make_net()
    H = 512
    img = weights.synthetic_image(H, seed=1)
    taps, hooks = {}, []

    def tap(name, mod):
        hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: taps.__setitem__(name, o.detach()[0].numpy().copy())))

    tap("conv2", net.image_filter_lr.conv2)
    for i in range(3):
        tap("hg%d" % i, getattr(net.image_filter_lr, "m%d" % i))
        tap("out%d" % i, getattr(net.image_filter_lr, "l%d" % i))
    t = time.time()
    with torch.no_grad(), rh.quiet():
        img_sr, f_lr, f_hr = net.super_res(torch.from_numpy(img.copy()))
        net.filter_hr(f_hr)
        net.filter_lr(f_lr)
    for h in hooks:
        h.remove()
    print("encoder H=512: %.1fs" % (time.time() - t))
    full = dict(img_sr=img_sr[0].numpy(), feature_lr=f_lr[0].numpy(), feature_hr=f_hr[0].numpy(),
                im_feat_lr=net.im_feat_list_lr[0][0].numpy(), im_feat_hr=net.im_feat_list_hr[0][0].numpy())
    step = dict(img_sr=8, feature_lr=8, feature_hr=16, im_feat_lr=8, im_feat_hr=16)
    out = {}
    for k, v in full.items():
        out[k + "_sub"] = _sub(v, step[k])
        out[k + "_mean"] = v.astype(np.float64).mean((1, 2))
        out[k + "_absmax"] = np.abs(v).max((1, 2))
    for k, v in taps.items():
        out["tap_" + k + "_sub"] = _sub(v, 16)
        out["tap_" + k + "_mean"] = v.astype(np.float64).mean((1, 2))
    np.savez_compressed(os.path.join(GOLD, "encoder_h512.npz"), **out)
    for k, v in out.items():
        print("  ", k, v.shape, float(np.abs(v).max()))


def gen_query512():
    """The 50 000-point query of BASELINE configs[1] on feature maps of BASELINE's sizes (256 x 256^2 and 64 x 1024^2, PRNG
    values, seed 7 = tests/common.synth_features(seed=7, hl=256, hh=1024))."""
    net, opt_ref, sd = make_net()
    fl, fh = synth_features(seed=7, hl=256, hh=1024)
    net.im_feat_list_lr = [torch.from_numpy(fl[None].copy())]
    net.im_feat_list_hr = [torch.from_numpy(fh[None].copy())]
    pts = weights.synthetic_points(50000, seed=2)
    phr, plr, lhr, llr = run_query(net, pts, CALIB)
    np.savez_compressed(os.path.join(GOLD, "query_h512.npz"), pred_hr=phr, pred_lr=plr, logit_hr=lhr, logit_lr=llr)
    print("query512: pred_hr range", phr.min(), phr.max(), "outside frac", float((phr == 0).mean()))


def gen_recon():
    net, opt_ref, sd = make_net()
    ns = rh.load_reference()
    img = weights.synthetic_image(64, seed=1)
    with torch.no_grad(), rh.quiet():
        img_sr, f_lr, f_hr = net.super_res(torch.from_numpy(img.copy()))
        net.filter_hr(f_hr)
        net.filter_lr(f_lr)
    calib = torch.from_numpy(CALIB[None].copy())
    b_min, b_max = np.array([-0.5, -0.5, -0.5]), np.array([0.5, 0.5, 0.5])
    for R in (32, 48):
        t = time.time()
        cap = {}
        orig = ns.sdf.eval_grid

        def spy(coords, eval_func, num_samples):
            a, b = orig(coords, eval_func, num_samples=num_samples)
            cap["hr"], cap["lr"] = a, b
            return a, b

        ns.mesh_util.eval_grid = spy
        try:
            with torch.no_grad(), rh.quiet():
                vh, fh, _, _, vl, fl_, _, _ = ns.mesh_util.reconstruction(
                    opt_ref, net, torch.device("cpu"), calib, R, b_min, b_max, use_octree=False, num_samples=50000)
        finally:
            ns.mesh_util.eval_grid = orig
        print("recon R=%d %.1fs verts %s faces %s | lr %s %s; sdf_hr range %.3f..%.3f" %
              (R, time.time() - t, vh.shape, fh.shape, vl.shape, fl_.shape, cap["hr"].min(), cap["hr"].max()))
        # the reference's own OBJ writer (lib/mesh_util.py:53-61) on the HR mesh: digest of the file bytes
        import hashlib, tempfile
        with tempfile.TemporaryDirectory() as d:
            ns.mesh_util.save_obj_mesh(os.path.join(d, "m.obj"), vh, fh)
            obj_sha = hashlib.sha256(open(os.path.join(d, "m.obj"), "rb").read()).hexdigest()
        np.savez_compressed(os.path.join(GOLD, "recon_r%d.npz" % R), sdf_hr=cap["hr"].astype(np.float32),
                            sdf_lr=cap["lr"].astype(np.float32), verts_hr=vh, faces_hr=fh, verts_lr=vl, faces_lr=fl_,
                            obj_hr_sha256=np.array(obj_sha))


def gen_recon128():
    """BASELINE configs[0] at its stated size: the reference's dense reconstruction (lib/mesh_util.py:8-49, lib/sdf.py:32-52) at
    R = 128 on the 512 x 512 synthetic image (encoder included).  Fixture: the whole hr field (fp32, so that the product's
    marching cubes can be run on exactly the reference's volume and held to the digests of the reference's mesh), every
    second voxel of the lr field, mesh sizes, SHA-256 of the vertex / face arrays and of the reference writer's OBJ bytes."""
    import hashlib
    import tempfile
    net, opt_ref, sd = make_net()
    ns = rh.load_reference()
    img = weights.synthetic_image(512, seed=1)
    t = time.time()
    with torch.no_grad(), rh.quiet():
        img_sr, f_lr, f_hr = net.super_res(torch.from_numpy(img.copy()))
        net.filter_hr(f_hr)
        net.filter_lr(f_lr)
    print("encoder H=512: %.1fs" % (time.time() - t))
    calib = torch.from_numpy(CALIB[None].copy())
    b_min, b_max = np.array([-0.5, -0.5, -0.5]), np.array([0.5, 0.5, 0.5])
    R = 128
    cap = {}
    orig = ns.sdf.eval_grid

    def spy(coords, eval_func, num_samples):
        a, b = orig(coords, eval_func, num_samples=num_samples)
        cap["hr"], cap["lr"] = a, b
        return a, b

    ns.mesh_util.eval_grid = spy
    t = time.time()
    try:
        with torch.no_grad(), rh.quiet():
            vh, fh, _, _, vl, fl_, _, _ = ns.mesh_util.reconstruction(
                opt_ref, net, torch.device("cpu"), calib, R, b_min, b_max, use_octree=False, num_samples=50000)
    finally:
        ns.mesh_util.eval_grid = orig
    print("recon R=128 H=512 %.1fs: hr %s %s | lr %s %s; sdf_hr range %.4f..%.4f" %
          (time.time() - t, vh.shape, fh.shape, vl.shape, fl_.shape, cap["hr"].min(), cap["hr"].max()))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    with tempfile.TemporaryDirectory() as d:
        ns.mesh_util.save_obj_mesh(os.path.join(d, "m.obj"), vh, fh)
        obj_sha = hashlib.sha256(open(os.path.join(d, "m.obj"), "rb").read()).hexdigest()
    hr, lr = cap["hr"].astype(np.float32), cap["lr"].astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "recon_r128_h512.npz"), sdf_hr=hr, sdf_lr_sub=lr[::2, ::2, ::2],
                        sdf_lr_mean=np.array(lr.astype(np.float64).mean()), sdf_lr_sha256=np.array(sha(lr)),
                        n_verts=np.array([len(vh), len(vl)]), n_faces=np.array([len(fh), len(fl_)]),
                        verts_hr_sha256=np.array(sha(vh.astype(np.float64))), faces_hr_sha256=np.array(sha(fh.astype(np.int32))),
                        verts_lr_sha256=np.array(sha(vl.astype(np.float64))), faces_lr_sha256=np.array(sha(fl_.astype(np.int32))),
                        verts_hr_sub=vh[::16].astype(np.float64), verts_lr_sub=vl[::16].astype(np.float64),
                        obj_hr_sha256=np.array(obj_sha))


def gen_octree():
    """eval_grid_octree of the reference: (i) an analytic field (exact restatement check of the cell logic),
    (ii) the network at R=128 (the default init_resolution=64 -> levels 2, 1)."""
    import types
    ns = rh.load_reference()
    bmin, bmax = np.array([-0.5] * 3), np.array([0.5] * 3)

    def field(points):
        x, y, z = points
        a = 0.5 + 0.4 * np.sin(7 * x + 1) * np.cos(5 * y) * np.sin(3 * z + 0.5) + 0.05 * np.sin(40 * x * y)
        b = 0.5 + 0.3 * np.cos(6 * x) * np.sin(4 * y + 1) * np.cos(5 * z)
        return a, b

    out = {}
    for tag, R, thr, init in (("a", 48, 0.05, 12), ("b", 40, 0.12, 20)):
        coords, mat = ns.sdf.create_grid(R, R, R, bmin, bmax)
        hr, lr = ns.sdf.eval_grid_octree(types.SimpleNamespace(threshold=thr), coords,
                                         lambda p: tuple(v[None, None, :] for v in field(p)), init_resolution=init,
                                         num_samples=50000)
        out[tag + "_hr"], out[tag + "_lr"] = hr, lr
        out[tag + "_cfg"] = np.array([R, thr, init])
        print("octree analytic", tag, R, "zero frac lr", float((lr == 0).mean()))
    np.savez_compressed(os.path.join(GOLD, "octree_analytic.npz"), **out)

    net, opt_ref, sd = make_net()
    img = weights.synthetic_image(64, seed=1)
    with torch.no_grad(), rh.quiet():
        img_sr, f_lr, f_hr = net.super_res(torch.from_numpy(img.copy()))
        net.filter_hr(f_hr)
        net.filter_lr(f_lr)
    calib = torch.from_numpy(CALIB[None].copy())
    R = 128
    cap = {}
    orig = ns.sdf.eval_grid_octree

    def spy(opt, coords, eval_func, **kw):
        a, b = orig(opt, coords, eval_func, **kw)
        cap["hr"], cap["lr"] = a, b
        return a, b

    ns.mesh_util.eval_grid_octree = spy
    t = time.time()
    try:
        with torch.no_grad(), rh.quiet():
            vh, fh, _, _, vl, fl_, _, _ = ns.mesh_util.reconstruction(opt_ref, net, torch.device("cpu"), calib, R, bmin, bmax,
                                                                      use_octree=True, num_samples=50000)
    finally:
        ns.mesh_util.eval_grid_octree = orig
    hr, lr = cap["hr"].astype(np.float32), cap["lr"].astype(np.float32)
    print("octree net R=128 %.1fs: verts hr %d lr %d; lr zeros %.4f; hr zeros %.4f" %
          (time.time() - t, len(vh), len(vl), float((lr == 0).mean()), float((hr == 0).mean())))
    np.savez_compressed(os.path.join(GOLD, "octree_r128.npz"), hr_sub=hr[::2, ::2, ::2], lr_sub=lr[1::2, ::2, 1::2],
                        n_verts=np.array([len(vh), len(vl)]), n_faces=np.array([len(fh), len(fl_)]),
                        zero_frac=np.array([float((hr == 0).mean()), float((lr == 0).mean())]),
                        mean=np.array([float(hr.mean()), float(lr.mean())]))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    what = sys.argv[1:] or ["keys", "query", "encoder", "recon"]
    torch.set_num_threads(8)
    for w in what:
        globals()["gen_" + w]()
