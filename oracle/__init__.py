"""ORACLE - test infrastructure only.

ctypes wrapper over oracle/liboracle.so (plain-C restatements, see mc_oracle.c
and surs_oracle.c) plus numpy compositions of the encoder.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the
product package (surs_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)


class _McResult(C.Structure):
    _fields_ = [("nverts", C.c_int), ("nfaces", C.c_int), ("verts", C.POINTER(C.c_float)),
                ("faces", C.POINTER(C.c_int)), ("normals", C.POINTER(C.c_float)),
                ("values", C.POINTER(C.c_float))]


class _Mlp(C.Structure):
    _fields_ = [("w", C.c_void_p * 5), ("b", C.c_void_p * 5), ("dims", C.c_int * 6)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_mc_lewiner.restype = C.c_int
        _lib.orc_mc_lewiner.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(_McResult)]
        _lib.orc_mc_free.argtypes = [C.POINTER(_McResult)]
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------ marching cubes

def marching_cubes_lewiner(volume, level):
    """measure.marching_cubes_lewiner(volume, level) with the reference's defaults
    (/root/reference/lib/mesh_util.py:40,45): returns verts f32 [V,3], faces i32 [F,3],
    normals f32 [V,3], values f32 [V]; raises like scikit-image does."""
    vol = _f32(volume)
    if vol.ndim != 3:
        raise ValueError("Input volume should be a 3D numpy array.")
    if min(vol.shape) < 2:
        raise ValueError("Input array must be at least 2x2x2.")
    r = _McResult()
    rc = lib().orc_mc_lewiner(_p(vol), vol.shape[0], vol.shape[1], vol.shape[2], float(level), C.byref(r))
    if rc == 1:
        raise ValueError("Surface level must be within volume data range.")
    if rc == 2:
        raise RuntimeError("No surface found at the given iso value.")
    try:
        nv, nf = r.nverts, r.nfaces
        verts = np.ctypeslib.as_array(r.verts, shape=(nv, 3)).copy()
        faces = np.ctypeslib.as_array(r.faces, shape=(nf, 3)).copy()
        normals = np.ctypeslib.as_array(r.normals, shape=(nv, 3)).copy()
        values = np.ctypeslib.as_array(r.values, shape=(nv,)).copy()
    finally:
        lib().orc_mc_free(C.byref(r))
    return verts, faces, normals, values


# ------------------------------------------------------------------ encoder primitives (NCHW, batch 1)

def conv2d(x, w, b=None, stride=1):
    x, w = _f32(x), _f32(w)
    cin, h, wd = x.shape
    cout, cin2, k, _ = w.shape
    assert cin == cin2
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (wd + 2 * pad - k) // stride + 1
    y = np.empty((cout, ho, wo), np.float32)
    bb = _f32(b) if b is not None else None
    lib().orc_conv2d(_p(x), cin, h, wd, _p(w), _p(bb) if bb is not None else None, cout, k, stride, _p(y))
    return y


def group_norm(x, gamma, beta, groups=32, eps=1e-5):
    x = _f32(x)
    c, h, w = x.shape
    y = np.empty_like(x)
    g, b = _f32(gamma), _f32(beta)
    lib().orc_group_norm(_p(x), c, h * w, groups, _p(g), _p(b), C.c_float(eps), _p(y))
    return y


def avg_pool2(x):
    x = _f32(x)
    c, h, w = x.shape
    y = np.empty((c, h // 2, w // 2), np.float32)
    lib().orc_avg_pool2(_p(x), c, h, w, _p(y))
    return y


def bicubic_up2(x, align_corners):
    x = _f32(x)
    c, h, w = x.shape
    y = np.empty((c, 2 * h, 2 * w), np.float32)
    lib().orc_bicubic_up2(_p(x), c, h, w, int(bool(align_corners)), _p(y))
    return y


def pixel_shuffle2(x):
    """nn.PixelShuffle(2): out[c, 2h+i, 2w+j] = in[4c+2i+j, h, w]  (SuRSSR_v3.py:111-115)"""
    c4, h, w = x.shape
    c = c4 // 4
    return np.ascontiguousarray(x.reshape(c, 2, 2, h, w).transpose(0, 3, 1, 4, 2).reshape(c, 2 * h, 2 * w))


def lrelu(x, slope):
    return np.where(x > 0, x, np.float32(slope) * x).astype(np.float32)


def relu(x):
    return np.maximum(x, np.float32(0))


# ------------------------------------------------------------------ encoder compositions

def super_res(sd, img, residual=True, n_block=(2, 2, 2)):
    """SuRSSR_v3.forward (/root/reference/lib/model/SuRSSR_v3.py:143-181), img [3,H,W].
    Returns (img_SR, new2=feature_lr, new_fin=feature_hr)."""
    P = "super_resolution."

    def cv(name, x, stride=1):
        return conv2d(x, sd[P + name + ".weight"], sd[P + name + ".bias"], stride)

    def resblocks(i, x, nb):
        for b in range(nb):
            r = cv("body%d.%d.body.2" % (i, b), relu(cv("body%d.%d.body.0" % (i, b), x)))
            x = r + x
        return x

    h = lrelu(cv("head.0", bicubic_up2(img, False)), 0.2)
    feats = []
    d = h
    for i, nb in zip((1, 2, 3), n_block):
        d = lrelu(cv("down%d.0" % i, d, 2), 0.2)
        if residual:
            d = resblocks(i, d, nb)
        d = lrelu(cv("tail%d.0" % i, d), 0.2)
        d = lrelu(cv("tail%d.2" % i, d), 0.2)
        feats.append(d)
    d1_f, d2_f, d3_f = feats
    bo = lrelu(cv("bottleneck.0", d3_f), 0.2)
    up1 = lrelu(pixel_shuffle2(lrelu(cv("bott2.0", np.concatenate([d3_f, bo], 0)), 0.2)), 0.2)
    new2 = np.concatenate([d2_f, up1], 0)
    up2 = lrelu(pixel_shuffle2(lrelu(cv("ups2.0", new2), 0.2)), 0.2)
    new3 = np.concatenate([d1_f, up2], 0)
    up3 = lrelu(pixel_shuffle2(lrelu(cv("ups3.0", new3), 0.2)), 0.2)
    fin = np.concatenate([h, up3], 0)
    new_fin = lrelu(cv("ups4.0", fin), 0.2)
    img_sr = cv("last.2", lrelu(cv("last.0", new_fin), 0.2))
    return img_sr, new2, new_fin


def conv_block(sd, prefix, x):
    """ConvBlock.forward, in_planes == out_planes (HGFilters.py:57-74)."""
    o1 = conv2d(relu(group_norm(x, sd[prefix + "bn1.weight"], sd[prefix + "bn1.bias"])), sd[prefix + "conv1.weight"])
    o2 = conv2d(relu(group_norm(o1, sd[prefix + "bn2.weight"], sd[prefix + "bn2.bias"])), sd[prefix + "conv2.weight"])
    o3 = conv2d(relu(group_norm(o2, sd[prefix + "bn3.weight"], sd[prefix + "bn3.bias"])), sd[prefix + "conv3.weight"])
    return np.concatenate([o1, o2, o3], 0) + x


def hourglass(sd, prefix, depth, x):
    """HourGlass._forward (HGFilters.py:96-117)."""

    def fwd(level, inp):
        up1 = conv_block(sd, prefix + "b1_%d." % level, inp)
        low1 = conv_block(sd, prefix + "b2_%d." % level, avg_pool2(inp))
        if level > 1:
            low2 = fwd(level - 1, low1)
        else:
            low2 = conv_block(sd, prefix + "b2_plus_%d." % level, low1)
        low3 = conv_block(sd, prefix + "b3_%d." % level, low2)
        return up1 + bicubic_up2(low3, True)

    return fwd(depth, x)


def filter_lr(sd, feature_lr, n_stack=3, depth=2, taps=None):
    """HGFilter.forward, down_type 'low_res', use_sigmoid False (HGFilters.py:183-206);
    eval keeps only the last stack's output (SuRSNet.py:109-110)."""
    P = "image_filter_lr."

    def c1(name, x):
        return conv2d(x, sd[P + name + ".weight"], sd[P + name + ".bias"])

    x = conv_block(sd, P + "conv2.", feature_lr)
    if taps is not None:
        taps["conv2"] = x
    previous = x
    out = None
    for i in range(n_stack):
        hg = hourglass(sd, P + "m%d." % i, depth, previous)
        ll = conv_block(sd, P + "top_m_%d." % i, hg)
        ll = relu(group_norm(c1("conv_last%d" % i, ll), sd[P + "bn_end%d.weight" % i], sd[P + "bn_end%d.bias" % i]))
        out = c1("l%d" % i, ll)
        if taps is not None:
            taps["hg%d" % i] = hg
            taps["out%d" % i] = out
        if i < n_stack - 1:
            previous = previous + c1("bl%d" % i, ll) + c1("al%d" % i, out)
    return out


def filter_hr(sd, feature_hr):
    """HGFilter.forward, down_type 'high_res': a single 1x1 conv (HGFilters.py:179-181)."""
    return conv2d(feature_hr, sd["image_filter_hr.conv5.weight"], sd["image_filter_hr.conv5.bias"])


# ------------------------------------------------------------------ point query

def _mlp_struct(sd, prefix, keep):
    m = _Mlp()
    dims = [sd[prefix + "conv0.weight"].shape[1]]
    for l in range(5):
        w = _f32(sd[prefix + "conv%d.weight" % l].reshape(sd[prefix + "conv%d.weight" % l].shape[0], -1))
        b = _f32(sd[prefix + "conv%d.bias" % l])
        keep += [w, b]
        m.w[l] = w.ctypes.data
        m.b[l] = b.ctypes.data
        dims.append(w.shape[0])
    for i, d in enumerate(dims):
        m.dims[i] = d
    return m


def query(sd, points, calib, feat_lr, feat_hr, load_size=1024, z_size=200.0, want_logits=False):
    """query_mr + query_sr + get_preds (SuRSNet.py:131-187, BaseSuRSNet.py:80-85), one view.
    points [3,N] f32, calib [4,4], feat_lr [C,H,W], feat_hr [C,H,W].  Returns (pred_hr, pred_lr[, logit_hr, logit_lr])."""
    pts = _f32(points)
    n = pts.shape[1]
    cal = _f32(np.asarray(calib).reshape(-1)[:16])
    fl, fh = _f32(feat_lr), _f32(feat_hr)
    keep = []
    mlr, mhr = _mlp_struct(sd, "mlp_lr.", keep), _mlp_struct(sd, "mlp_hr.", keep)
    outs = [np.empty(n, np.float32) for _ in range(4)]
    lib().orc_query(_p(pts), n, _p(cal), C.c_float(load_size // 2), C.c_float(z_size), _p(fl), fl.shape[0], fl.shape[1],
                    fl.shape[2], _p(fh), fh.shape[0], fh.shape[1], fh.shape[2], C.byref(mlr), C.byref(mhr),
                    _p(outs[0]), _p(outs[1]), _p(outs[2]), _p(outs[3]))
    return tuple(outs) if want_logits else (outs[0], outs[1])


def _bilinear_np(feat, u, v):
    """grid_sample(feat[C,H,W], (u,v), bilinear, zeros padding, align_corners=True) for N points -> [C,N] (geometry.py:4-12)."""
    f32 = np.float32
    C_, H, W = feat.shape
    ix = ((u + f32(1.0)) / f32(2.0)) * f32(W - 1)
    iy = ((v + f32(1.0)) / f32(2.0)) * f32(H - 1)
    fx, fy = np.floor(ix), np.floor(iy)
    x0, y0 = fx.astype(np.int64), fy.astype(np.int64)
    x1, y1 = x0 + 1, y0 + 1
    fx1, fy1 = (fx + f32(1.0)), (fy + f32(1.0))
    w = {(0, 0): (fx1 - ix) * (fy1 - iy), (0, 1): (ix - fx) * (fy1 - iy), (1, 0): (fx1 - ix) * (iy - fy), (1, 1): (ix - fx) * (iy - fy)}
    out = np.zeros((C_, u.shape[0]), f32)
    for (dy, dx), wt in w.items():
        xs, ys = (x1 if dx else x0), (y1 if dy else y0)
        ok = (xs >= 0) & (xs < W) & (ys >= 0) & (ys < H)
        xs_c, ys_c = np.clip(xs, 0, W - 1), np.clip(ys, 0, H - 1)
        out += np.where(ok, wt, f32(0.0)).astype(f32)[None, :] * feat[:, ys_c, xs_c]
    return out


def query_views(sd, points, calibs, feat_lr, feat_hr, projection="orthogonal", load_size=1024, z_size=200.0, transforms=None,
                points_sr=None, calibs_sr=None):
    """query_mr + query_sr + get_preds for ONE subject seen from V views (num_views = V >= 1), numpy restatement of
    SuRSNet.py:131-187 with SurfaceClassifier.forward's view mean (SurfaceClassifier.py:53-81: after layer
    len(filters)//2 = 2 both the activations and the input features are averaged over the views), the sample layout of
    reshape_sample_tensor (train_util.py:40-51) and both projections (geometry.py:15-48).
    points [V,3,N], calibs [V,4,4], feat_lr [V,256,h,w], feat_hr [V,64,H,W].
    transforms [V,2,3]: the image-space affine map of geometry.py:27-30 / 43-46, xy' = S xy + s applied after the projection, in
    PIFu's meaning `transforms[:, :2, :2]` / `[:, :2, 2:3]` (the reference's own slices `transforms[:2, :2]` fail in baddbmm for
    every shape, so no output of the reference exists for it: this leg of the oracle is unpinned).
    points_sr [V,3,N] (and calibs_sr, default calibs): query_sr called on OTHER points than query_mr (SuRSNet.py:161-187 -
    features, depth and in_img from points_sr, the lr occupancies of query_mr's points index by index).
    Returns pred_hr [V,N], pred_lr [V,N], logit_hr [N], logit_lr [N] (fp32)."""
    f32 = np.float32

    def sample(points, calibs):
        pts, cal = _f32(points), _f32(calibs)
        feats, masks = [], []
        for v in range(pts.shape[0]):
            rot, trans = cal[v, :3, :3], cal[v, :3, 3:4]
            xyz = (trans + rot @ pts[v]).astype(f32)                 # baddbmm
            if projection == "perspective":
                xy = (xyz[:2] / xyz[2:3]).astype(f32)
            else:
                xy = xyz[:2]
            if transforms is not None:
                tr = _f32(transforms)[v]
                xy = (tr[:2, 2:3] + tr[:2, :2] @ xy).astype(f32)
            z = xyz[2:3]
            masks.append(((xy[0] >= -1.0) & (xy[0] <= 1.0) & (xy[1] >= -1.0) & (xy[1] <= 1.0)).astype(f32))
            zf = (z * f32(load_size // 2) / f32(z_size)).astype(f32)   # DepthNormalizer.py:18
            feats.append(np.concatenate([_bilinear_np(_f32(feat_lr[v]), xy[0], xy[1]), _bilinear_np(_f32(feat_hr[v]), xy[0], xy[1]), zf], 0))
        return feats, masks

    feats, masks = sample(points, calibs)
    V = len(feats)

    def mlp(prefix, x_views):
        W = [_f32(sd[prefix + "conv%d.weight" % l]).reshape(sd[prefix + "conv%d.weight" % l].shape[0], -1) for l in range(5)]
        B = [_f32(sd[prefix + "conv%d.bias" % l])[:, None] for l in range(5)]
        lrelu = lambda a: np.where(a > 0, a, f32(0.01) * a).astype(f32)
        ys = []
        for x in x_views:                                        # layers 0..2 per view (layer 2 takes cat([y, feature]))
            y = lrelu(W[0] @ x + B[0])
            y = lrelu(W[1] @ y + B[1])
            ys.append(lrelu(W[2] @ np.concatenate([y, x], 0) + B[2]))
        inv = f32(1.0) / f32(len(x_views))
        y, xm = ys[0].copy(), x_views[0].copy()
        for v in range(1, len(x_views)):
            y, xm = y + ys[v], xm + x_views[v]
        y, xm = (y * inv).astype(f32), (xm * inv).astype(f32)    # .mean(dim=1): sum in view order, times 1/V
        y = lrelu(W[3] @ np.concatenate([y, xm], 0) + B[3])
        return (W[4] @ np.concatenate([y, xm], 0) + B[4])[0].astype(f32)

    sig = lambda a: (f32(1.0) / (f32(1.0) + np.exp(-a))).astype(f32)
    logit_lr = mlp("mlp_lr.", feats)
    pred_lr = np.stack([m * sig(logit_lr) for m in masks])       # in_img[:, None].float() * mlp(...)   SuRSNet.py:156
    if points_sr is not None:
        feats, masks = sample(points_sr, calibs if calibs_sr is None else calibs_sr)
    logit_hr = mlp("mlp_hr.", [np.concatenate([feats[v], pred_lr[v:v + 1]], 0) for v in range(V)])
    pred_hr = np.stack([m * sig(logit_hr) for m in masks])
    return pred_hr, pred_lr, logit_hr, logit_lr


def grid_points(res, b_min, b_max, i0=0, i1=None):
    """Flat grid coordinates [3, i1-i0] as float32 (create_grid + eval_func cast: sdf.py:4-29, mesh_util.py:24)."""
    rx, ry, rz = (res, res, res) if np.isscalar(res) else res
    total = rx * ry * rz
    i1 = total if i1 is None else i1
    bmin = np.ascontiguousarray(b_min, np.float64)
    bmax = np.ascontiguousarray(b_max, np.float64)
    out = np.empty((3, i1 - i0), np.float32)
    lib().orc_grid_points(rx, ry, rz, _p(bmin), _p(bmax), C.c_longlong(i0), C.c_longlong(i1), _p(out))
    return out


def coords_matrix(res, b_min, b_max):
    """create_grid's index->world matrix (sdf.py:16-21)."""
    rx, ry, rz = (res, res, res) if np.isscalar(res) else res
    m = np.eye(4)
    length = np.asarray(b_max, np.float64) - np.asarray(b_min, np.float64)
    m[0, 0], m[1, 1], m[2, 2] = length[0] / rx, length[1] / ry, length[2] / rz
    m[0:3, 3] = np.asarray(b_min, np.float64)
    return m


def reconstruction_dense(sd, feat_lr, feat_hr, calib, res, b_min, b_max, load_size=1024, z_size=200.0):
    """reconstruction(..., use_octree=False): dense sweep + 2x Lewiner MC + index->world transform
    (/root/reference/lib/mesh_util.py:8-49, lib/sdf.py:32-52).  Returns dict."""
    pts = grid_points(res, b_min, b_max)
    phr, plr = query(sd, pts, calib, feat_lr, feat_hr, load_size, z_size)
    shape = (res, res, res)
    sdf_hr, sdf_lr = phr.astype(np.float64).reshape(shape), plr.astype(np.float64).reshape(shape)
    mat = coords_matrix(res, b_min, b_max)
    out = {"sdf_hr": sdf_hr, "sdf_lr": sdf_lr, "mat": mat}
    for tag, sdf in (("hr", sdf_hr), ("lr", sdf_lr)):
        v, f, nrm, val = marching_cubes_lewiner(sdf, 0.5)
        vw = (np.matmul(mat[:3, :3], v.T) + mat[:3, 3:4]).T
        out["verts_" + tag], out["faces_" + tag] = vw, f
    return out


def eval_grid_octree(res, b_min, b_max, eval_func, threshold=0.05, init_resolution=64, trace=None, index_func=None):
    """eval_grid_octree (/root/reference/lib/sdf.py:55-120), vectorised per level instead of the reference's Python
    triple loop.  The cell walk is order independent (a cell writes only its own block; the corners it reads are the
    min-corners of cells later in the loop order), so deciding all cells of a level from the arrays as evaluated and
    then applying the fills gives the same arrays - tests/test_oracle_octree.py checks that against the reference's
    loop.  eval_func(points[3,n] float64) -> (hr[n], lr[n]); or index_func(reso, ii, jj, kk) -> (hr[n], lr[n]) on the voxel
    indices of the same points (tests that drive the walk with the product's own evaluator, whose rounding may depend on the
    level).  Keeps the shared-`dirty` quirk.
    Returns float64 (sdf_hr, sdf_lr) of shape [res]*3."""
    R = res
    bmin, bmax = np.asarray(b_min, np.float64), np.asarray(b_max, np.float64)
    scale = (bmax - bmin) / R
    sdf_hr, sdf_lr = np.zeros((R, R, R)), np.zeros((R, R, R))
    dirty = np.ones((R, R, R), dtype=bool)
    grid_mask = np.zeros((R, R, R), dtype=bool)
    reso = R // init_resolution
    while reso > 0:
        grid_mask[0:R:reso, 0:R:reso, 0:R:reso] = True
        test = np.logical_and(grid_mask, dirty)
        ii, jj, kk = np.nonzero(test)
        if index_func is not None:
            hr, lr = index_func(reso, ii, jj, kk)
        else:
            pts = np.stack([scale[0] * ii + bmin[0], scale[1] * jj + bmin[1], scale[2] * kk + bmin[2]])
            hr, lr = eval_func(pts)
        sdf_hr[test], sdf_lr[test] = hr, lr
        dirty[test] = False
        if reso <= 1:
            break
        if trace is not None:
            trace.append(("evaluated", reso, sdf_hr.copy(), sdf_lr.copy(), dirty.copy()))
        c = np.arange(0, R - reso, reso)
        if len(c):
            X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
            active = dirty[X + reso // 2, Y + reso // 2, Z + reso // 2]
            for sdf in (sdf_hr, sdf_lr):
                corners = np.stack([sdf[X + a * reso, Y + b * reso, Z + d * reso] for a in (0, 1) for b in (0, 1) for d in (0, 1)])
                vmin, vmax = corners.min(0), corners.max(0)
                flat = active & ((vmax - vmin) < threshold)
                mid = (vmax + vmin) / 2
                for x, y, z, m in zip(X[flat], Y[flat], Z[flat], mid[flat]):
                    sdf[x:x + reso, y:y + reso, z:z + reso] = m
                    dirty[x:x + reso, y:y + reso, z:z + reso] = False
        if trace is not None:
            trace.append(("cells", reso, sdf_hr.copy(), sdf_lr.copy(), dirty.copy()))
        reso //= 2
    return sdf_hr, sdf_lr



def point_runs(points, tile=64, cap=4096):
    """The runs of a point array [3, N] as surs_query_points_columns' run finder defines them (csrc/surs_query.hip point_runs_kernel;
    what lib/sdf.py:32-45 hands lib/mesh_util.py:20-28 per call: consecutive grid points, z fastest): a run starts where (x, y)
    differs from the predecessor's, or `cap` points into a run.  Returns colstart, kcount, tiles [(run, z tile)], and
    (ascending violated, descending violated): whether z goes down / up anywhere inside a run (NaN violates both).  Plain loops."""
    p = np.asarray(points, np.float32)
    z = p[2]   # (x, y compared as floats, like the kernel: -0.0 == 0.0, NaN != NaN)
    n = p.shape[1]
    colstart, r = [], 0
    asc_v = desc_v = False
    for i in range(n):
        nh = i == 0 or not (p[0, i] == p[0, i - 1]) or not (p[1, i] == p[1, i - 1])
        if nh:
            r = i
        head = nh or (i - r) % cap == 0
        if head:
            colstart.append(i)
        else:
            asc_v |= bool(z[i] < z[i - 1]) or bool(z[i] != z[i])
            desc_v |= bool(z[i] > z[i - 1]) or bool(z[i] != z[i])
    colstart = np.asarray(colstart, np.int32)
    kcount = np.diff(np.append(colstart, n)).astype(np.int32)
    tiles = np.asarray([(c, t) for c, k in enumerate(kcount) for t in range((k + tile - 1) // tile)], np.int32).reshape(-1, 2)
    return colstart, kcount, tiles, (asc_v, desc_v)


def num_threads():
    return int(lib().orc_num_threads())
