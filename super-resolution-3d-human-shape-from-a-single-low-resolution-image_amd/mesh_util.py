"""reconstruction() and the OBJ writers: same signatures and return values as the reference
(/root/reference/lib/mesh_util.py:8-89), implemented on the device:

    create_grid -> eval_grid (dense sweep, surs_query_grid) -> 2x Lewiner marching cubes (surs_mc_lewiner)
    -> index->world transform (surs_transform_points).

The occupancy volumes never leave the GPU; only the meshes are copied to the host.  `use_octree=True` (the
reference's default in gen_mesh) runs the reference's coarse-to-fine sweep (lib/sdf.py:55-120) level by level on the
device, shared-`dirty` artefact included, with the fp32 kernels; `use_octree=False` runs the dense
sweep in `opt.precision` - on this hardware the dense bf16 sweep is both faster and free of the artefact.
"""
import os

import numpy as np
import torch

from . import native, settings
from .sdf import create_grid


def eval_volumes(opt, net, calib_tensor, resolution, b_min, b_max, transform=None, i0=0, i1=None, precision=None, features=None):
    """Dense occupancy volumes of grid slab [i0, i1) as float32 device tensors (vol_hr, vol_lr), and the grid matrix."""
    _, mat = create_grid(resolution, resolution, resolution, b_min, b_max, transform=transform)
    i1 = resolution if i1 is None else i1
    calib = calib_tensor[0].detach().to("cpu", torch.float32).numpy().reshape(-1)[:12]
    fl, fh = features if features is not None else net.features()
    zmul, zdiv = net._zscale()
    prec = precision or getattr(opt, "precision", "fp32")
    blob = net._mlp_blob()
    if prec not in ("fp32", "fp32x") and native.DTYPES[prec] != net._core_dtype:
        raise ValueError("network was packed for %s, reconstruction asked for %s: set opt.precision before loading" %
                         (net.precision, prec))
    # (the dense column kernels where the sweep lists most channels: the same choice for every slab of this grid)
    kern = native.grid_kernel_for(resolution, resolution, resolution, mat[:3].reshape(-1), calib, zmul, zdiv, fl, fh, blob, prec,
                                  net._workspace())
    try:
        vh, vl = native.query_grid(i0, i1, resolution, resolution, mat[:3].reshape(-1), calib, zmul, zdiv, fl, fh, blob, prec,
                                   net._workspace(), kernel=kern)
    except native._lib.SursError as e:
        if e.code != -3:
            raise
        # general calibration / grid transform: the column kernel does not apply, evaluate in fp32
        vh, vl = native.query_grid(i0, i1, resolution, resolution, mat[:3].reshape(-1), calib, zmul, zdiv, fl, fh, blob, "fp32",
                                   net._workspace())
    return vh, vl, mat


def eval_volumes_views(opt, net, calib_tensor, resolution, b_min, b_max, transform=None, num_samples=1 << 18, loop=False):
    """Dense sweep for num_views > 1 or the perspective projection: eval_grid's batch loop (lib/sdf.py:32-52) over
    eval_func (lib/mesh_util.py:20-28) - every batch of grid points repeated per view, query_mr + query_sr, view 0's
    prediction kept (`net.get_preds()[0][0]`).  Grid coordinates in float64, cast to float32, like create_grid.
    One library call (surs_query_grid_views: the voxels generated in the gather, the feature maps laid out once); loop=True - and the
    retry on three bf16 parts after an f16 overflow - walks the batches through the facade as the reference does (same kernels, same bits:
    tests/test_gpu_model.py)."""
    _, mat = create_grid(resolution, resolution, resolution, b_min, b_max, transform=transform)
    dev = net._device()
    R = int(resolution)
    total = R * R * R
    if not loop and not native.wide_operands_active() and net.im_feat_list_lr and net.im_feat_list_hr \
            and net.im_feat_list_lr[-1].shape[0] == net.num_views == net.im_feat_list_hr[0].shape[0] == calib_tensor.shape[0]:
        fl = net.im_feat_list_lr[-1].to(dev).permute(0, 2, 3, 1).contiguous()
        fh = net.im_feat_list_hr[0].to(dev).permute(0, 2, 3, 1).contiguous()
        zmul, zdiv = net._zscale()
        vh, vl = native.query_grid_views(0, R, R, R, mat[:3].reshape(-1), net._calib_rows(calib_tensor, None), net.projection_mode, zmul,
                                         zdiv, fl, fh, net._mlp_blob(), net._workspace())
        return vh, vl, mat
    M = torch.from_numpy(np.asarray(mat, np.float64)).to(dev)
    vh = torch.empty(total, dtype=torch.float32, device=dev)
    vl = torch.empty_like(vh)
    calib = calib_tensor.to(dev)
    for s in range(0, total, num_samples):
        f = torch.arange(s, min(total, s + num_samples), device=dev, dtype=torch.int64)
        samples = _grid_points(M, f, R).unsqueeze(0).repeat(net.num_views, 1, 1)
        net.query_mr(samples, calib)
        net.query_sr(samples, calib)
        phr, plr = net.get_preds()
        vh[s:s + f.numel()] = phr[0, 0]
        vl[s:s + f.numel()] = plr[0, 0]
    return vh.view(R, R, R), vl.view(R, R, R), mat


def _grid_points(M, f, R):
    """float32 [3,n] world positions of the flat voxel indices f (int64 device tensor): np.matmul(coords_matrix, idx) in
    float64, then .float(), exactly as create_grid + eval_func do (lib/sdf.py:22-26, lib/mesh_util.py:24)."""
    i, j, k = (f // (R * R)).double(), ((f // R) % R).double(), (f % R).double()
    return torch.stack([(((M[r, 0] * i + M[r, 1] * j) + M[r, 2] * k) + M[r, 3]) for r in range(3)]).float()


def eval_volumes_octree_views(opt, net, calib_tensor, resolution, b_min, b_max, transform=None, init_resolution=64):
    """eval_grid_octree for num_views > 1 / the perspective projection: the same level walk on the device (selection,
    scatter and cell pass are the single-view kernels - they only see the two fields), the lattice points evaluated by
    eval_func's multi-view recipe (lib/mesh_util.py:20-28: points repeated per view, query_mr + query_sr, view 0 kept)."""
    R = int(resolution)
    _, mat = create_grid(R, R, R, b_min, b_max, transform=transform)
    dev = net._device()
    M = torch.from_numpy(np.asarray(mat, np.float64)).to(dev)
    calib = calib_tensor.to(dev)

    def evaluate(idx):
        samples = _grid_points(M, idx, R).unsqueeze(0).repeat(net.num_views, 1, 1)
        net.query_mr(samples, calib)
        net.query_sr(samples, calib)
        phr, plr = net.get_preds()
        return phr[0, 0], plr[0, 0]

    vh, vl = native.octree_volumes(R, mat[:3].reshape(-1), None, 0.0, 1.0, None, None, None, net._workspace(), opt.threshold,
                                   init_resolution, evaluate=evaluate, device=dev)
    return vh, vl, mat


def eval_volumes_octree(opt, net, calib_tensor, resolution, b_min, b_max, transform=None, init_resolution=64, features=None):
    """eval_grid_octree: float64 device volumes (sdf_hr, sdf_lr) and the grid matrix."""
    _, mat = create_grid(resolution, resolution, resolution, b_min, b_max, transform=transform)
    calib = calib_tensor[0].detach().to("cpu", torch.float32).numpy().reshape(-1)[:12]
    fl, fh = features if features is not None else net.features()
    zmul, zdiv = net._zscale()
    # The levels are fp32-grade whatever --precision says (default): a flat / not-flat decision of the walk flips where a cell's corner
    # range is within the evaluator's error of --threshold, and a flipped block is interpolated instead of evaluated - with bf16's
    # 7e-3 on the occupancies 5.6 % of the vertices of the body field's octree mesh move by more than half a voxel (most of an octree
    # mesh is the walk's artefact surfaces; tests/test_gpu_octree.py).  --octree_precision sweep: the levels in --precision's arithmetic
    # (the 16-bit column kernel; 0.066 against 0.085 s at 512^3 in bf16).
    prec = getattr(opt, "precision", "fp32")
    if str(getattr(opt, "octree_precision", "fp32")) != "sweep" or native.wide_operands_active():
        prec = "fp32"
    blob = net._mlp_blob()
    if prec != "fp32" and native.DTYPES[prec] != net._core_dtype:
        raise ValueError("network was packed for %s, reconstruction asked for %s: set opt.precision before loading" % (net.precision, prec))
    vh, vl = native.octree_volumes(resolution, mat[:3].reshape(-1), calib, zmul, zdiv, fl, fh, blob, net._workspace(),
                                   opt.threshold, init_resolution, dtype=prec)
    return vh, vl, mat


def mesh_from_volume(net, vol, mat, level=0.5, want_normals=True):
    """marching_cubes_lewiner(vol, level) + index->world transform; numpy outputs like the reference.
    want_normals=False skips normals/values (gen_mesh discards them: lib/train_util.py:72) and returns None for them."""
    ws = net._workspace()
    if vol.dtype == torch.float64:
        vol = native.f64_to_f32(vol)   # marching_cubes_lewiner converts its input to float32
    v, f, n, val = native.marching_cubes_lewiner(vol, level, ws, want_normals=want_normals)
    vw = native.transform_points(v, mat[:3].reshape(-1))
    return tuple(ws.to_host([vw, f, n, val]))


def meshes_from_volumes(net, vols, mat, level=0.5, want_normals=True):
    """mesh_from_volume for several fields: the device -> host copy of one mesh runs (on the workspace's copy stream)
    while the next field's marching cubes executes.  Returns the concatenated tuples, in order."""
    ws = net._workspace()
    pending = []
    for k, vol in enumerate(vols):
        if vol.dtype == torch.float64:
            vol = native.f64_to_f32(vol)
        v, f, n, val = native.marching_cubes_lewiner(vol, level, ws, want_normals=want_normals, key=k)
        vw = native.transform_points(v, mat[:3].reshape(-1))
        pending.append(ws.to_host_async([vw, f, n, val]))
    out = ()
    for p in pending:
        out += tuple(p.result())
    return out


SLAB_COLUMNS = 32768   # the library's COL_BATCH (csrc/surs_query.hip): columns per launch of the column kernel


def sweep_schedule(nplanes, big, equal=None):
    """[(a, b)] plane ranges of the launches of a streamed sweep over `nplanes` axis-0 planes: launches of `big` planes (one batch of
    the column kernel), and the LAST batch's planes as a taper (5/8, 1/4, 1/8 of it) - what follows the sweep, the extraction and
    the copies of the last launch's cell layers, shrinks with it (mesh tail 1.3 -> 0.65 ms at 512^3).  equal: that many planes per
    launch instead (timing experiments, tests)."""
    if equal:
        return [(a, min(nplanes, a + equal)) for a in range(0, nplanes, equal)]
    cuts, a = [], 0
    while nplanes - a > big:
        cuts.append(big)
        a += big
    rem = nplanes - a
    cuts += [c for c in (rem * 5 // 8, rem // 4, rem - rem * 5 // 8 - rem // 4) if c > 0] if rem >= 8 else [rem]
    sched, a = [], 0
    for c in cuts:
        sched.append((a, a + c))
        a += c
    return sched


def reconstruction_streamed(opt, net, calib_tensor, resolution, b_min, b_max, transform=None, want_normals=True, timing=None,
                            planes=None, features=None, after_enqueue=None):
    """Dense reconstruction with the mesh extraction pipelined into the sweep: the sweep writes the volumes 32 768 columns
    (whole axis-0 planes) per launch; after every launch the cell layers that have become final are extracted
    (surs_mc_lewiner_range: Lewiner's sweep has axis 0 outermost, so their vertex / face numbers are final too) and their
    vertices and faces travel to the host under the next launches.  Same outputs as eval_volumes + meshes_from_volumes.
    Returns None if it cannot run (first extraction of this workspace: no buffer sizes yet; multi-view; octree).
    features: (Img feat_lr, Img feat_hr) instead of the model's current ones; after_enqueue: called once the whole sweep is
    enqueued and before the host starts driving the extraction (gen_mesh_pipelined enqueues the next subject's encoder there)."""
    ws = net._workspace()
    if ws.mc_capacity.get(0) is None or ws.mc_capacity.get(1) is None or net.num_views != 1:
        return None
    R = int(resolution)
    _, mat = create_grid(R, R, R, b_min, b_max, transform=transform)
    calib = calib_tensor[0].detach().to("cpu", torch.float32).numpy().reshape(-1)[:12]
    fl, fh = features if features is not None else net.features()
    zmul, zdiv = net._zscale()
    prec = getattr(opt, "precision", "fp32")
    blob = net._mlp_blob()
    if prec != "fp32" and native.DTYPES[prec] != net._core_dtype:
        raise ValueError("network was packed for %s, reconstruction asked for %s: set opt.precision before loading" %
                         (net.precision, prec))
    dev = blob.device
    vh = torch.empty((R, R, R), dtype=torch.float32, device=dev)
    vl = torch.empty_like(vh)
    streams = [native.MeshStream(ws, 0, vh, mat[:3].reshape(-1), 0.5, want_normals),
               native.MeshStream(ws, 1, vl, mat[:3].reshape(-1), 0.5, want_normals)]
    if planes is None and settings.get("SURS_SLAB_COLUMNS"):
        planes = max(1, int(settings.get("SURS_SLAB_COLUMNS")) // R)   # equal slabs of that many columns (timing experiments)
    # the whole sweep is enqueued first (no host synchronisation in it), with an event behind every slab ...
    sweep = torch.cuda.current_stream(dev)
    done = []
    kern = native.grid_kernel_for(R, R, R, mat[:3].reshape(-1), calib, zmul, zdiv, fl, fh, blob, prec, ws)
    sched = sweep_schedule(R, max(1, SLAB_COLUMNS // R), planes)
    for i0, i1 in sched:
        try:
            native.query_grid(i0, i1, R, R, mat[:3].reshape(-1), calib, zmul, zdiv, fl, fh, blob, prec, ws, vh[i0:i1], vl[i0:i1],
                              kernel=kern)
        except native._lib.SursError as e:
            if e.code != -3:
                raise
            prec = "fp32"   # general calibration / grid transform: the column kernel does not apply
            native.query_grid(i0, i1, R, R, mat[:3].reshape(-1), calib, zmul, zdiv, fl, fh, blob, prec, ws, vh[i0:i1], vl[i0:i1])
        ev = torch.cuda.Event()
        ev.record(sweep)
        done.append((i1, ev))
    if timing is not None:
        timing.record()
    if after_enqueue is not None:
        after_enqueue()
    # ... then the extraction follows it slab by slab on the fields' own streams: once planes < i1 are final the cell
    # layers below i1 - 1 are (a cell layer needs the plane above it); the count read-backs of the extraction wait on
    # those streams only, and its kernels run in the gaps and tails of the sweep's launches
    for i1, ev in done[:-1]:
        for s in streams:
            s.advance(i1 - 1, after=ev)
    outs = [s.finish(after=done[-1][1]) for s in streams]
    if any(o is None for o in outs):   # a buffer was too small: extract the finished volumes in one piece
        return meshes_from_volumes(net, [vh, vl], mat, want_normals=want_normals)
    return outs[0] + outs[1]


def reconstruction(opt, net, cuda, calib_tensor, resolution, b_min, b_max, use_octree=False, num_samples=50000,
                   transform=None, want_normals=True, features=None, after_enqueue=None):
    """-> verts_hr, faces_hr, normals_hr, values_hr, verts_lr, faces_lr, normals_lr, values_lr  (numpy).
    features / after_enqueue: see reconstruction_streamed (single view only).

    Range: with `--precision fp32` every fp32-grade product runs on two f16 parts per operand (|x| < 65504); the reference is
    plain fp32.  A field that comes out non-finite (marching cubes reports it: NonFiniteVolumeError) is computed again on three
    bf16 parts - fp32's exponent range - in every branch: the dense sweep on the per-point layer kernels ('fp32x'), the octree and
    multi-view sweeps under native.wide_operands(), and the encoder re-run the same way when its features are what overflowed."""
    called = []
    hook = (lambda: (called.append(1), after_enqueue())) if after_enqueue is not None else None

    def run(wide):
        if net.num_views > 1 or getattr(net, "projection_mode", "orthogonal") != "orthogonal":
            # multi-view / perspective: the per-point layer kernels (surs_query_points_views) behind the same two sweeps
            if use_octree:
                vh, vl, mat = eval_volumes_octree_views(opt, net, calib_tensor, resolution, b_min, b_max, transform)
            else:
                vh, vl, mat = eval_volumes_views(opt, net, calib_tensor, resolution, b_min, b_max, transform)
        elif use_octree:
            vh, vl, mat = eval_volumes_octree(opt, net, calib_tensor, resolution, b_min, b_max, transform, features=features)
        elif wide:
            vh, vl, mat = eval_volumes(opt, net, calib_tensor, resolution, b_min, b_max, transform, precision="fp32x", features=features)
        else:
            out = reconstruction_streamed(opt, net, calib_tensor, resolution, b_min, b_max, transform, want_normals, features=features,
                                          after_enqueue=hook)
            if out is not None:
                return out
            vh, vl, mat = eval_volumes(opt, net, calib_tensor, resolution, b_min, b_max, transform, features=features)
        if hook is not None and not called:
            hook()
        return meshes_from_volumes(net, [vh, vl], mat, want_normals=want_normals)

    try:
        return run(False)
    except native._lib.NonFiniteVolumeError:
        if getattr(opt, "precision", "fp32") != "fp32":
            raise   # (bf16 / fp16 were asked for explicitly: their range is theirs)
        import warnings
        warnings.warn("reconstruction: non-finite occupancies from the two-part f16 operand split; repeating on three bf16 parts "
                      "(fp32's exponent range)", stacklevel=2)
        if hook is not None and not called:
            hook()
        with native.wide_operands():
            if features is None:
                fl, fh = net.features()
                if not (bool(torch.isfinite(fl.buf).all()) and bool(torch.isfinite(fh.buf).all())) and not net.reencode_wide():
                    raise   # (features this object did not encode from its last images: nothing to re-encode)
            return run(True)


def _obj_text(verts, faces):
    v = np.asarray(verts, np.float64)
    f = np.asarray(faces, np.int64) + 1
    vs = "".join("v %.4f %.4f %.4f\n" % (a, b, c) for a, b, c in v)
    fs = "".join("f %d %d %d\n" % (a, c, b) for a, b, c in f)   # winding swapped: f0 f2 f1, 1-based
    return vs + fs


def save_obj_mesh(mesh_path, verts, faces):
    """Same bytes as the reference writer (lib/mesh_util.py:53-61), formatted natively and in parallel
    (surs_save_obj_mesh); `_obj_text` is the plain-Python statement of the format, kept for the tests."""
    import ctypes as C
    v = np.ascontiguousarray(verts, np.float64).reshape(-1, 3)
    f = np.ascontiguousarray(faces, np.int32).reshape(-1, 3)
    native.check(native.lib().surs_save_obj_mesh(str(mesh_path).encode(), v.ctypes.data_as(C.c_void_p), len(v),
                                                 f.ctypes.data_as(C.c_void_p), len(f), 0))


def save_obj_mesh_with_color(mesh_path, verts, faces, colors):
    with open(mesh_path, "w") as fh:
        for v, c in zip(verts, colors):
            fh.write("v %.4f %.4f %.4f %.4f %.4f %.4f\n" % (v[0], v[1], v[2], c[0], c[1], c[2]))
        for f in np.asarray(faces, np.int64) + 1:
            fh.write("f %d %d %d\n" % (f[0], f[2], f[1]))
