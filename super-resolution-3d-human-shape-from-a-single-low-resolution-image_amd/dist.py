"""Multi-GPU sharding of ONE dense reconstruction (BASELINE configs[3]): one process per GPU, contiguous slabs along array
axis 0 (world x, the slowest index of create_grid's flattening, /root/reference/lib/sdf.py:14-15,28, and the outermost
loop of Lewiner's sweep).  The reference has no distributed code; this is new.

Every rank sweeps its slab AND extracts its slab's part of both meshes while it sweeps (surs_mc_lewiner_range_slab), so
rank 0 receives meshes, not volumes.  Exchange steps (RCCL over xGMI when the backend is "nccl"; the same code runs on
gloo with host staging in the tests):

  1. halo        rank r+1 -> r   the first plane of each field (2 x R^2 fp32), as soon as the first launch has written it:
                                 the last cell layer of a slab needs the plane above it
  2. counts      all_gather      (n_verts, n_faces, min, max) per field: exclusive vertex offsets, the level-range check
  3. boundary    rank r -> r+1   the vertex ids of the x- / y-edges in the slab's top plane (2 x 2 x R^2 int32): the first
                                 cell layer of slab r+1 references vertices that slab r created
  4. meshes      rank r -> dst   vertices (world space, float64) and faces (int32, whole-mesh numbering): on one node every rank
                                 copies its part over its OWN PCIe link into a POSIX shared-memory block that `dst` maps
                                 (SharedMeshStore; 8 links instead of 0.7 GB funnelled through rank dst's xGMI links and then its
                                 one PCIe link); across nodes point-to-point to dst

Concatenated in rank order the result is bit-identical to the single-GPU extraction (tests/test_gpu_dist.py): vertex
coordinates are computed in whole-grid coordinates, a slab numbers its vertices in sweep order, and the faces are renumbered
with the exclusive sums of the vertex counts.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import settings


def slab_range(resolution, rank, world):
    """[i0, i1) of rank `rank`: contiguous, ordered, sizes differ by at most one."""
    base, rem = divmod(resolution, world)
    i0 = rank * base + min(rank, rem)
    return i0, i0 + base + (1 if rank < rem else 0)


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _host_staged(t, group=None):
    """gloo moves host memory only: device tensors are staged through the host there (tests); RCCL takes them as they are."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _global_rank(group, r):
    """P2POp peers are ranks of the default group: translate a rank of `group`."""
    return r if group is None else dist.get_global_rank(group, r)


class Exchange:
    """One batch of point-to-point transfers (grouped like ncclGroupStart/End, so a ring of sends and receives cannot
    deadlock).  start() enqueues them behind the work already on the current stream; wait() makes the current stream
    (RCCL) or the host (gloo) wait for them."""

    def __init__(self, group=None):
        self.group, self.ops, self.keep, self.post, self.works = group, [], [], [], []

    def send(self, t, dst):
        t = t.contiguous()
        if _host_staged(t, self.group):
            t = t.cpu()     # synchronises: the producer of t has finished
        self.keep.append(t)
        self.ops.append(dist.P2POp(dist.isend, t, _global_rank(self.group, dst), self.group))

    def recv(self, out, src):
        assert out.is_contiguous()
        if _host_staged(out, self.group):
            tmp = torch.empty(out.shape, dtype=out.dtype)
            self.post.append((out, tmp))
            out = tmp
        self.keep.append(out)
        self.ops.append(dist.P2POp(dist.irecv, out, _global_rank(self.group, src), self.group))

    def start(self):
        self.works = dist.batch_isend_irecv(self.ops) if self.ops else []
        return self

    def wait(self):
        for w in self.works:
            w.wait()
        for out, tmp in self.post:
            out.copy_(tmp)
        self.works, self.post = [], []


_host_groups = {}


def _host_group(group):
    """The group the tiny host-side rows (counts, status and ok flags) travel through.  With RCCL as the data path that is a gloo
    group over the same ranks, created collectively at the first row exchange of the job: a rank whose device has just faulted can
    still tell the others (a device-side all_gather would fail with the sticky error and leave the peers in the collective until
    the NCCL timeout), and the rows do not queue behind the sweep on the device.  Only for the default group (new_group must be
    entered by every process of the job); a caller's sub-group keeps the device path."""
    if group is not None or dist.get_backend() == "gloo":
        return group
    key = dist.distributed_c10d._get_default_group()
    g = _host_groups.get(key)
    if g is None:
        _host_groups.clear()   # (a destroyed default group leaves its entry behind)
        g = _host_groups[key] = dist.new_group(backend="gloo")
    return g


def _warn_safely(msg, stacklevel=3):
    """warnings.warn that never raises (-W error / filterwarnings = error): these warnings are issued on ONE rank behind a point where
    the ranks have agreed what to do next - an exception there would leave the peers waiting in the step that follows."""
    import warnings
    try:
        warnings.warn(msg, stacklevel=stacklevel + 1)
    except Exception:   # noqa: BLE001
        import sys
        print("warning: " + msg, file=sys.stderr)


def init_host_group(group=None):
    """Create the host-side (gloo) group of an RCCL job NOW - collectively, right after init_process_group - instead of at the first
    row exchange inside a reconstruction: new_group must be entered by every rank, and a rank that has already failed by then would
    leave its peers waiting in it.  bench.py and the eval driver call it at start-up; without it the group is still created lazily."""
    return _host_group(group)


def all_gather_rows(row, device, group=None):
    """row: sequence of floats -> float64 array [world, len(row)] (one tiny all_gather, on the host: _host_group; a sub-group of an
    RCCL job gathers on the device)."""
    world, _ = _world(group)
    hg = _host_group(group)
    on = device if dist.get_backend(hg) != "gloo" else torch.device("cpu")
    mine = torch.tensor([float(v) for v in row], dtype=torch.float64, device=on)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=hg)
    return torch.stack(out).cpu().numpy()


def gather_slabs(local, resolution, dst=0, group=None):
    """local: [n_rank, R, R] tensor of this rank's slab.  Returns the full [R, R, R] tensor on `dst`, None elsewhere: the
    volume-level exchange (used when normals are wanted, and by tools that need the whole field on one rank).  The slabs
    land directly in their place of one preallocated volume - no per-rank buffers, no concatenation copy."""
    world, rank = _world(group)
    if world == 1:
        return local
    ex = Exchange(group)
    full = None
    if rank == dst:
        full = torch.empty((resolution,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        for r in range(world):
            a, b = slab_range(resolution, r, world)
            if r == rank:
                full[a:b].copy_(local)
            elif b > a:
                ex.recv(full[a:b], r)
    elif local.shape[0] > 0:
        ex.send(local, dst)
    ex.start().wait()
    return full


class SharedMeshStore:
    """Host memory that every rank of one node can write and `dst` can read: a file in /dev/shm, mapped by each process and
    (on a GPU box) registered with the HIP runtime as pinned memory, so that a rank's device-to-host copy of its part of a mesh
    is one DMA over its own PCIe link.  One store per (creator pid, creator token, tag); grows on demand; the creator unlinks it
    at exit.  The space is RESERVED (posix_fallocate: tmpfs hands out pages lazily, and a write into a hole of a full /dev/shm is
    a SIGBUS, not an exception); the name carries a random per-process token of the creator; the creator opens with
    O_CREAT | O_EXCL | O_NOFOLLOW, the others with O_NOFOLLOW and check owner and file type.  Every failure here is an ordinary
    exception: assemble_slab_meshes turns it into an agreed fallback to the point-to-point delivery."""
    _maps = {}
    _owned = set()
    TOKEN = None

    @classmethod
    def token(cls):
        if cls.TOKEN is None:
            import secrets
            cls.TOKEN = secrets.randbits(48)   # (travels in a float64 of the counts row: 53 bits are exact)
        return cls.TOKEN

    @staticmethod
    def _path(owner_pid, owner_token, tag):
        return "/dev/shm/surs_mesh_%d_%012x_%s" % (owner_pid, owner_token, tag)

    @classmethod
    def open(cls, owner_pid, owner_token, tag, nbytes, create):
        import atexit
        import mmap
        import os
        import stat
        path = cls._path(owner_pid, owner_token, tag)
        cur = cls._maps.get(path)
        if cur is not None and cur[1] >= nbytes and (create or os.path.exists(path)):
            return cur[0]
        if cur is not None:
            cls._release(path)
        if create:
            size = max(1 << 20, int(nbytes * 1.25))
            if path in cls._owned:
                fd = os.open(path, os.O_RDWR | os.O_NOFOLLOW)
            else:
                fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR | os.O_NOFOLLOW, 0o600)
                cls._owned.add(path)
                atexit.register(lambda p=path: os.path.exists(p) and os.unlink(p))
            try:
                if os.fstat(fd).st_size < size:
                    os.posix_fallocate(fd, 0, size)   # reserves the pages: ENOSPC here instead of SIGBUS in copy_
            except BaseException:
                os.close(fd)
                try:            # (no half-grown block left behind: the next reconstruction starts from nothing)
                    os.unlink(path)
                except OSError:
                    pass
                cls._owned.discard(path)
                cls._maps.pop(path, None)
                raise
        else:
            fd = os.open(path, os.O_RDWR | os.O_NOFOLLOW)
            st = os.fstat(fd)
            if not stat.S_ISREG(st.st_mode) or st.st_uid != os.getuid():
                os.close(fd)
                raise PermissionError("%s is not a regular file of this user" % path)
        try:
            size = os.fstat(fd).st_size
            if size < nbytes:
                raise OSError("%s holds %d bytes, %d needed" % (path, size, nbytes))
            mm = mmap.mmap(fd, size)
        finally:
            os.close(fd)
        buf = torch.frombuffer(mm, dtype=torch.uint8)
        pinned = False
        if torch.cuda.is_available():
            try:   # page-lock the mapping: device-to-host copies into it then run at PCIe rate
                pinned = int(torch.cuda.cudart().cudaHostRegister(buf.data_ptr(), size, 0)) == 0
            except Exception:
                pinned = False
        cls._maps[path] = (buf, size, mm, pinned)
        return buf

    @classmethod
    def _release(cls, path):
        buf, size, mm, pinned = cls._maps.pop(path)
        if pinned:
            try:
                torch.cuda.cudart().cudaHostUnregister(buf.data_ptr())
            except Exception:
                pass
        del buf

    @classmethod
    def release_all(cls):
        for path in list(cls._maps):
            cls._release(path)


def node_identity():
    """A number that two processes share iff they run under the same kernel: crc32 of the boot id (the host name alone is the same
    on two nodes of a misconfigured cluster, and differs between containers of one node).  Sharing /dev/shm is then PROBED, not
    assumed: see assemble_slab_meshes."""
    import socket
    import zlib
    try:
        boot = open("/proc/sys/kernel/random/boot_id").read().strip()
    except OSError:
        boot = ""
    return float(zlib.crc32((boot + "|" + socket.gethostname()).encode()))


def slab_feature_columns(i0, i1, mat, calib, width):
    """Columns [lo, hi) of a `width`-column feature map that the bilinear taps of the x-slab [i0, i1) touch, or None when the
    projected image column is not a function of the voxel's axis-0 index alone (general calibration / grid transform).
    lib/geometry.py:4-31: X = calib[0,:3] . p + calib[0,3], pixel = (X + 1) / 2 * (width - 1); one column of slack each side."""
    m, c = np.asarray(mat, np.float64).reshape(3, 4), np.asarray(calib, np.float64).reshape(-1)[:12].reshape(3, 4)
    row = c[0, :3] @ m[:, :3]                      # dX / d(i, j, k)
    if row[1] != 0.0 or row[2] != 0.0:
        return None
    x0 = c[0, :3] @ m[:, 3] + c[0, 3]
    us = [((row[0] * i + x0) + 1.0) / 2.0 * (width - 1) for i in (i0, i1 - 1)]
    lo, hi = int(np.floor(min(us))) - 1, int(np.floor(max(us))) + 3
    return max(0, lo), min(width, hi)


def encode_sharded(net, images, calib_tensor, resolution, b_min, b_max, transform=None, group=None):
    """The encoder for ONE subject on all ranks of a slab-mode reconstruction (call it instead of super_res -> filter_hr ->
    filter_lr; every rank passes the same image).  What is sharded and what is not, and why (NOTES.md R4.0a item 1):

      super_res  (SuRSSR_v3.py:143-181, no normalisation, receptive field 119 columns of the 2W map): rank r runs it on the image
                 strip its share of the columns depends on (encoder.super_res_strip: share + a 128-column halo each side, recomputed,
                 no exchange) - bit-identical to the columns of the full maps;
      feature_lr strips are all-gathered (67 MB at H = 512, the one collective of the encoder: RCCL over xGMI / gloo in the tests);
      filter_lr  (three stacked hourglasses: its receptive field covers the whole map and its 78 GroupNorms need global statistics)
                 runs replicated on the gathered feature_lr;
      filter_hr  (one 1x1 convolution) runs on the rank's strip of feature_hr only: the rank's x-slab samples no other columns.

    Leaves net.im_feat_list_lr / im_feat_list_hr as the replicated encoder would, except that im_feat_hr holds zeros outside the
    rank's strip.  Falls back to the replicated encoder (returns False) for one rank, several views, a general calibration, or a
    width the ranks cannot share in even strips."""
    from . import encoder, native
    from .model import _as_img, _as_nchw_view
    from .sdf import create_grid
    world, rank = _world(group)
    R = int(resolution)

    def replicated():
        _, f_lr, f_hr = net.super_res(images)
        net.filter_hr(f_hr)
        net.filter_lr(f_lr)
        return False

    if world == 1 or net.num_views != 1 or images.shape[0] != 1:
        return replicated()
    x = _as_img(images[0:1].to(net._device()))
    wl, wh = x.w // 2, 2 * x.w
    _, mat = create_grid(R, R, R, b_min, b_max, transform=transform)
    calib = calib_tensor[0].detach().to("cpu", torch.float64).numpy().reshape(-1)[:12]
    i0, i1 = slab_range(R, rank, world)
    need_hr = slab_feature_columns(i0, i1, mat[:3], calib, wh)
    if need_hr is None or wl % (2 * world) != 0:
        return replicated()
    share = wl // world
    a, b = rank * share, (rank + 1) * share
    a, b = min(a, need_hr[0] // 4 // 2 * 2), max(b, -(-need_hr[1] // 4) + (-(-need_hr[1] // 4)) % 2)   # + what the slab samples of feature_hr (even bounds)
    b = min(b, wl)
    W = net._encoder_weights()
    _, new2, new_fin = encoder.super_res_strip_g(W, x, a, b, want_image=False)   # (img_SR is not part of what this returns)
    dev = x.buf.device
    # ---- feature_lr: every rank's share, gathered
    mine = new2.buf.view(new2.h, new2.w, new2.c)[:, rank * share - a:(rank + 1) * share - a, :].contiguous()
    parts = [torch.empty_like(mine) for _ in range(world)]
    if _host_staged(mine, group):
        host = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(world)]
        dist.all_gather(host, mine.cpu(), group=group)
        parts = [h.to(dev) for h in host]
    else:
        dist.all_gather(parts, mine, group=group)
    # (gathered into a buffer the network keeps: filter_lr's captured graph reads its input in place - encoder.filter_lr_g)
    f_lr = getattr(net, "_gathered_lr", None)
    if f_lr is None or tuple(f_lr.shape) != (new2.h, wl, new2.c) or f_lr.device != dev:
        f_lr = net._gathered_lr = torch.empty((new2.h, wl, new2.c), dtype=torch.float32, device=dev)
    torch.cat(parts, dim=1, out=f_lr)
    feature_lr = native.Img(new2.h, wl, new2.c, buf=f_lr.reshape(-1), device=dev)
    encoder.persistent(feature_lr)
    # ---- filter_lr replicated, filter_hr on the strip (placed in a zeroed full-size map: the sweep addresses absolute columns)
    outs = encoder.filter_lr_g(W, feature_lr, keep_all=net.training)
    net._feat_lr_imgs = [[o] for o in outs]
    net.im_feat_list_lr = [_as_nchw_view(o) for o in outs]
    strip = encoder.filter_hr(W, new_fin)[0]
    full = torch.zeros((new_fin.h, wh, strip.c), dtype=torch.float32, device=dev)
    full[:, 4 * a:4 * b, :] = strip.buf.view(strip.h, strip.w, strip.c)
    hr = native.Img(new_fin.h, wh, strip.c, buf=full.reshape(-1), device=dev)
    net._feat_hr_imgs = [[hr]]
    net.im_feat_list_hr = [_as_nchw_view(hr)]
    net.im_SR = net.feature_hr = None
    net.feature_lr = _as_nchw_view(feature_lr)
    net._last_images = None          # (reencode_wide re-runs super_res on one device: not how these features came about)
    net._sr_out = net._lr_from = net._hr_from = None
    # what reconstruction_sharded's retry after an f16 overflow re-runs on every rank, under native.wide_operands()
    # (the produced tensor itself, compared with `is`: an address can come back under another subject's features)
    net._sharded_encode = (images, calib_tensor, resolution, b_min, b_max, transform, group, net.im_feat_list_lr[-1])
    return True


def offsets_from_counts(counts):
    """counts [world] -> exclusive prefix sums (int64): where each rank's vertices / faces start in the whole mesh."""
    c = np.asarray(counts, np.int64)
    return np.concatenate([[0], np.cumsum(c)[:-1]])


def reconstruction_sharded(opt, net, calib_tensor, resolution, b_min, b_max, transform=None, dst=0, want_normals=False,
                           timing=None, group=None, copy_out=True):
    """copy_out=False: on one node `dst` gets VIEWS of the shared-memory blocks the ranks delivered their parts into instead of
    owning arrays - valid until the next sharded reconstruction of this process overwrites them (saves one host copy of the meshes).

    reconstruction_sharded_once, and - as mesh_util.reconstruction does on one GPU - once more on three bf16 parts per operand
    (fp32's exponent range) when the fp32-grade sweep produced non-finite occupancies on any rank (every rank raises and
    repeats together: the counts exchange carries the flag)."""
    from . import native
    if _world(group)[0] == 1:   # (mesh_util.reconstruction repeats a non-finite sweep itself: no second retry around it)
        return reconstruction_sharded_once(opt, net, calib_tensor, resolution, b_min, b_max, transform, dst, want_normals, timing, group)
    try:
        return reconstruction_sharded_once(opt, net, calib_tensor, resolution, b_min, b_max, transform, dst, want_normals, timing, group,
                                           copy_out=copy_out)
    except native._lib.NonFiniteVolumeError:
        if getattr(opt, "precision", "fp32") != "fp32":
            raise
        import warnings
        warnings.warn("reconstruction_sharded: non-finite occupancies from the two-part f16 operand split; repeating on three "
                      "bf16 parts", stacklevel=2)
        with native.wide_operands():
            fl, fh = net.features()
            bad = not (bool(torch.isfinite(fl.buf).all()) and bool(torch.isfinite(fh.buf).all()))
            # features of the sharded encoder are re-made by the sharded encoder - a collective: every rank must take that branch, so the
            # branch travels with the flag and is taken only if EVERY rank holds a sharded encoder's features - those of the replicated
            # one by reencode_wide
            se = getattr(net, "_sharded_encode", None)
            sharded = se is not None and bool(net.im_feat_list_lr) and net.im_feat_list_lr[-1] is se[7]
            flags = all_gather_rows([1.0 if bad else 0.0, 1.0 if sharded else 0.0], fl.buf.device, group)
            if flags[:, 0].any():
                if flags[:, 1].all():
                    encode_sharded(net, *se[:7])
                elif not net.reencode_wide():
                    raise
            return reconstruction_sharded_once(opt, net, calib_tensor, resolution, b_min, b_max, transform, dst, want_normals, timing,
                                               group, wide=True, copy_out=copy_out)


def reconstruction_sharded_once(opt, net, calib_tensor, resolution, b_min, b_max, transform=None, dst=0, want_normals=False,
                                timing=None, group=None, wide=False, copy_out=True):
    """One reconstruction on all ranks of the group: each rank sweeps the x-slab slab_range(R, rank, world) and extracts its
    part of the two meshes.  Returns the 8-tuple of mesh_util.reconstruction on `dst` (normals / values None), None on the
    other ranks.  want_normals=True: the slabs' VOLUMES are gathered on `dst`, which extracts with normals and values
    (_reconstruction_sharded_with_normals).  Every rank raises the same ValueError / RuntimeError as marching_cubes_lewiner when the level is outside
    the whole volume's range / there is no surface."""
    from . import mesh_util, native
    from .sdf import create_grid
    world, rank = _world(group)
    R = int(resolution)
    if world == 1:
        return mesh_util.reconstruction(opt, net, net._device(), calib_tensor, R, b_min, b_max, use_octree=False,
                                        transform=transform, want_normals=want_normals)
    if R < 2 * world:
        raise ValueError("resolution %d is too small for %d slabs (every rank needs at least two planes)" % (R, world))
    i0, i1 = slab_range(R, rank, world)
    nloc, halo = i1 - i0, rank < world - 1
    _, mat = create_grid(R, R, R, b_min, b_max, transform=transform)
    m12 = mat[:3].reshape(-1)
    calib = calib_tensor[0].detach().to("cpu", torch.float32).numpy().reshape(-1)[:12]
    fl, fh = net.features()
    zmul, zdiv = net._zscale()
    prec = "fp32x" if wide else getattr(opt, "precision", "fp32")
    blob = net._mlp_blob()
    ws = net._workspace()
    dev = blob.device
    vols = [torch.empty((nloc + (1 if halo else 0), R, R), dtype=torch.float32, device=dev) for _ in range(2)]
    keys = ("slab", 0), ("slab", 1)
    first = any(ws.mc_capacity.get(k) is None for k in keys)
    streams = None if first else [native.MeshStream(ws, k, v, m12, 0.5, False, zoff=i0) for k, v in zip(keys, vols)]
    # A failure on one rank (out of memory, a kernel error, ...) must not leave the others waiting in the next collective: the
    # halo exchange is the only collective step inside this block and comes first; whatever is raised behind it is carried to the
    # counts exchange as a status flag, where every rank raises.
    status, failure = 0, None
    res = tables = runs = ex = None

    def halo_exchange():
        e = Exchange(group)
        for v in vols:
            if rank > 0:
                e.send(v[0], rank - 1)
            if halo:
                e.recv(v[nloc], rank + 1)
        return e.start()

    # the column-kernel choice probes the grid's MIDDLE plane: the rank whose slab holds it decides for everybody (with a sharded
    # encoder - encode_sharded - the other ranks do not have that plane's feature_hr columns).  A one-element row exchange of its own,
    # in FRONT of the guarded block: every rank enters it whatever happens later, and what is raised behind it travels with the
    # 13-element status row of the counts exchange
    owner = next(r for r in range(world) if slab_range(R, r, world)[0] <= R // 2 < slab_range(R, r, world)[1])
    k, kern_err = 0.0, None
    if rank == owner:
        try:
            k = float(native.grid_kernel_for(R, R, R, m12, calib, zmul, zdiv, fl, fh, blob, prec, ws))
        except Exception as e:   # noqa: BLE001 - the others are waiting in the gather below: tell them
            k, kern_err = -1.0, e
    k = all_gather_rows([k], dev, group)[owner, 0]
    if k < 0:
        raise kern_err or RuntimeError("the column-kernel probe failed on rank %d" % owner)
    kern = int(k)
    if want_normals:
        def sweep_slab():
            try:
                return native.query_grid(i0, i1, R, R, m12, calib, zmul, zdiv, fl, fh, blob, prec, ws, kernel=kern)
            except native._lib.SursError as e:
                if e.code != -3:
                    raise
                return native.query_grid(i0, i1, R, R, m12, calib, zmul, zdiv, fl, fh, blob, "fp32", ws)   # general calibration
        return _reconstruction_sharded_with_normals(net, sweep_slab, mat, R, dst, group)
    try:
        # ---- the sweep, enqueued in one go; the halo exchange starts behind the first launch
        import os
        # Launches of one column batch (32 768 columns), the last batch's planes as a taper (mesh_util.sweep_schedule); a slab that
        # is no more than one batch - a rank of eight at 512^3 - goes out as ONE launch: with the two-workgroup column kernel (512
        # workgroups) the small launches of a taper cost more than the shorter extraction tail gives back (round 5, one rank of
        # eight on a dedicated GPU: 16 384 + taper 15.9 ms of sweep + 0.76 of tail, one launch 15.0 + 0.85).  SURS_SLAB_COLUMNS:
        # equal launches of that many columns (tests)
        env = settings.get("SURS_SLAB_COLUMNS")
        big = max(1, 32768 // R)
        if env:
            sched = mesh_util.sweep_schedule(nloc, big, max(1, int(env) // R))
        elif nloc <= big:
            sched = [(0, nloc)]
        else:
            sched = mesh_util.sweep_schedule(nloc, big)
        sweep = torch.cuda.current_stream(dev)
        done = []
        for a, b in sched:
            try:
                native.query_grid(i0 + a, i0 + b, R, R, m12, calib, zmul, zdiv, fl, fh, blob, prec, ws, vols[0][a:b], vols[1][a:b],
                                  kernel=kern)
            except native._lib.SursError as e:
                if e.code != -3:
                    raise
                prec = "fp32"   # general calibration / grid transform: the column kernel does not apply
                native.query_grid(i0 + a, i0 + b, R, R, m12, calib, zmul, zdiv, fl, fh, blob, prec, ws, vols[0][a:b], vols[1][a:b])
            ev = torch.cuda.Event()
            ev.record(sweep)
            done.append((b, ev))
            if ex is None:
                ex = halo_exchange()
        ex.wait()                      # the sweep's stream (RCCL) / the host (gloo) has the halo planes from here on
        halo_ev = torch.cuda.Event()
        halo_ev.record(sweep)
        if timing is not None:
            timing.record()
        # ---- extraction: layer by layer behind the sweep's launches (from the second reconstruction on), or in one piece
        res = None
        if streams is not None:
            for b, ev in done[:-1]:
                for s in streams:
                    s.advance(b - 1, after=ev)
            res = [s.finish(after=halo_ev) for s in streams]
            if any(r is None for r in res):
                res = None
            else:
                runs = [s.run for s in streams]
                tables = [s.w for s in streams]
        if res is None:
            torch.cuda.current_stream(dev).synchronize()
            res, runs, tables = [], [], []
            for k, v in zip(keys, vols):
                world_v, faces, run, w = native.slab_mesh_one_piece(ws, k, v, m12, 0.5, i0)
                res.append((world_v, faces))
                runs.append(run)
                tables.append(w)
        counts = [(r.n_verts, r.n_faces, r.vmin, r.vmax) for r in runs]

    except native._lib.NonFiniteVolumeError:
        status = 1
    except Exception as e:   # noqa: BLE001 - reported to every rank below, re-raised on this one
        status, failure = 2, e
    if status and ex is None:
        halo_exchange().wait()   # (failed before the halo step: the neighbours are waiting in theirs)
    elif status:
        try:                     # (failed between the start of the halo step and its wait: the posted transfers must complete,
            ex.wait()            #  or the peers' matching sends / receives never do)
        except Exception:        # noqa: BLE001 - a sticky device error: the status flag below still reaches the peers through gloo
            pass
    if status:
        nan = float("nan")
        counts = [(0, 0, nan, nan)] * 2
        res = [(torch.empty((0, 3), dtype=torch.float64, device=dev), torch.empty((0, 3), dtype=torch.int32, device=dev))] * 2
        tables = [None, None]
    n0 = vols[0].shape[0]

    def top_ids(f):
        top = torch.empty((2, R, R), dtype=torch.int32, device=dev)
        native.check(native.lib().surs_mc_slab_top_ids(native._ptr(tables[f]), tables[f].numel(), n0, R, R, native._ptr(top),
                                                       native._stream()))
        return top

    def fixup(f, faces, own_off, below, below_off):
        native.check(native.lib().surs_mc_slab_fixup(native._ptr(faces), faces.shape[0], own_off, native._ptr(below), below_off,
                                                     native._stream()))

    out = assemble_slab_meshes(res, counts, top_ids, fixup, R, dev, dst, group, status=status, failure=failure, copy_out=copy_out)
    if out is None:
        torch.cuda.current_stream(dev).synchronize()   # the sends have left before the buffers go back to the allocator
        return None
    flat = [out[0][0], out[0][1], out[1][0], out[1][1]]
    if all(not t.is_cuda for t in flat):   # delivered through shared host memory (owning copies unless copy_out=False)
        host = [t.numpy() for t in flat]
    else:
        host = ws.to_host(flat)
    return host[0], host[1], None, None, host[2], host[3], None, None


def _reconstruction_sharded_with_normals(net, sweep_slab, mat, R, dst, group):
    """want_normals=True in slab mode (mesh_util.reconstruction's full 8-tuple; gen_mesh itself never asks for it,
    lib/train_util.py:72): a vertex normal accumulates over the faces of BOTH slabs at a boundary, so the volumes - not the
    meshes - are exchanged: every rank sweeps its x-slab (sweep_slab() -> (vol_hr, vol_lr) of its planes), gather_slabs puts the
    slabs in their place of one volume on `dst`, and `dst` extracts both meshes with normals and values from the whole volumes
    exactly as one GPU does.  What dst raises (marching_cubes_lewiner's ValueError / RuntimeError, a non-finite volume) is raised
    on every rank."""
    from . import mesh_util, native
    world, rank = _world(group)
    dev = net._device()
    i0, i1 = slab_range(R, rank, world)
    status, failure = 0.0, None
    try:
        vols = list(sweep_slab())
    except Exception as e:   # noqa: BLE001 - the others are waiting in the gather below
        status, failure = 2.0, e
        vols = [torch.zeros((i1 - i0, R, R), dtype=torch.float32, device=dev) for _ in range(2)]
    full = [gather_slabs(v, R, dst, group) for v in vols]
    out = None
    if rank == dst and not status:
        try:
            out = mesh_util.meshes_from_volumes(net, full, mat, want_normals=True)
        except native._lib.NonFiniteVolumeError as e:
            status, failure = 1.0, e
        except ValueError as e:
            status, failure = 3.0, e
        except RuntimeError as e:
            status, failure = 4.0, e
    codes = all_gather_rows([status], dev, group)[:, 0]
    if codes.any():
        if failure is not None:
            raise failure
        code = int(codes.max())
        if code == 1:
            raise native._lib.NonFiniteVolumeError("non-finite occupancies in the volume (reported by another rank)")
        if code == 3:
            raise ValueError("Surface level must be within volume data range.")
        if code == 4:
            raise RuntimeError("No surface found at the given iso value.")
        raise RuntimeError("reconstruction_sharded (normals): a peer rank failed")
    return out


def _agree(ok, dev, group):
    """all ranks report a flag; True iff every rank's is set (one tiny all_gather: also the barrier between two steps)"""
    return bool((all_gather_rows([1.0 if ok else 0.0], dev, group)[:, 0] == 1.0).all())


def _deliver_shared(res, allc, voff, owner_pid, owner_token, dev, dst, group, copy_out):
    """Step 4 of the module docstring through SharedMeshStore.  Returns the meshes on dst / None elsewhere, or False when the
    ranks agreed to fall back (a block could not be created, reserved or mapped, or a copy failed, on any rank)."""
    import warnings
    world, rank = _world(group)
    nfields = len(res)
    stores = []
    for f in range(nfields):
        nv, nf = allc[:, 4 * f].astype(np.int64), allc[:, 4 * f + 1].astype(np.int64)
        vb, fb = int(nv.sum()) * 3 * res[f][0].element_size(), int(nf.sum()) * 3 * 4
        stores.append((nv, nf, vb, fb))
    ok, why = True, None
    if rank == dst:
        try:
            for f, (nv, nf, vb, fb) in enumerate(stores):
                SharedMeshStore.open(owner_pid, owner_token, "f%d_v" % f, vb, True)
                SharedMeshStore.open(owner_pid, owner_token, "f%d_f" % f, fb, True)
        except Exception as e:   # noqa: BLE001 - ENOSPC of a small /dev/shm, EEXIST, EPERM, ...: agreed fallback
            ok, why = False, e
    if not _agree(ok, dev, group):        # the blocks exist (and are reserved) before anybody maps them
        if why is not None:
            _warn_safely("slab meshes: shared-memory delivery unavailable (%s); sending point to point" % (why,))
        return False
    out = []
    try:
        for f, (nv, nf, vb, fb) in enumerate(stores):
            bv = SharedMeshStore.open(owner_pid, owner_token, "f%d_v" % f, vb, False)
            bf = SharedMeshStore.open(owner_pid, owner_token, "f%d_f" % f, fb, False)
            out.append((bv[:vb].view(res[f][0].dtype).view(-1, 3), bf[:fb].view(torch.int32).view(-1, 3)))
    except Exception as e:   # noqa: BLE001 - dst's /dev/shm is not this rank's (two containers of one node), permissions, ...
        ok, why = False, e
    if not _agree(ok, dev, group):
        if why is not None:
            _warn_safely("slab meshes: rank %d cannot map dst's shared block (%s); sending point to point" % (rank, why))
        return False
    try:
        for f, (nv, nf, vb, fb) in enumerate(stores):
            V, F = out[f]
            va, fa = int(voff[f][rank]), int(offsets_from_counts(nf)[rank])
            if nv[rank]:
                V[va:va + int(nv[rank])].copy_(res[f][0])
            if nf[rank]:
                F[fa:fa + int(nf[rank])].copy_(res[f][1])
        if res[0][0].is_cuda:
            torch.cuda.current_stream(dev).synchronize()
    except Exception as e:   # noqa: BLE001
        ok, why = False, e
    if not _agree(ok, dev, group):        # every part has landed
        if why is not None:
            _warn_safely("slab meshes: copy into the shared block failed on rank %d (%s); sending point to point" % (rank, why))
        return False
    if rank != dst:
        return None
    # the blocks are overwritten by the next sharded reconstruction: hand out owning arrays like mesh_util.reconstruction does,
    # unless the caller takes the views and their lifetime (copy_out=False: bench.py, which drops the meshes at once)
    return [(V.clone(), F.clone()) for V, F in out] if copy_out else out


def assemble_slab_meshes(res, counts, top_ids, fixup, R, dev, dst=0, group=None, level=0.5, status=0, failure=None, copy_out=True):
    """Steps 2-4 of the module docstring.  res[f] = (verts [V,3], faces int32 [F,3]) of this rank's slab of field f in local
    numbering (references to the slab below as -(2 + slot)); counts[f] = (n_verts, n_faces, vmin, vmax); top_ids(f) -> int32
    [2,R,R] ids of the x- / y-edge vertices in the slab's top plane; fixup(f, faces, own_offset, below_ids, below_offset)
    renumbers `faces` in place.  Returns [(verts, faces)] per field of the whole mesh on `dst`, None on the other ranks.
    status: 0, 1 (this rank's slab holds NaN) or 2 (this rank failed with `failure`): travels with the counts, every rank raises.
    (The kernels behind top_ids / fixup are the product's on the GPU; the gloo test passes numpy stand-ins.)"""
    import os
    world, rank = _world(group)
    nfields = len(res)
    shm_ok = settings.get("SURS_SLAB_P2P") != "1" and os.path.isdir("/dev/shm")
    row = [float(status), node_identity(), float(os.getpid()), float(SharedMeshStore.token()), 1.0 if shm_ok else 0.0]
    for c in counts:
        row += list(c)
    NH = 5
    allh = all_gather_rows(row, dev, group)          # [world, NH + 4 * nfields]
    st, allc = allh[:, 0], allh[:, NH:]
    if (st == 2).any():
        if failure is not None:
            raise failure
        raise RuntimeError("the slab sweep failed on rank(s) %s" % np.nonzero(st == 2)[0].tolist())
    if (st == 1).any() or np.isnan(allc).any():   # some rank's slab holds NaN values: every rank raises
        from ._lib import NonFiniteVolumeError
        raise NonFiniteVolumeError("the occupancy volume contains NaN values (rank(s) %s)" %
                                   sorted(set(np.nonzero(np.isnan(allc))[0].tolist()) | set(np.nonzero(st == 1)[0].tolist())))
    # one node = the same kernel on every rank AND no rank asked for point to point (an environment variable set on one rank only
    # must not send the ranks down different paths: the flags travel with the counts)
    one_node = bool((allh[:, 1] == allh[0, 1]).all()) and bool((allh[:, 4] == 1.0).all())
    owner_pid, owner_token = int(allh[dst, 2]), int(allh[dst, 3])
    for f in range(nfields):
        lo, hi = allc[:, 4 * f + 2].min(), allc[:, 4 * f + 3].max()
        if level < lo or level > hi:
            raise ValueError("Surface level must be within volume data range.")
        if allc[:, 4 * f].sum() == 0:
            raise RuntimeError("No surface found at the given iso value.")
    voff = [offsets_from_counts(allc[:, 4 * f]) for f in range(nfields)]
    # ---- boundary vertex ids to the slab above, faces to the whole mesh's numbering
    ex = Exchange(group)
    below = [torch.empty((2, R, R), dtype=torch.int32, device=dev) if rank > 0 else None for _ in range(nfields)]
    for f in range(nfields):
        if rank < world - 1:
            ex.send(top_ids(f), rank + 1)
        if rank > 0:
            ex.recv(below[f], rank - 1)
    ex.start().wait()
    for f in range(nfields):
        fixup(f, res[f][1], int(voff[f][rank]), below[f], int(voff[f][rank - 1]) if rank > 0 else 0)
    # ---- meshes to dst: on one node through shared host memory, every rank over its own PCIe link.  Every step that can fail
    #      (creating / reserving the blocks on dst, mapping them elsewhere, the copies) is followed by an all_gather of ok flags
    #      in place of a bare barrier: unless EVERY rank reports ok, every rank takes the point-to-point path below
    if one_node:
        out = _deliver_shared(res, allc, voff, owner_pid, owner_token, dev, dst, group, copy_out)
        if out is not False:
            return out
    # ---- meshes to dst
    ex = Exchange(group)
    out = []
    for f in range(nfields):
        nv, nf = allc[:, 4 * f].astype(np.int64), allc[:, 4 * f + 1].astype(np.int64)
        foff = offsets_from_counts(nf)
        if rank == dst:
            V = torch.empty((int(nv.sum()), 3), dtype=res[f][0].dtype, device=dev)
            F = torch.empty((int(nf.sum()), 3), dtype=torch.int32, device=dev)
            for r in range(world):
                va, fa = int(voff[f][r]), int(foff[r])
                if r == rank:
                    V[va:va + int(nv[r])].copy_(res[f][0])
                    F[fa:fa + int(nf[r])].copy_(res[f][1])
                else:
                    if nv[r]:
                        ex.recv(V[va:va + int(nv[r])], r)
                    if nf[r]:
                        ex.recv(F[fa:fa + int(nf[r])], r)
            out.append((V, F))
        else:
            if nv[rank]:
                ex.send(res[f][0], dst)
            if nf[rank]:
                ex.send(res[f][1], dst)
    ex.start().wait()
    return out if rank == dst else None
