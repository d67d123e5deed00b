"""Multi-GPU sharding of the dense sweep: one process per GPU, contiguous slabs along array axis 0 (world x, the
slowest index of create_grid's flattening, /root/reference/lib/sdf.py:14-15,28), so concatenating the slabs in rank
order reproduces the single-GPU volume bit for bit.  The only exchange step is the gather of the per-rank
occupancy slabs to the rank that runs marching cubes (RCCL over xGMI when the backend is "nccl"; the same code runs
on gloo/CPU tensors in the tests).  The reference has no distributed code; this is new.
"""
import torch
import torch.distributed as dist


def slab_range(resolution, rank, world):
    """[i0, i1) of rank `rank`: contiguous, ordered, sizes differ by at most one."""
    base, rem = divmod(resolution, world)
    i0 = rank * base + min(rank, rem)
    return i0, i0 + base + (1 if rank < rem else 0)


def gather_slabs(local, resolution, dst=0, group=None):
    """local: [n_rank, R, R] tensor of this rank's slab.  Returns the full [R, R, R] tensor on `dst`, None elsewhere."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [slab_range(resolution, r, world) for r in range(world)]
    nmax = max(b - a for a, b in sizes)
    plane = local.shape[1:]
    if local.shape[0] == nmax:
        send = local.contiguous()
    else:  # pad to the common size (collectives want equal shapes)
        send = torch.zeros((nmax,) + tuple(plane), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    if rank == dst:
        bufs = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, bufs, dst=dst, group=group)
        return torch.cat([bufs[r][: sizes[r][1] - sizes[r][0]] for r in range(world)], 0)
    dist.gather(send, None, dst=dst, group=group)
    return None


def reconstruction_sharded(opt, net, calib_tensor, resolution, b_min, b_max, transform=None, dst=0, want_mesh=True):
    """Each rank sweeps its slab of the grid, slabs are gathered on `dst`, which extracts both meshes.
    Returns the 8-tuple of mesh_util.reconstruction on `dst`, None on the other ranks."""
    from . import mesh_util
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    i0, i1 = slab_range(resolution, rank, world)
    vh, vl, mat = mesh_util.eval_volumes(opt, net, calib_tensor, resolution, b_min, b_max, transform, i0, i1)
    full_hr = gather_slabs(vh, resolution, dst)
    full_lr = gather_slabs(vl, resolution, dst)
    if rank != dst:
        return None
    if not want_mesh:
        return full_hr, full_lr, mat
    return mesh_util.meshes_from_volumes(net, [full_hr, full_lr], mat)
