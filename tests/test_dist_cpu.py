"""World-size-2 (and 3) gloo runs of the slab sharding on CPU tensors: the N>1 path of bench.py / dist.py."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from surs_amd import dist as sdist


def test_slab_ranges_partition():
    for R in (1, 7, 24, 128, 512):
        for world in (1, 2, 3, 8):
            rs = [sdist.slab_range(R, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == R
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, R, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = torch.arange(R * R * R, dtype=torch.float32).reshape(R, R, R)
    i0, i1 = sdist.slab_range(R, rank, world)
    got = sdist.gather_slabs(full[i0:i1].clone(), R, dst=0)
    ok = True
    if rank == 0:
        ok = bool(torch.equal(got, full))
    else:
        ok = got is None
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and t.item() == world
    out[rank] = ok
    dist.destroy_process_group()


def _run(world, R):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), R, out), nprocs=world, join=True)
    assert all(out[r] for r in range(world)), dict(out)


def test_gather_slabs_world2_even():
    _run(2, 8)


def test_gather_slabs_world3_ragged():
    _run(3, 7)


# ------------------------------------------------------------------ slab-mode mesh assembly (dist.assemble_slab_meshes)
# The per-slab extraction is a GPU kernel; here every rank's slab result is DERIVED from the oracle's mesh of the whole
# volume exactly as surs_mc_lewiner_range_slab defines it (a slab owns the vertices / faces its cell layers create, numbers
# them from 0 and refers to the slab below as -(2 + slot)), and the protocol - counts all_gather, boundary-id exchange,
# renumbering, mesh gather - must give the whole mesh back.  tests/test_gpu_dist.py runs the real kernels.

def _slab_fixture(R=20, world=3):
    import mc_volumes
    import oracle
    # a smooth blob that reaches into every slab, plus a little seeded noise (centre vertices, ambiguous cells)
    z, y, x = np.mgrid[:R, :R, :R].astype(np.float64)
    c = (R - 1) / 2.0
    vol = 1.0 / (1.0 + np.exp(((x - c) ** 2 / 30.0 + (y - c - 0.7) ** 2 / 50.0 + (z - c + 0.3) ** 2 / 70.0) - 1.0))
    vol = (vol + 0.08 * (mc_volumes.noise((R, R, R), 5) - 0.5)).astype(np.float32)
    V, F, _, _ = oracle.marching_cubes_lewiner(vol.astype(np.float64), 0.5)

    def prefix(i1):   # (vertices, faces) created by the cell layers [0, i1)
        if i1 < 1:
            return 0, 0
        try:
            v, f, _, _ = oracle.marching_cubes_lewiner(vol[:i1 + 1].astype(np.float64), 0.5)
        except (ValueError, RuntimeError):
            return 0, 0
        assert np.array_equal(v, V[:len(v)]) and np.array_equal(f, F[:len(f)])   # a prefix of the whole mesh
        return len(v), len(f)

    slabs = []
    for r in range(world):
        i0, i1 = sdist.slab_range(R, r, world)
        top = i1 if r < world - 1 else R - 1
        (v0, f0), (v1, f1) = prefix(i0), prefix(top)
        faces = F[f0:f1].astype(np.int64).copy()
        ref = faces < v0
        pos = V[faces[ref]]                       # vertices of the slab below: on plane i0, on an x- or a y-edge
        assert np.all(pos[:, 0] == i0)
        on_x = pos[:, 2] != np.floor(pos[:, 2])
        slot = np.where(on_x, np.floor(pos[:, 1]) * R + np.floor(pos[:, 2]), R * R + np.floor(pos[:, 1]) * R + np.floor(pos[:, 2]))
        assert np.all(on_x ^ (pos[:, 1] != np.floor(pos[:, 1])))
        faces[ref] = -(2 + slot.astype(np.int64))
        faces[~ref] -= v0
        ids = np.full((2, R, R), -7, np.int32)    # entries of edges the surface does not cross are undefined
        own = V[v0:v1]
        for k in np.nonzero(own[:, 0] == i1)[0]:
            p = own[k]
            if p[2] != np.floor(p[2]):
                ids[0, int(p[1]), int(np.floor(p[2]))] = k
            elif p[1] != np.floor(p[1]):
                ids[1, int(np.floor(p[1])), int(p[2])] = k
        mm = vol[i0:top + 1]
        slabs.append(dict(verts=own.astype(np.float64), faces=faces.astype(np.int32), ids=ids,
                          counts=(v1 - v0, f1 - f0, float(mm.min()), float(mm.max()))))
    return V, F, slabs


def _assemble_worker(rank, world, port, R, slabs, out, sabotage=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if sabotage is not None:
        # one rank cannot create (dst: a full /dev/shm) or map (another container's /dev/shm) the shared block: the ranks must
        # agree on the point-to-point delivery instead of hanging in a barrier
        real = sdist.SharedMeshStore.open.__func__
        import warnings
        warnings.simplefilter("ignore")

        def broken(cls, owner_pid, owner_token, tag, nbytes, create):
            if (sabotage == "create" and create) or (sabotage == "map" and rank == 1 and not create):
                raise OSError(28, "No space left on device")
            return real(cls, owner_pid, owner_token, tag, nbytes, create)

        sdist.SharedMeshStore.open = classmethod(broken)
    s = slabs[rank]
    res = [(torch.from_numpy(s["verts"].copy()), torch.from_numpy(s["faces"].copy())) for _ in range(2)]   # two fields, same data

    def fixup(f, faces, own_off, below, below_off):
        a = faces.numpy().reshape(-1)
        neg = a < 0
        if neg.any():
            a[neg] = below.numpy().reshape(-1)[-a[neg] - 2] + below_off
        a[~neg] += own_off

    got = sdist.assemble_slab_meshes(res, [s["counts"], s["counts"]], lambda f: torch.from_numpy(s["ids"].copy()), fixup, R,
                                     torch.device("cpu"), dst=0)
    if got is not None and sabotage is None and os.environ.get("SURS_SLAB_P2P", "0") != "1":
        # delivered through the shared block: owning copies by default (a second delivery must not change them)
        keep = [(v.clone(), f.clone()) for v, f in got]
        for blk in sdist.SharedMeshStore._maps.values():
            blk[0].zero_()
        assert all(torch.equal(a, c) and torch.equal(b, d) for (a, b), (c, d) in zip(got, keep))
        assert not any(n.startswith("surs_mesh_%d_" % os.getpid()) and oct(os.stat("/dev/shm/" + n).st_mode)[-3:] != "600"
                       for n in os.listdir("/dev/shm"))
    out[rank] = None if got is None else [(v.numpy(), f.numpy()) for v, f in got]
    dist.destroy_process_group()


def _run_assemble(world, R=20, sabotage=None):
    V, F, slabs = _slab_fixture(R, world)
    assert len(V) > 100 and sum(s["counts"][0] for s in slabs) == len(V)
    assert all((s["faces"] < 0).any() for s in slabs[1:])       # every upper slab refers to the slab below
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_assemble_worker, args=(world, _free_port(), R, slabs, out, sabotage), nprocs=world, join=True)
    assert all(out[r] is None for r in range(1, world))
    for v, f in out[0]:
        assert np.array_equal(v, V.astype(np.float64)) and np.array_equal(f, F)


def test_slab_mesh_assembly_world2():
    _run_assemble(2)


def test_slab_mesh_assembly_point_to_point(monkeypatch):
    """The cross-node delivery (meshes sent to dst point to point) instead of the shared-memory blocks of one node."""
    monkeypatch.setenv("SURS_SLAB_P2P", "1")
    _run_assemble(3)


def test_shared_delivery_falls_back_when_dst_cannot_reserve_the_block():
    _run_assemble(3, sabotage="create")


def test_shared_delivery_falls_back_when_one_rank_cannot_map_the_block():
    _run_assemble(3, sabotage="map")


def test_shared_store_refuses_symlinks_and_foreign_files(tmp_path):
    """The block's name is not guessable (creator pid + a random 48-bit token) and is created O_EXCL | O_NOFOLLOW: a planted file
    or link of that name makes open() raise (-> the agreed fallback), it is never followed or reused."""
    import pytest
    tok = sdist.SharedMeshStore.token()
    assert 0 < tok < 2 ** 48 and sdist.SharedMeshStore.token() == tok
    path = sdist.SharedMeshStore._path(os.getpid(), tok, "plant")
    target = tmp_path / "victim"
    target.write_bytes(b"x" * 64)
    os.symlink(str(target), path)
    try:
        with pytest.raises(OSError):
            sdist.SharedMeshStore.open(os.getpid(), tok, "plant", 1024, True)
        with pytest.raises(OSError):
            sdist.SharedMeshStore.open(os.getpid(), tok, "plant", 16, False)
        assert target.read_bytes() == b"x" * 64
    finally:
        os.unlink(path)
    buf = sdist.SharedMeshStore.open(os.getpid(), tok, "own", 4096, True)
    assert buf.numel() >= 4096 and os.stat(sdist.SharedMeshStore._path(os.getpid(), tok, "own")).st_blocks * 512 >= 4096   # reserved
    big = sdist.SharedMeshStore.open(os.getpid(), tok, "own", 3 << 20, True)                                            # grows in place
    assert big.numel() >= 3 << 20
    sdist.SharedMeshStore.release_all()
    os.unlink(sdist.SharedMeshStore._path(os.getpid(), tok, "own"))


def _failing_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    empty = [(torch.empty((0, 3), dtype=torch.float64), torch.empty((0, 3), dtype=torch.int32))] * 2
    nan = float("nan")
    try:
        sdist.assemble_slab_meshes(empty, [(0, 0, nan, nan)] * 2 if rank == 1 else [(5, 7, 0.0, 1.0)] * 2, None, None, 8, torch.device("cpu"),
                                   status=2 if rank == 1 else 0, failure=MemoryError("rank 1 ran out of memory") if rank == 1 else None)
        out[rank] = "no error"
    except MemoryError as e:
        out[rank] = "MemoryError"
    except RuntimeError as e:
        out[rank] = "RuntimeError: " + str(e)
    dist.destroy_process_group()


def test_failure_on_one_rank_raises_on_every_rank():
    """A rank that fails in its sweep carries a status flag into the counts exchange: it re-raises its own exception, the others
    raise too instead of waiting in the next collective."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_failing_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    assert out[1] == "MemoryError" and out[0].startswith("RuntimeError") and "rank(s) [1]" in out[0] and out[2] == out[0], dict(out)


def test_slab_mesh_assembly_world3_ragged():
    _run_assemble(3)


def test_offsets_and_small_grids():
    assert list(sdist.offsets_from_counts([3, 0, 5])) == [0, 3, 3]
