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
