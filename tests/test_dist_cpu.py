"""Multi-process CPU tests (gloo, world_size 2 and 3) of the N > 1 host path: slab layout, halo
exchange and the statistics all-reduce of fluidnet_amd.dist. The kernels themselves need a GPU; what
is verified here is that every rank ends up holding exactly the planes of the global field it should."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from fluidnet_amd.dist import DistComm, SlabLayout, run_lockstep


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _global_fields(z_total):
    g = torch.Generator().manual_seed(7)
    return [torch.rand(1, 3, z_total, 6, 5, generator=g), torch.rand(1, 1, z_total, 6, 5, generator=g)]


def _worker(rank, world, port, z_total, halo, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lay = SlabLayout(z_total, world, rank, halo)
        glob = _global_fields(z_total)
        local = [lay.extract(t) for t in glob]
        for t in local:   # poison the halos: the exchange must restore them from the neighbours
            if lay.has_lower:
                t[:, :, :lay.c0] = -1.0
            if lay.has_upper:
                t[:, :, lay.c1:] = -2.0
        comm = DistComm()
        comm.exchange(lay, local)
        ok = all(torch.equal(l, lay.extract(g)) for l, g in zip(local, glob))
        stats = torch.tensor([[float(rank + 1), 2.0 * (rank + 1)]], dtype=torch.float64)
        comm.allreduce_sum(stats)
        tot = world * (world + 1) / 2
        ok = ok and stats.tolist() == [[tot, 2 * tot]]
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,z_total,halo", [(2, 16, 3), (3, 24, 4), (2, 8, 4)])
def test_halo_exchange_and_allreduce_gloo(world, z_total, halo):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), z_total, halo, out), nprocs=world, join=True)
    assert dict(out) == {r: True for r in range(world)}


def test_layout_partition_covers_grid():
    for world, z in [(1, 17), (2, 32), (4, 64), (8, 256)]:
        lays = [SlabLayout(z, world, r, 10 if world > 1 else 0) for r in range(world)]
        assert lays[0].z0 == 0 and lays[-1].z1 == z
        for a, b in zip(lays, lays[1:]):
            assert a.z1 == b.z0
        for l in lays:
            assert l.lo == max(l.z0 - l.halo, 0) and l.hi == min(l.z1 + l.halo, z)
            assert l.c1 - l.c0 == z // world and (l.has_lower, l.has_upper) == (l.rank > 0, l.rank < world - 1)
    with pytest.raises(ValueError):
        SlabLayout(30, 4, 0)
    with pytest.raises(ValueError):
        SlabLayout(32, 4, 0, halo=10)


def test_lockstep_virtual_ranks_match_dist_semantics():
    """run_lockstep (in-process virtual ranks, used by the GPU equivalence test) moves the same planes."""
    z_total, world, halo = 24, 3, 4
    glob = _global_fields(z_total)
    lays = [SlabLayout(z_total, world, r, halo) for r in range(world)]
    locs = []
    for lay in lays:
        l = [lay.extract(t) for t in glob]
        for t in l:
            t[:, :, :lay.c0] = -1.0
            t[:, :, lay.c1:] = -2.0
        locs.append(l)
    stats = [torch.tensor([[1.0 + r, 3.0]], dtype=torch.float64) for r in range(world)]

    def gen(r):
        yield ("halo", locs[r])
        yield ("allreduce", stats[r])

    run_lockstep([(gen(r), lays[r]) for r in range(world)])
    for lay, l in zip(lays, locs):
        assert all(torch.equal(a, lay.extract(g)) for a, g in zip(l, glob))
    assert all(s.tolist() == [[6.0, 9.0]] for s in stats)
