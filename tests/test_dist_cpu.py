"""Multi-process CPU tests (gloo, world_size 2 and 3) of the N > 1 host path of fluidnet_amd.dist: slab layout and the
transport callbacks tfl_simulate_step_slab drives (exchange_start / exchange_wait / allreduce_sum). The step itself needs
a GPU (tests/test_hip_simulate.py runs it on virtual ranks); what is verified here is the C-callback plumbing -- raw
(pointer, count) pairs into views of the workspace, neighbour pairing, tags in flight at the same time -- through the
very ctypes trampolines the library calls."""
import ctypes
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from fluidnet_amd.dist import DistComm, SlabLayout, ThreadComm, slab_halo


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _drive(comm, rank, world, ws):
    """What the native step does with a transport: two messages in flight at once (tags 0, 1) with different sizes per
    direction, a third one started and finished in between, one all-reduce. Returns True when every received buffer
    holds exactly what the neighbour sent."""
    base = ws.data_ptr()
    P = lambda off: ctypes.c_void_p(base + 4 * off)
    lower, upper = rank > 0, rank < world - 1
    n_lo_s, n_lo_r, n_hi_s, n_hi_r = 24, 40, 40, 24          # send_lo pairs with the lower rank's recv_hi, etc.
    ok = True

    def payload(src_rank, tag, direction, n):
        return torch.arange(n, dtype=torch.float32) + 1000.0 * src_rank + 100.0 * tag + (0.5 if direction == "hi" else 0.0)

    def fill(tag):
        o = 200 * tag
        ws[o:o + n_lo_s] = payload(rank, tag, "lo", n_lo_s)
        ws[o + 50:o + 50 + n_hi_s] = payload(rank, tag, "hi", n_hi_s)
        ws[o + 100:o + 100 + n_lo_r] = -1.0
        ws[o + 150:o + 150 + n_hi_r] = -1.0

    def start(tag):
        o = 200 * tag
        return comm.struct.exchange_start(None, tag, P(o) if lower else None, n_lo_s if lower else 0,
                                          P(o + 100) if lower else None, n_lo_r if lower else 0,
                                          P(o + 50) if upper else None, n_hi_s if upper else 0,
                                          P(o + 150) if upper else None, n_hi_r if upper else 0)

    def check(tag):
        o = 200 * tag
        good = True
        if lower:    # my recv_lo = the lower rank's send_hi
            good &= torch.equal(ws[o + 100:o + 100 + n_lo_r], payload(rank - 1, tag, "hi", n_hi_s))
        if upper:
            good &= torch.equal(ws[o + 150:o + 150 + n_hi_r], payload(rank + 1, tag, "lo", n_lo_s))
        return good

    for t in (0, 1, 2):
        fill(t)
    assert start(0) == 0 and start(1) == 0
    assert start(2) == 0 and comm.struct.exchange_wait(None, 2) == 0
    ok &= check(2)
    so = 800                                       # 8-byte aligned slot for two doubles
    ws[so:so + 4].view(torch.float64)[:] = torch.tensor([float(rank + 1), 2.0 * (rank + 1)], dtype=torch.float64)
    assert comm.struct.allreduce_sum(None, P(so), 2) == 0
    tot = world * (world + 1) / 2
    ok &= ws[so:so + 4].view(torch.float64).tolist() == [tot, 2 * tot]
    assert comm.struct.exchange_wait(None, 0) == 0 and comm.struct.exchange_wait(None, 1) == 0
    ok &= check(0) and check(1)
    return bool(ok)


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ws = torch.zeros(1024)
        comm = DistComm(rank, world)
        comm.bind(ws)
        out[rank] = _drive(comm, rank, world, ws)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_transport_callbacks_gloo(world):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {r: True for r in range(world)}


def _worker_absent_neighbour(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if rank == 0:
            ws = torch.zeros(1024)
            comm = DistComm(rank, world, timeout_s=1.5)
            comm.bind(ws)
            p = lambda off: ctypes.cast(ws.data_ptr() + 4 * off, ctypes.POINTER(ctypes.c_float))
            # rank 0 posts a send/recv with rank 1, which never takes part
            rc0 = comm.struct.exchange_start(None, 0, None, 0, None, 0, p(0), 8, p(8), 8)
            rc1 = comm.struct.exchange_wait(None, 0)
            out[0] = (rc0, rc1, type(comm.error).__name__)
        else:
            import time
            time.sleep(4.0)
            out[1] = "idle"
    finally:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


def test_absent_neighbour_times_out_instead_of_hanging():
    """a halo partner that never arrives must surface as an error. With gloo the wait gives up
    after DistComm.timeout_s; the callback returns non-zero (the native step reports TFL_ECOMM) and keeps the exception."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_absent_neighbour, args=(2, _free_port(), out), nprocs=2, join=True)
    rc0, rc1, err = out[0]
    assert rc0 == 0 and rc1 != 0 and err == "TimeoutError", out[0]


def test_transport_callbacks_threads():
    """ThreadComm (virtual ranks of the single-GPU decomposition tests) obeys the same contract."""
    import threading
    world = 3
    hub = ThreadComm.Hub(world)
    res = {}

    def work(r):
        ws = torch.zeros(1024)
        comm = ThreadComm(hub, r)
        comm.bind(ws)
        res[r] = _drive(comm, r, world, ws)

    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert res == {r: True for r in range(world)}


def test_callback_exceptions_become_return_codes():
    comm = DistComm.__new__(DistComm)
    from fluidnet_amd.dist import _CommBase
    _CommBase.__init__(comm)
    comm.bind(torch.zeros(16))
    comm.wait = lambda tag: (_ for _ in ()).throw(RuntimeError("link down"))
    assert comm.struct.exchange_wait(None, 3) == 1 and isinstance(comm.error, RuntimeError)


def test_layout_partition_covers_grid():
    assert (slab_halo(1), slab_halo(2), slab_halo(3)) == (4, 5, 7)
    for world, z in [(1, 17), (2, 32), (4, 64), (8, 256), (8, 128)]:
        lays = [SlabLayout(z, world, r) for r in range(world)]
        assert lays[0].z0 == 0 and lays[-1].z1 == z
        for a, b in zip(lays, lays[1:]):
            assert a.z1 == b.z0
        for l in lays:
            assert l.halo == (4 if world > 1 else 0)
            assert l.lo == max(l.z0 - l.halo, 0) and l.hi == min(l.z1 + l.halo, z)
            assert l.c1 - l.c0 == z // world and (l.has_lower, l.has_upper) == (l.rank > 0, l.rank < world - 1)
    t = torch.arange(2 * 3 * 16 * 2 * 2, dtype=torch.float32).view(2, 3, 16, 2, 2)
    lay = SlabLayout(16, 2, 1)
    loc = lay.extract(t)
    assert loc.is_contiguous() and torch.equal(lay.owned(loc), t[:, :, 8:16]) and loc.data_ptr() != t.data_ptr()
    with pytest.raises(ValueError):
        SlabLayout(30, 4, 0)
    with pytest.raises(ValueError):
        SlabLayout(24, 8, 0)          # 3-plane slabs cannot hold a 4-plane halo
