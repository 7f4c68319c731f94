"""The library's own z-slab transport (fluidnet_amd/csrc/comm_rccl.cpp: ncclSend / ncclRecv / ncclAllReduce bound at
run time). RCCL refuses several ranks on one device, and the GPU box has one: the full path -- native slab step ->
tfl_comm callbacks -> nccl* entry points -> stream-ordered transfers -- is exercised with tests/stub_rccl.cpp (an
in-process RCCL stand-in with RCCL's stream semantics) between virtual ranks; the real librccl.so is bound, initialised
and used for a world of one."""
import ctypes
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def stub_so(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("stub") / "libstub_rccl.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O2", "-o", out, os.path.join(HERE, "stub_rccl.cpp")])
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("world,overlap", [(2, 0), (3, 1)])
def test_native_transport_between_virtual_ranks(stub_so, world, overlap):
    env = dict(os.environ, TFL_RCCL_LIBRARY=stub_so)
    r = subprocess.run([sys.executable, os.path.join(HERE, "native_transport_run.py"), str(world), str(overlap)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "native transport ok" in r.stdout


@pytest.mark.gpu
def test_rank_step_graph_equals_eager_step(stub_so):
    """the rank-step recorded into a HIP graph with the transport's sends / receives / all-reduce inside
    (tfl_slab_graph_create) replays to the same bits as the eager step -- middle and end rank of a 4-rank layout, with and
    without the boundary-strip / interior split, through the native transport over the stub in STUB_RCCL_NULL mode."""
    env = dict(os.environ, TFL_RCCL_LIBRARY=stub_so, STUB_RCCL_NULL="1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "slab_graph_run.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "slab graph ok" in r.stdout


@pytest.mark.gpu
def test_real_rccl_world_of_one():
    """Binds the real RCCL (the copy PyTorch already loaded, else librccl.so.1), creates a communicator of one rank and
    drives the three callbacks the slab step uses: an empty neighbour exchange and a 4-double all-reduce."""
    import torch
    from fluidnet_amd import _lib, tfluids
    from fluidnet_amd.dist import RcclComm
    lib, ctx = tfluids._context(torch.zeros(1, device="cuda:0"))
    if lib.tfl_rccl_available(ctx) != 1:
        pytest.skip("no RCCL in this environment: " + lib.tfl_last_error(ctx).decode())
    assert "rccl" in lib.tfl_rccl_comm_origin(ctx).decode()
    try:
        comm = RcclComm(ctx, RcclComm.unique_id(ctx), 0, 1)
    except tfluids.TfluidsError as e:      # RCCL present but its bootstrap cannot run here (no usable network interface)
        pytest.skip("ncclCommInitRank failed in this environment: %s" % e)
    x = torch.tensor([1.5, -2.0, 3.25, 1e-30], dtype=torch.float64, device="cuda:0")
    st = comm.struct
    assert st.exchange_start(st.user, 2, None, 0, None, 0, None, 0, None, 0) == 0
    assert st.exchange_wait(st.user, 2) == 0
    assert st.allreduce_sum(st.user, ctypes.c_void_p(x.data_ptr()), 4) == 0
    torch.cuda.synchronize()
    assert x.tolist() == [1.5, -2.0, 3.25, 1e-30]
    comm.close()


def test_transport_symbols_are_exported():
    lib = ctypes.CDLL(os.path.join(ROOT, "fluidnet_amd", "libtfluids_hip.so"))
    for n in ("tfl_rccl_available", "tfl_rccl_comm_origin", "tfl_rccl_get_unique_id", "tfl_rccl_comm_create",
              "tfl_rccl_comm_wrap", "tfl_rccl_comm_callbacks", "tfl_rccl_comm_destroy", "tfl_rccl_comm_set_inline", "tfl_slab_graph_create",
              "tfl_slab_graph_step", "tfl_slab_graph_nodes", "tfl_slab_graph_destroy"):
        assert hasattr(lib, n), n
