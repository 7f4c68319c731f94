"""The native z-slab transport between REAL ranks over the REAL librccl: one process per rank, ncclCommInitRank through
tfl_rccl_comm_create, three slab steps in both message forms (in-place chunks; TFL_RCCL_PACKED=1 staged buffers), the
rank-step recorded into a HIP graph with the ncclSend / ncclRecv / ncclAllReduce inside, the exact reach mode; owned planes
against the single-GPU step.
With >= `world` GPUs in the box: one GPU per rank (the 8-GPU scaling bench uses exactly this path). On the 1-GPU boxes of this
pool (round 6): every rank on device 0, each process under its own NCCL_HOSTID -- RCCL's "Duplicate GPU detected" test compares
host hash AND bus id, so ranks that call themselves different hosts pass it and their messages take RCCL's socket transport
over the loopback interface. That is no measurement of anything, but every call the transport makes runs against the real
library with real peers (tools/rccl_one_gpu.sh is the same thing as a script; profiles/r06_rccl_one_gpu.txt its record)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _gpus():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:      # noqa: BLE001
        return 0


def one_gpu_rank_env(env, rank):
    """Environment of rank `rank` when all ranks share device 0 (see the module docstring)."""
    return dict(env, TFL_RCCL_ONE_GPU="1", NCCL_HOSTID="tflhost%d" % rank, NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1")


@pytest.mark.gpu
@pytest.mark.skipif(_gpus() < 1, reason="needs a GPU")
@pytest.mark.parametrize("world", [2, 4])
def test_slab_steps_over_real_rccl(tmp_path, world):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TFL_RCCL_LIBRARY", None)
    one_gpu = _gpus() < world
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "rccl_multiproc_run.py"), str(r), str(world), str(tmp_path)],
                              env=one_gpu_rank_env(env, r) if one_gpu else env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=240)[0])      # (a healthy run takes 10-20 s)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if any(p.returncode == 77 for p in procs):
        pytest.skip("RCCL could not initialise between processes here: " + " | ".join(o.strip().splitlines()[-1] for o in outs if o.strip()))
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("rccl multiproc ok rank %d" % r) in o, "rank %d:\n%s" % (r, o[-4000:])


def test_worker_imports():
    """CPU: the per-rank worker parses (its body needs GPUs)."""
    import ast
    ast.parse(open(os.path.join(HERE, "rccl_multiproc_run.py")).read())
