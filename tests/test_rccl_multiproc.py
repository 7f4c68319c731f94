"""The native z-slab transport between REAL ranks: one process per GPU, ncclCommInitRank through tfl_rccl_comm_create,
three slab steps in both message forms (in-place chunks; TFL_RCCL_PACKED=1 staged buffers), owned planes against the
single-GPU step. Needs two GPUs in the box: skipped on the 1-GPU boxes of this pool, and the first thing that runs when a
multi-GPU node shows up (the 8-GPU scaling bench uses exactly this path)."""
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


@pytest.mark.gpu
@pytest.mark.skipif(_gpus() < 2, reason="needs >= 2 GPUs in one box (RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("world", [2, 4])
def test_slab_steps_over_real_rccl(tmp_path, world):
    if _gpus() < world:
        pytest.skip("needs %d GPUs" % world)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TFL_RCCL_LIBRARY", None)
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "rccl_multiproc_run.py"), str(r), str(world), str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600)[0])
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
