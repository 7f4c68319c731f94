"""CPU-side checks of bench.py's bookkeeping: the git blob hash bench.py uses to decide whether a stored PMC traffic
figure still describes a kernel, and the kernel -> source file map behind it."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_blob_sha_is_git_hash_object():
    from fluidnet_amd import _kernels
    p = os.path.join(_kernels.CSRC, "advect_vel3.hip")
    try:
        want = subprocess.check_output(["git", "hash-object", p], cwd=ROOT).decode().strip()
    except Exception:      # noqa: BLE001  (no git in the environment)
        import pytest
        pytest.skip("git not available")
    assert _kernels.blob_sha(p) == want


def test_every_profiled_kernel_maps_to_existing_sources():
    from fluidnet_amd import _kernels
    for k in _kernels.KERNEL_SOURCE:
        for path in ("mfma16", "winograd", "mfma", "direct"):
            f, sha = _kernels.source_sha(k, path)
            assert f and sha and len(sha) == 40, (k, path, f, sha)
            for n in f.split("+"):
                assert os.path.exists(os.path.join(_kernels.CSRC, n)), (k, n)


def test_pmc_traffic_json_carries_source_hashes():
    """profiles/pmc_traffic.json (what bench.py reads for roofline.traffic) records, per kernel, the source file(s) and their
    hash at measurement time; bench.py refuses a figure whose kernel has changed since."""
    tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    meta = tj["_meta"]
    assert meta["cells"] == 128 ** 3 and meta["commit"]
    for k, v in tj.items():
        if k.startswith("_"):
            continue
        assert v > 0 and k in meta["source_sha"], k


def test_bench_gpus_n_without_a_launcher_refuses_cleanly_without_gpus():
    """`python bench.py --gpus 2` started with no launcher around it becomes its own launcher; on a box
    without enough GPUs it must end with rc != 0 AND one JSON error line, not a traceback."""
    import sys
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box could really run it")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TFL_DIST_BACKEND")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] is None and "GPUs" in line["error"]


def test_bench_refuses_a_launcher_with_the_wrong_world_size():
    import sys
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["value"] is None and "WORLD_SIZE=3" in line["error"]


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_gpus_2_launches_itself_over_gloo_on_one_gpu():
    """`python bench.py --gpus 2` with NO launcher spawns its own two ranks; under TFL_DIST_BACKEND=gloo
    they share this box's GPU (host-staged messages: the control-flow check) and rank 0 prints ONE line with n_gpus = 2."""
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["TFL_DIST_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--preroll", "2",
                        "--blocks", "1", "--res", "64", "--no-config5", "--no-configs"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "gloo" in line["config"]["decomposition"]


@pytest.mark.gpu
@pytest.mark.parametrize("graph", ["0", "1"], ids=["eager-rank-step", "recorded-rank-step"])
def test_bench_gpus_2_over_real_rccl_with_ranks_sharing_the_gpu(graph):
    """The whole `--gpus N` path against the REAL RCCL on a one-GPU box (round 6): TFL_RANKS_SHARE_GPU=1 puts both ranks on device
    0, each under its own NCCL_HOSTID (RCCL's duplicate-GPU test passes, its socket transport carries the messages). The line
    must name the library's native transport (comm_rccl.cpp: the trial step on every rank succeeded) -- and, with
    TFL_SLAB_GRAPH=1, a rank-step recorded with the ncclSend / ncclRecv / ncclAllReduce inside the graph. A box with two GPUs
    runs the same thing one rank per GPU."""
    import sys
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TFL_DIST_BACKEND")}
    if torch.cuda.device_count() < 2:
        env["TFL_RANKS_SHARE_GPU"] = "1"
    env["TFL_SLAB_GRAPH"] = graph
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--preroll", "2",
                        "--blocks", "1", "--res", "64", "--no-config5", "--no-configs"], env=env, capture_output=True, text=True, timeout=360)
    if p.returncode != 0 and any(m in p.stderr for m in ("Bootstrap : no socket interface", "no socket interface found", "unhandled system error", "Duplicate GPU detected")):
        pytest.skip("RCCL cannot bootstrap between processes sharing this GPU in this environment (no usable network interface, or an RCCL that ignores NCCL_HOSTID)")
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["range_errors"] == 0
    assert "native RCCL send/recv" in line["config"]["decomposition"], line["config"]["decomposition"]
    issued = line["config"]["rank_step"]["issued_as"]
    assert ("HIP graph" in issued) == (graph == "1"), issued
