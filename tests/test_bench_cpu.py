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
