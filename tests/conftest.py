import os
import sys

# The checkers use two OpenMP runtimes in one process (libgomp in oracle/*.so, torch's own for the CPU
# convolutions). On a many-core host their spinning worker pools fight each other (a 10 s suite took 110 s
# on the 256-core GPU box); passive waiting keeps the test suite fast. Set before either is loaded.
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")
os.environ.setdefault("KMP_BLOCKTIME", "0")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "tests"), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests are skipped (not failed) on a box without an MI355X or without the built library."""
    import torch
    lib_ok = os.path.exists(os.path.join(ROOT, "fluidnet_amd", "libtfluids_hip.so"))
    if torch.cuda.is_available() and lib_ok:
        return
    why = "no GPU visible" if lib_ok else "fluidnet_amd/libtfluids_hip.so is not built"
    skip = pytest.mark.skip(reason="gpu test: " + why)
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The plain-C restatement (oracle/tfluids_oracle.c), built on demand."""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    from oracle.oracle import OracleTfluids
    return OracleTfluids()


@pytest.fixture(scope="session")
def ref():
    """The reference's own CPU sources compiled here (oracle/_ref); skipped when neither the
    prebuilt .so nor /root/reference is present (e.g. on a box that received no oracle/_ref)."""
    import subprocess
    from oracle import ref as refmod
    if not refmod.available() and os.path.isdir("/root/reference/torch/tfluids"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    if not refmod.available():
        pytest.skip("oracle/_ref/libtfluids_ref.so not built and /root/reference absent")
    return refmod.RefTfluids()


@pytest.fixture(scope="session")
def ref_pcg(ref):
    """The reference's CUDA-only PCG host function (generic/tfluids.cu:864-1759) compiled for the host over the
    cuSPARSE / cuBLAS stand-ins of oracle/ref_shim/cusparse_host.h (`make ref_pcg`); skipped like `ref`."""
    import subprocess
    from oracle import ref as refmod
    if not refmod.pcg_available() and os.path.isdir("/root/reference/torch/tfluids"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref_pcg"])
    if not refmod.pcg_available():
        pytest.skip("oracle/_ref/libtfluids_ref_pcg.so not built and /root/reference absent")
    return ref
