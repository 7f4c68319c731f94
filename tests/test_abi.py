"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports exactly the symbols include/tfluids_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tfluids_hip.h")


@pytest.fixture(scope="module")
def lib_path():
    import __graft_entry__ as g
    g.build()
    from fluidnet_amd import _lib
    assert os.path.exists(_lib.LIB_PATH)
    return _lib.LIB_PATH


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tfl_[A-Za-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib_path):
    names = _declared()
    assert len(names) >= 18
    lib = ctypes.CDLL(lib_path)
    for n in names:
        assert hasattr(lib, n), "library does not export " + n
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib_path]).decode()
    exported = sorted(set(re.findall(r" T (tfl_[A-Za-z0-9_]+)$", out, flags=re.M)))
    assert exported == names, (set(exported) ^ set(names))


def test_binding_covers_header(lib_path):
    from fluidnet_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    assert lib.tfl_abi_version() == 4


def test_library_has_gfx950_code_object(lib_path):
    blob = open(lib_path, "rb").read()
    assert b"gfx950" in blob


def test_cpu_tensors_are_refused():
    """No CPU fallback: the product path must fail loudly without a GPU."""
    import torch
    from fluidnet_amd import tfluids, TfluidsError
    with pytest.raises(TfluidsError):
        tfluids.setWallBcsForward(torch.zeros(1, 2, 1, 8, 8), torch.ones(1, 1, 1, 8, 8))


def _build_c_example(tmp_path):
    exe = str(tmp_path / "step_from_c")
    subprocess.check_call(["gcc", "-std=c99", "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "examples", "step_from_c.c"),
                           "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                           "-L" + os.path.join(ROOT, "fluidnet_amd"), "-ltfluids_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "fluidnet_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_plain_c_host_links_against_the_abi(lib_path, tmp_path):
    """A C99 translation unit that includes only tfluids_hip.h compiles (gcc, not hipcc) and links: the boundary
    has no C++/torch types in it (what a cgo / JNI / LuaJIT-FFI binding relies on)."""
    assert os.path.exists(_build_c_example(tmp_path))


@pytest.mark.gpu
def test_plain_c_host_runs(lib_path, tmp_path):
    out = subprocess.run([_build_c_example(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr
