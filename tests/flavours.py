"""The two flavours of the HIP library (fluidnet_amd/csrc/Makefile): the product library libtfluids_hip.so carries one kernel
per job; libtfluids_hip_exp.so (-DTFL_EXPERIMENTS, `make exp`) also carries the earlier and the measured-slower kernel forms
and reads the switches that select them. A test that forces such a form runs in a CHILD process against the second one
(TFL_LIBRARY, fluidnet_amd/_lib.py): the library is chosen when it is loaded, once per process."""
import functools
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP_LIB = os.path.join(ROOT, "fluidnet_amd", "libtfluids_hip_exp.so")

# switches only the EXPERIMENTS flavour reads (tfl_host.hpp exp_env, conv_mfma16_exp.inc, advect_scalar3_march.inc, ...)
EXPERIMENT_SWITCHES = ("TFL_ADVECT_GATHER", "TFL_SCALAR_GATHER", "TFL_M16_", "TFL_NO_VEC4", "TFL_SCAL3_MARCH", "TFL_SCAL3M_CZ_",
                       "TFL_STATS_FOLD", "TFL_VEL3_KZ_B", "TFL_VORT_CZ", "TFL_VORT_TILE", "TFL_WF_", "TFL_XCD_", "TFL_SLAB_WIDEN", "TFL_CONV_DEBUG",
                       "TFL_CONV_TRACE")


def is_experiments_process():
    return os.path.basename(os.environ.get("TFL_LIBRARY", "")) == os.path.basename(EXP_LIB)


def child_env(env, extra=None):
    """`env` + `extra`; when `extra` holds a switch of the EXPERIMENTS flavour the child loads that library"""
    e = dict(env)
    e.update(extra or {})
    if any(k.startswith(EXPERIMENT_SWITCHES) for k in (extra or {})):
        if not os.path.exists(EXP_LIB):
            import pytest
            pytest.skip("fluidnet_amd/libtfluids_hip_exp.so is not built (make -C fluidnet_amd/csrc exp)")
        e["TFL_LIBRARY"] = EXP_LIB
    return e


def experiments_flavour(fn):
    """Decorator: the test body switches kernel forms inside one process (monkeypatch.setenv + a new model), so the whole test
    runs in a child pytest process that loads the EXPERIMENTS flavour; the parent only checks that the child passed."""
    @functools.wraps(fn)
    def wrapper(*a, **kw):
        if is_experiments_process():
            return fn(*a, **kw)
        import pytest
        if not os.path.exists(EXP_LIB):
            pytest.skip("fluidnet_amd/libtfluids_hip_exp.so is not built (make -C fluidnet_amd/csrc exp)")
        nodeid = os.environ["PYTEST_CURRENT_TEST"].rsplit(" ", 1)[0]
        env = dict(os.environ, TFL_LIBRARY=EXP_LIB)
        out = subprocess.run([sys.executable, "-m", "pytest", nodeid, "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                             capture_output=True, text=True, timeout=1800)
        assert out.returncode == 0 and " passed" in out.stdout and "failed" not in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    return wrapper
