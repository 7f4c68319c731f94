"""CPU-only guard on a code-generation property the streaming kernels depend on: a guarded load
(`if (ok) v = *p`) compiles to a branch whose join waits for vmcnt(0), so guarded loads written back to back run one memory
round trip after the other (k_confine_v4 had 20 such drains between its 30 loads, k_project_v4 16, k_pcg_apply 14). The
kernels below were rewritten to load unconditionally and select the value; this test cross-compiles them for gfx950 (hipcc,
no GPU) and counts, in the ISA, the full drains that are followed by further loads (tools/isa_loads.py)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fluidnet_amd", "csrc")

# kernel (demangled prefix) -> (source file, most full drains allowed before its last load; the value before round 4's rewrite)
CASES = {
    "k_project_v4<true, false>": ("model.hip", 0, 18),
    "k_project_v4<true, true>": ("model.hip", 0, 18),             # (round 6: on the flags' wall codes, tfl_wall_plan)
    "k_bcs_div_stats_code<true>": ("model.hip", 0, 17),
    "k_bcs_div_stats_v4<true, false>": ("model.hip", 0, 17),     # (<.., true> = the opt-in ticket tail, model.hip)
    "k_curl_v4<true>": ("vorticity.hip", 0, 11),
    "k_pcg_apply<true>": ("pcg.hip", 0, 14),
    "k_jacobi<true, false>": ("jacobi.hip", 0, 1),
    "k_jacobi<false, false>": ("jacobi.hip", 0, 1),
    "k_conv2_mfma<16, false, false>": ("conv2d_mfma.hip", 0, 8),
    "k_minmax3_v4<false>": ("advect.hip", 0, 24),
    "k_velocity_update<true>": ("stencil.hip", 0, 5),
}


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_streaming_kernels_issue_their_loads_in_one_batch():
    by_file = {}
    for k, (f, _, _) in CASES.items():
        by_file.setdefault(f, []).append(k)
    found = {}
    for f, kernels in by_file.items():
        pat = "|".join("^" + re.escape(k) + "$" for k in kernels)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_loads.py"), pat, os.path.join(CSRC, f)],
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        for line in out.stdout.splitlines():
            m = re.match(r"^(\S.*?)\s+(\d+) loads,\s+(\d+) full drains", line)
            if m:
                found[m.group(1).strip()] = (int(m.group(2)), int(m.group(3)))
    for k, (f, allowed, before) in CASES.items():
        assert k in found, (k, sorted(found))
        loads, drains = found[k]
        assert loads >= 4 and drains <= allowed, "%s (%s): %d full drains between its %d loads (%d before the rewrite)" % (k, f, drains, loads, before)
