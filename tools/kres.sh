#!/bin/bash
# CPU-only: registers / spills / LDS / occupancy of every kernel in one csrc file (hipcc -Rpass-analysis=kernel-resource-usage)
# usage: tools/kres.sh <file.hip> [grep pattern] [extra hipcc flags]
f=$1; pat=${2:-.}; shift; shift
cd "$(dirname "$0")/../fluidnet_amd/csrc"
extra=""; case $f in advect.hip|advect_vel3.hip|advect_scalar3.hip) extra="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function $extra -I../../include "$@" \
  -Rpass-analysis=kernel-resource-usage -x hip -c -o /dev/null "$f" 2>&1 |
  python3 -c '
import re, sys, subprocess
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
for r in rows:
    try: name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0]
    except Exception: name = r["name"]
    name = name.replace("tfl::", "").replace("void ", "")
    print("%-60s sgpr %3s vgpr %3s agpr %3s occ %s sspill %3s vspill %3s lds %s" % (name[:60], r.get("TotalSGPRs"), r.get("VGPRs"), r.get("AGPRs"), r.get("Occupancy [waves/SIMD]"), r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("LDS Size [bytes/block]")))
' | grep -E "$pat"
