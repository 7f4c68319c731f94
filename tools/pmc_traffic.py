"""Per-kernel HBM-side traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (counters in KiB; FETCH_SIZE reports half of the bytes
of wide coalesced reads -> doubled). Writes profiles/<tag>_pmc_traffic.txt and profiles/pmc_traffic.json (read by
bench.py for roofline.traffic).
usage: python tools/pmc_traffic.py <fetch counter csv> <write counter csv> <tag> [cells]"""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fluidnet_amd"))
import _kernels  # noqa: E402  (fluidnet_amd/_kernels.py, without importing the package and torch)
ALG = {"k_stream_copy": 8, "k_conv3_mfma": 64, "k_conv3_mfma_tail": 36, "k_conv3_mfma_in": 44, "k_conv3_mid": 64, "k_conv3_tail": 36, "k_conv3_in": 44, "k_vel_bwd": 40, "k_vel_fwd": 28,
       "k_scalar_fwd": 24, "k_scalar_bwd": 28, "k_confine": 44, "k_curl": 28, "k_vort_fused": 28, "k_bcs_div_stats": 30, "k_minmax3": 16,
       "k_project": 60, "k_add_buoyancy": 32}


def short(name):
    m = re.search(r"::(k_\w+)(<[^>]*>)?", name)     # tfl::k_x<...> or tfl::(anonymous namespace)::k_x
    if not m:
        return None
    k, targs = m.group(1), m.group(2) or ""
    k = {"k_vel3_fwd": "k_vel_fwd", "k_vel3_bwd": "k_vel_bwd", "k_vort_pipe": "k_vort_fused", "k_bcs_div_stats_code": "k_bcs_div_stats"}.get(k, k)     # profiler names of advect_vel3.hip's launches; the pipelined fused confinement
    if k == "k_conv3_mfma":
        a = [t.strip() for t in targs.strip("<>").split(",")]
        return "k_conv3_mfma_in" if a[1] == "true" else ("k_conv3_mfma_tail" if a[2] == "true" else "k_conv3_mfma")
    if k in ("k_conv3_valu", "k_conv3_wino"):          # profiler names of conv_valu.hip's launches
        a = [t.strip() for t in targs.strip("<>").split(",")]
        return "k_conv3_in" if a[0] == "3" else ("k_conv3_tail" if a[1] == "true" else "k_conv3_mid")
    if k == "k_conv3_m16p_in":
        return "k_conv3_in"
    if k in ("k_scal3_fwd", "k_scal3_bwd"):          # advect_scalar3.hip
        return "k_scalar_fwd" if k.endswith("fwd") else "k_scalar_bwd"
    if k in ("k_conv3_m16", "k_conv3_m16z", "k_conv3_m16p", "k_conv3_m16q"):   # conv_mfma16.hip: <0|1|2> = in / mid / tail, z-marched <TAIL, ...> = mid / tail
        a = targs.strip("<>").split(",")[0].strip()
        return {"0": "k_conv3_in", "1": "k_conv3_mid", "2": "k_conv3_tail", "false": "k_conv3_mid", "true": "k_conv3_tail"}.get(a, k)
    if k == "k_conv3_m16p_f2":
        return "k_conv3_in_mid"
    if k == "k_apply_bcs_indexed_multi":
        return "k_apply_bcs_indexed"
    return k[:-3] if k.endswith("_v4") else k


def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        if k:
            tot[k] += float(r["Counter_Value"])
            n[k] += 1
    return {k: tot[k] / n[k] for k in tot}


def main():
    fetch, write, tag = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE"), sys.argv[3]
    cells = float(sys.argv[4]) if len(sys.argv) > 4 else 128.0 ** 3
    commit = sys.argv[5] if len(sys.argv) > 5 else "unknown"
    rows, out = [], {}
    for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0))):
        f, w = fetch[k], write.get(k, 0.0)
        t = (2 * f + w) * 1024
        out[k] = t
        rows.append("%-24s %11.1f %12.1f %14d %10.1f %10s" % (k, f, w, t, t / cells, ALG.get(k, "-")))
    hdr = ("# HBM-side traffic per launch from rocprofv3 PMC (separate passes: --pmc FETCH_SIZE / --pmc WRITE_SIZE), MI355X,\n"
           "# bench.py workload (3-D config 4 scene, %d cells, code at commit %s). FETCH_SIZE/WRITE_SIZE are in KiB; per MI355X_MICROARCH.md (HBM section)\n" % (int(cells), commit) +
           "# FETCH_SIZE reports half of the bytes of a wide coalesced read on gfx950 -> doubled here.\n"
           "# traffic = 2*FETCH + WRITE (bytes per launch). NOTE: the 128^3 working set fits the 256 MiB Infinity Cache,\n"
           "# whose hits these fabric-side counters include.\n"
           "%-24s %11s %12s %14s %10s %10s\n" % ("kernel", "fetch_KiB", "write_KiB", "traffic_bytes", "B/cell", "alg B/cell"))
    open(os.path.join(ROOT, "profiles", tag + "_pmc_traffic.txt"), "w").write(hdr + "\n".join(rows) + "\n")
    if int(cells) == 128 ** 3:     # the file bench.py reads for roofline.traffic (default workload only)
        conv_path = os.environ.get("TFL_CONV_PATH", "mfma16")
        out["_meta"] = {"commit": commit, "cells": int(cells), "source": tag + "_pmc_traffic.txt",
                        # git blob hash of each kernel's source file at measurement time: bench.py refuses the figure when it differs
                        "source_sha": {k: list(_kernels.source_sha(k, conv_path)) for k in out if not k.startswith("_")}}
        json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    sys.stdout.write(hdr + "\n".join(rows) + "\n")


if __name__ == "__main__":
    main()
