"""Per kernel, the order of vector-memory loads and s_waitcnt vmcnt(n) in the compiled ISA (L = a load, wN = wait until
at most N loads are in flight, S = a store): how many memory round trips a wave makes in a row. A guarded load
(`if (ok) v = *p`) compiles to a branch whose join waits for vmcnt(0), i.e. "L w0 L w0 ..." = one round trip per load.
CPU-only (hipcc cross-compiles): usage: isa_loads.py <kernel name regex> file.hip [file2.hip ...] [-- extra hipcc flags]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    extra = args[args.index("--") + 1:]
    args = args[:args.index("--")]
pat, files = args[0], args[1:]
for f in files:
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        flags = ["-fno-slp-vectorize"] if os.path.basename(f).startswith("advect") else []
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                               "-I" + os.path.join(ROOT, "include"), "--offload-device-only", "-S", "-o", tmp.name, f] + flags + extra,
                              stderr=subprocess.DEVNULL)
        lines = open(tmp.name).read().splitlines()
    cur, body = None, {}
    for l in lines:
        m = re.match(r"^(_Z\S+):\s*(;.*)?$", l)
        if m:
            cur = m.group(1); body[cur] = []; continue
        if cur is not None:
            if l.startswith("\t.section") or l.startswith(".Lfunc_end"):
                cur = None; continue
            body[cur].append(l)
    for name, b in body.items():
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = re.sub(r"\(.*", "", dn.replace("(anonymous namespace)::", "").replace("void ", "").replace("tfl::", ""))
        if not re.search(pat, dn):
            continue
        seq = []
        for l in b:
            t = l.strip().split(";")[0].strip()
            if t.startswith(("global_load", "flat_load", "buffer_load")): seq.append("L")
            elif t.startswith(("global_store", "flat_store")): seq.append("S")
            elif t.startswith("s_waitcnt") and "vmcnt" in t: seq.append("w" + re.search(r"vmcnt\((\d+)\)", t).group(1))
        # the kernel's main batch of loads = the stretch between two stores that holds the most loads (a kernel may read a word or
        # two more after its stores: k_project publishes the z-slab reach word at its very end); a full drain counts when further
        # loads of THAT stretch follow it
        runs, cur_run = [], []
        for x in seq:
            if x == "S":
                runs.append(cur_run); cur_run = []
            else:
                cur_run.append(x)
        runs.append(cur_run)
        main = max(runs, key=lambda r: r.count("L"))
        drains = sum(1 for n, x in enumerate(main) if x == "w0" and "L" in main[n + 1:])
        print("%-34s %3d loads, %2d full drains before the last load:  %s" % (dn[:34], main.count("L"), drains, " ".join(seq)[:260]))
