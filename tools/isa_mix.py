"""CPU-only: static instruction mix per kernel of one csrc file (VALU / MFMA / SALU / LDS / VMEM / waits), whole kernel
and, with --loops, per basic block that is the target of a backward branch (the loop bodies).
usage: python tools/isa_mix.py <file.hip> [name pattern] [--loops] [-- extra hipcc flags]"""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
loops = "--loops" in args
args = [a for a in args if a != "--loops"]
src = args[0]; pat = args[1] if len(args) > 1 else "."
flags = ["-fno-slp-vectorize"] if src in ("advect.hip", "advect_vel3.hip", "advect_scalar3.hip") else []
out = tempfile.mktemp(suffix=".s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                       "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-x", "hip",
                       "-o", out, src] + flags + extra, cwd=os.path.join(ROOT, "fluidnet_amd", "csrc"), stderr=subprocess.DEVNULL)
s = open(out).read(); os.unlink(out)


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    return None


lines = s.split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\S+:", l)]
for (i0, name) in starts:
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0].replace("tfl::", "").replace("void ", "")
    if not re.search(pat, dem): continue
    i1 = next(j for j in range(i0, len(lines)) if lines[j].startswith(".Lfunc_end"))
    body = lines[i0 + 1:i1]
    tot = collections.Counter(); blocks = []; cur = ("entry", collections.Counter()); blocks.append(cur)
    labels = {}
    for k, l in enumerate(body):
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            cur = (m.group(1), collections.Counter()); blocks.append(cur); labels[m.group(1)] = len(blocks) - 1
            continue
        mm = re.match(r"\s+([a-z_0-9]+)\s", l + " ")
        if not mm: continue
        c = classify(mm.group(1))
        if c: tot[c] += 1; cur[1][c] += 1
        tgt = re.search(r"s_c?branch\S*\s+(\.LBB\S+)", l)
        if tgt: cur[1]["->" + tgt.group(1)] += 1
    print("%-44s %s" % (dem[:44], " ".join("%s %d" % kv for kv in sorted(tot.items()))))
    if loops:
        for bi, (lab, c) in enumerate(blocks):
            for k in list(c):
                if k.startswith("->") and k[2:] in labels and labels[k[2:]] <= bi:      # backward branch: blocks [target, bi] = a loop
                    lc = collections.Counter()
                    for (_, cc) in blocks[labels[k[2:]]:bi + 1]:
                        for kk, vv in cc.items():
                            if not kk.startswith("->"): lc[kk] += vv
                    print("    loop %s..%s: %s" % (k[2:], lab, " ".join("%s %d" % kv for kv in sorted(lc.items()))))
