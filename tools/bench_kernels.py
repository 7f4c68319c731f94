"""Pretty-print the per-kernel table of a bench.py JSON line (stdin or file): us per step, sorted."""
import json
import sys

src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
line = [l for l in src.read().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
print(f"{d['ms_per_step']:.4f} ms/step  {d['value']:.1f} {d['unit']}  roofline {d['roofline']['kernel'] if 'kernel' in d['roofline'] else ''} "
      f"{d['roofline']['achieved']:.1f}/{d['roofline']['peak']} {d['roofline']['unit']}")
tot = 0.0
for name, k in sorted(d.get("kernels", {}).items(), key=lambda kv: -kv[1]["ms_per_step"]):
    tot += k["ms_per_step"]
    print(f"  {name:28s} {k['ms_per_step'] * 1e3:8.1f} us/step  ({k['launches_per_step']:.0f} x {k['avg_ms'] * 1e3:.1f} us)")
print(f"  {'sum':28s} {tot * 1e3:8.1f} us/step")
