"""Pretty-print the per-kernel table of a bench.py JSON line (stdin or file): us per step, sorted."""
import json
import sys

src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
line = [l for l in src.read().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
r = d["roofline"]
print(f"{d['ms_per_step']:.4f} ms/step  {d['value']:.1f} {d['unit']}  roofline {r.get('kernel', '')} "
      f"{(r.get('achieved') or 0.0):.1f}/{r.get('peak')} {r.get('unit')}")
tot = 0.0
for name, k in sorted(d.get("kernels", {}).items(), key=lambda kv: -kv[1]["ms_per_step"]):
    tot += k["ms_per_step"]
    print(f"  {name:28s} {k['ms_per_step'] * 1e3:8.1f} us/step  ({k['launches_per_step']:.0f} x {k['avg_ms'] * 1e3:.1f} us)")
print(f"  {'sum':28s} {tot * 1e3:8.1f} us/step")
