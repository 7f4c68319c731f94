import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{"metric"'):
        d=json.loads(l)
        print("ms/step %.4f  cfg5 %.4f  sum_kernel %.4f"%(d["ms_per_step"], (d.get("config5_256") or {}).get("ms_per_step",0), d["sum_kernel_ms"]))
        print("  "+"  ".join("%s %.1f"%(k[2:],v["avg_ms"]*1e3) for k,v in d["kernels"].items()))
