#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; O=gpurun_out/r02conv; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_simulate.py tests/test_hip_fullsize.py -m gpu -q -x -k "conv_paths or model_forward or config3 or baseline_size or zslab" 2>&1 | tail -4
for p in mfma valu; do
  if [ $p = mfma ]; then export TFL_CONV_PATH=mfma; else unset TFL_CONV_PATH; fi
  python bench.py --no-cpu-baseline --no-config5 --steps 30 > $O/bench_$p.json 2>/dev/null
  python - $O/bench_$p.json $p <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms/step %.4f"%j["ms_per_step"], {n: (round(k["ms_per_step"]*1e3,1), round(k.get("frac",0),3)) for n,k in j["kernels"].items() if "conv" in n})
PY
  python bench.py --no-cpu-baseline --no-config5 --steps 10 --res 256 --preroll 4 > $O/bench256_$p.json 2>/dev/null
  python - $O/bench256_$p.json "$p 256" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms/step %.4f"%j["ms_per_step"], {n: round(k["ms_per_step"]*1e3,1) for n,k in j["kernels"].items() if "conv" in n})
PY
done
