#!/bin/bash
# round 4, late: the whole GPU suite (incl. the PCG test with the compiled reference host function beside the restatement),
# smoke(), and the default bench line at HEAD -> gpurun_out/r04val/
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
O=$REPO/gpurun_out/r04val; rm -rf $O; mkdir -p $O
ls -la oracle/_ref/ > $O/ref_libs.txt 2>&1
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 240 python bench.py 2> $O/bench.err | tail -1 > $O/bench.json; echo "bench rc=$?"
python - $O/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("ms/step", round(d["ms_per_step"], 4), "min/max", round(d["ms_per_step_min"], 4), round(d["ms_per_step_max"], 4), "value", round(d["value"], 1),
      "roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), "traffic", d["roofline"]["traffic"], "cfg5", round(d["config5_256"]["ms_per_step"], 3),
      "cpu", round(d["cpu_baseline"]["value"], 2))
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:14]:
    print("  %-18s %.1f us/step" % (k, v["ms_per_step"] * 1e3))
PY
