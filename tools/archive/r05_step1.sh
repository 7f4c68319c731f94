#!/bin/bash
# round 5, session 4: the buoyancy fold / From-confinement / range gate / bench audit -- parity first, then the bench line
mkdir -p gpurun_out/r05d
timeout 1500 python -m pytest tests/test_hip_simulate.py -x -q -m gpu -k "native_simulate or bc_fold or fp16_range or conv_fused or config3_4 or graphed or zslab_decomposition_equals" 2>&1 | tail -15 | tee gpurun_out/r05d/tests.txt
for cfg in "TFL_BUOY_FOLD=0" "TFL_BUOY_FOLD=1"; do
  echo "== bench [$cfg]" | tee -a gpurun_out/r05d/bench.txt
  env $cfg timeout 300 python bench.py --no-cpu-baseline --no-config5 --no-configs --steps 50 2>gpurun_out/r05d/bench_err.txt | tail -1 > gpurun_out/r05d/b.json
  python tools/bench_kernels.py < gpurun_out/r05d/b.json | tee -a gpurun_out/r05d/bench.txt
done
timeout 600 python bench.py --steps 50 2>>gpurun_out/r05d/bench_err.txt | tail -1 > gpurun_out/r05d/bench_full.json
python - <<'PY' | tee -a gpurun_out/r05d/bench.txt
import json
j = json.load(open("gpurun_out/r05d/bench_full.json"))
for k in ("ms_per_step", "value", "range_errors", "trace_errors", "conv_exact_fp32", "conv_witness_ratio", "config5_256"):
    print(k, j.get(k))
print("cpu per-op", json.dumps(j["cpu_baseline"]["per_op_ms"])[:1500])
print("witness", j["cpu_baseline"].get("conv_witness"))
print("roofline", {k: v for k, v in j["roofline"].items() if k not in ("note",)})
print({k: v["steps_per_s"] for k, v in (j.get("configs") or {}).items() if "steps_per_s" in v})
PY
tail -5 gpurun_out/r05d/bench_err.txt
