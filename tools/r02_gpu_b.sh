#!/bin/bash
# bench at N=1, and the multi-rank control flow with gloo on the one GPU (not a measurement)
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; O=gpurun_out/r02b; rm -rf $O; mkdir -p $O
python bench.py > $O/bench1.json 2> $O/bench1.err; echo "bench1 rc=$?"; python - <<'PY'
import json
j=json.loads(open("gpurun_out/r02b/bench1.json").read().strip().splitlines()[-1])
print({k:j[k] for k in ("value","steps_per_s","ms_per_step","scaling")}); print(j["roofline"]); print(j["advection_headline"]); print(j["config5_256"]); print(j.get("cpu_baseline"))
for n,k in sorted(j["kernels"].items(), key=lambda kv:-kv[1]["ms_per_step"]): print("%-22s %.4f ms  frac %s"%(n,k["ms_per_step"],("%.3f"%k["frac"]) if "frac" in k else "-"))
PY
for n in 2 8; do
  TFL_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 5 --warmup 1 --preroll 2 > $O/bench_gloo$n.json 2> $O/bench_gloo$n.err; echo "gloo $n rc=$?"
  tail -1 $O/bench_gloo$n.json | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print({k:j[k] for k in ('value','steps_per_s','ms_per_step','n_gpus','scaling')}); print(j['config']['slab']); print(j['config5_256'])" || tail -5 $O/bench_gloo$n.err
done
