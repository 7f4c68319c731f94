#!/bin/bash
# late round 5: whole GPU suite, smoke(), and the multi-rank control flow of bench.py on ONE GPU (gloo, ranks share the GPU)
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
O=$REPO/gpurun_out/r05check; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -6 | tee $O/pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
for n in 2 8; do
  TFL_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29510 + n)) bench.py --gpus $n --steps 10 --warmup 2 --blocks 2 --no-configs 2> $O/gloo$n.err | tail -1 > $O/gloo$n.json
  python - $O/gloo$n.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("gloo ranks", d["n_gpus"], "ms/step", round(d["ms_per_step"], 3), d["config"]["decomposition"], d["config"].get("strong_scaling"))
except Exception as e:
    print("gloo run failed:", e)
PY
done
tail -3 $O/gloo8.err
