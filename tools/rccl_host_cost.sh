#!/bin/bash
# host cost of a rank-step over the REAL RCCL (tools/rccl_host_cost.py), ranks sharing this box's GPU. usage: rccl_host_cost.sh [world] [res]
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
W=${1:-2}; RES=${2:-128}; R=$(mktemp -d /tmp/rdv.XXXXXX)
pids=()
for r in $(seq 0 $((W-1))); do
  NCCL_HOSTID=tflhost$r NCCL_SOCKET_IFNAME=lo NCCL_IB_DISABLE=1 NCCL_DEBUG=WARN HSA_ENABLE_IPC_MODE_LEGACY=0 \
    timeout -k 5 300 python tools/rccl_host_cost.py $r $W $R $RES > $R/out$r.txt 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
for r in $(seq 0 $((W-1))); do grep -oE "res [0-9]+ rank.*|.*rror.*|Traceback.*" $R/out$r.txt; done
[ $rc -ne 0 ] && { echo "rc=$rc"; tail -20 $R/out0.txt; }
exit 0
