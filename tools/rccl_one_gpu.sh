#!/bin/bash
# two (or $1) REAL RCCL ranks on ONE GPU: each process calls itself another host (NCCL_HOSTID), so RCCL's "Duplicate GPU" test
# (host hash + bus id) lets them through and the messages take the socket transport over the loopback interface. Not a
# measurement of anything -- it is the library's native transport (comm_rccl.cpp) run against the real librccl:
# ncclCommInitRank, grouped ncclSend / ncclRecv, ncclAllReduce, the recorded rank-step, the exact reach mode.
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
W=${1:-2}; R=$(mktemp -d /tmp/rdv.XXXXXX)
pids=()
for r in $(seq 0 $((W-1))); do
  NCCL_HOSTID=tflhost$r NCCL_SOCKET_IFNAME=lo NCCL_IB_DISABLE=1 NCCL_DEBUG=${NCCL_DEBUG:-WARN} TFL_RCCL_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 \
    timeout -k 5 ${RCCL_ONE_GPU_TIMEOUT:-240} python tests/rccl_multiproc_run.py $r $W $R > $R/out$r.txt 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
for r in $(seq 0 $((W-1))); do echo "---- rank $r"; if [ -n "$SUMMARY" ]; then grep -E "owned planes|recorded step|multiproc ok|rror|Traceback|RCCL_INIT" $R/out$r.txt; else grep -v amdgpu.ids $R/out$r.txt | tail -${TAIL:-12}; fi; done
echo "rc=$rc"
