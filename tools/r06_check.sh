#!/bin/bash
# late round 6 at HEAD: the whole GPU suite against BOTH flavours of the library + smoke()
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
O=$REPO/gpurun_out/r06check; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_default.txt 2>&1; echo "product library rc=$?"; grep -E "passed|failed" $O/pytest_default.txt | tail -1
TFL_LIBRARY=$REPO/fluidnet_amd/libtfluids_hip_exp.so timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_exp.txt 2>&1; echo "EXPERIMENTS flavour rc=$?"; grep -E "passed|failed" $O/pytest_exp.txt | tail -1
grep -E "^FAILED|^ERROR" $O/pytest_default.txt $O/pytest_exp.txt | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
# the native transport against the real librccl, 2 / 3 / 4 / 8 ranks on this box's one GPU (tools/rccl_one_gpu.sh)
{ for w in 2 3 4 8; do echo "==== $w ranks"; SUMMARY=1 bash tools/rccl_one_gpu.sh $w 2>&1; done; echo "==== transport lines of a 2-rank run (NCCL_DEBUG=INFO)"; NCCL_DEBUG=INFO TAIL=400 bash tools/rccl_one_gpu.sh 2 2>&1 | grep -E "RCCL version|nNodes|via NET|Init COMPLETE|Duplicate" | sed -e 's/^runc:[0-9]*:[0-9]* //' | sort | uniq -c | sort -rn | head -20; } > $O/rccl_one_gpu.txt 2>&1; tail -3 $O/rccl_one_gpu.txt
