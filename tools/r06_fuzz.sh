# round 6: random scenes, HIP <-> oracle, at HEAD (one transcendental per back-trace, the consumer-side statistics sum, k_vort_pipe's prologue): the default
# kernels, the big-grid variants, and -- against the EXPERIMENTS flavour -- the hardware block order and runs of 3 tiles
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06fuzz; rm -rf $O; mkdir -p $O
E=$PWD/fluidnet_amd/libtfluids_hip_exp.so
{
echo "# tools/fuzz_parity.py, round 6: n scenes, seed; every operator of the step, all advection methods, bit-exact bar"
echo "== defaults (product library)";                 timeout 300 python tools/fuzz_parity.py 160 6101 2>&1 | tail -3
echo "== TFL_VEL3_KZ=2 TFL_SCAL3_TZ=14 TFL_VORT_FUSED=1 (product library)"; TFL_VEL3_KZ=2 TFL_SCAL3_TZ=14 TFL_VORT_FUSED=1 timeout 300 python tools/fuzz_parity.py 160 6202 2>&1 | tail -3
echo "== TFL_XCD_ORDER=0 (EXPERIMENTS flavour)";      TFL_LIBRARY=$E TFL_XCD_ORDER=0 timeout 200 python tools/fuzz_parity.py 80 6303 2>&1 | tail -3
echo "== TFL_XCD_RUN=3 TFL_VORT_FUSED=1 TFL_VORT_PIPE=0 TFL_SCAL3_MARCH=1 (EXPERIMENTS flavour)"; TFL_LIBRARY=$E TFL_XCD_RUN=3 TFL_VORT_FUSED=1 TFL_VORT_PIPE=0 TFL_SCAL3_MARCH=1 timeout 200 python tools/fuzz_parity.py 80 6404 2>&1 | tail -3
echo "== TFL_VORT_FUSED=1 TFL_VORT_TILE=32 TFL_VORT_CZ=5 (EXPERIMENTS flavour)"; TFL_LIBRARY=$E TFL_VORT_FUSED=1 TFL_VORT_TILE=32 TFL_VORT_CZ=5 timeout 200 python tools/fuzz_parity.py 80 6505 2>&1 | tail -3
} | tee $O/fuzz.txt
