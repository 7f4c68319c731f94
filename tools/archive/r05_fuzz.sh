# late round 5: random scenes, HIP <-> oracle, at HEAD: the default kernels, the big-grid variants (two-plane advectVel, fused
# confinement = k_vort_pipe), the hardware block order and runs of 3 tiles -> gpurun_out/r05fuzz/ (copied to profiles/r05_fuzz_parity.txt)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05fuzz; rm -rf $O; mkdir -p $O
{
echo "# tools/fuzz_parity.py at $(cat .git_rev 2>/dev/null): n scenes, seed; every operator of the step, all advection methods, bit-exact bar"
echo "== defaults";                                   timeout 200 python tools/fuzz_parity.py 120 9101 2>&1 | tail -3
echo "== TFL_VEL3_KZ=2 TFL_SCAL3_TZ=14 TFL_VORT_FUSED=1"; TFL_VEL3_KZ=2 TFL_SCAL3_TZ=14 TFL_VORT_FUSED=1 timeout 200 python tools/fuzz_parity.py 120 9202 2>&1 | tail -3
echo "== TFL_XCD_ORDER=0";                            TFL_XCD_ORDER=0 timeout 200 python tools/fuzz_parity.py 80 9303 2>&1 | tail -3
echo "== TFL_XCD_RUN=3 TFL_VORT_FUSED=1 TFL_VORT_PIPE=0"; TFL_XCD_RUN=3 TFL_VORT_FUSED=1 TFL_VORT_PIPE=0 timeout 200 python tools/fuzz_parity.py 80 9404 2>&1 | tail -3
} | tee $O/fuzz.txt
