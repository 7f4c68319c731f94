cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04fuzz; rm -rf $O; mkdir -p $O
timeout 500 python -m pytest tests/test_hip_parity.py -x -q -k "variants" 2>&1 | tail -15 | tee $O/variants.txt
timeout 150 python tools/fuzz_parity.py 150 7777 2>&1 | tail -5 | tee $O/fuzz_default.txt
TFL_VEL3_KZ=2 TFL_SCAL3_TZ=14 timeout 150 python tools/fuzz_parity.py 150 8888 2>&1 | tail -5 | tee $O/fuzz_big.txt
