# A/B of the block -> tile order (tfl_device.hpp block_tile): TFL_XCD_ORDER = 0 hardware | 1 one run per XCD | 2 an eighth of a plane
# per XCD; TFL_XCD_RUN = tiles per run. One session, alternating.
for rep in 1 2; do
for m in "TFL_XCD_ORDER=0" "TFL_XCD_ORDER=1" "TFL_XCD_ORDER=2" "TFL_XCD_RUN=8" "TFL_XCD_RUN=32"; do
  echo "== $m"
  env $m python tools/adv_abl.py 2>/dev/null
  env $m python tools/vort_abl.py 2>/dev/null
done; done
for m in "TFL_XCD_ORDER=0" "TFL_XCD_ORDER=1" "TFL_XCD_ORDER=2" "TFL_XCD_RUN=8" "TFL_XCD_RUN=32"; do
  echo "== bench $m"
  env $m python bench.py --steps 50 --no-configs --no-cpu-baseline --no-config5 2>/dev/null | python tools/bench_kernels_line.py
  env $m python bench.py --res 256 --steps 10 --blocks 3 --no-configs --no-config5 --no-cpu-baseline 2>/dev/null | python tools/bench_kernels_line.py
done
