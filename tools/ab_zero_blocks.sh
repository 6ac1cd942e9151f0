#!/bin/bash
# Same-box A/B of the zero-block map (ops.ATTN_SKIP_ZERO_BLOCKS): the headline (positions in [0,1): nothing to skip, the map is pure
# overhead) and the same step on raster pixel positions (what the reference's preprocessing stores), two runs per side, alternating.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_zero_blocks_ab.txt; : > $OUT
for round in 1 2; do
  for sw in True False; do
    echo "== headline, ATTN_SKIP_ZERO_BLOCKS=$sw, run $round" >> $OUT
    timeout -k 10 300 python tools/bench_with.py ATTN_SKIP_ZERO_BLOCKS=$sw -- --no-cpu-baseline --no-strict --no-raster --no-gather --sustain-seconds 0 >> $OUT 2>&1 || exit 1
  done
done
for round in 1 2; do
  for sw in True False; do
    echo "== raster positions at 224 px, ATTN_SKIP_ZERO_BLOCKS=$sw, run $round" >> $OUT
    timeout -k 10 300 python tools/bench_with.py ATTN_SKIP_ZERO_BLOCKS=$sw -- --pixel-positions 224 --no-cpu-baseline --no-strict --no-raster --no-gather --sustain-seconds 0 >> $OUT 2>&1 || exit 1
  done
done
echo "== raster positions at 8 units, on" >> $OUT
timeout -k 10 300 python tools/bench_with.py ATTN_SKIP_ZERO_BLOCKS=True -- --pixel-positions 8 --no-cpu-baseline --no-strict --no-raster --no-gather --sustain-seconds 0 >> $OUT 2>&1 || exit 1
