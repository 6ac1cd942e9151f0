#!/bin/bash
# Same-box A/B of ops.FUSE_EPILOGUES = "auto" (DynamicGraphLayer as one node, activation backwards as GEMM epilogues from 20 000 rows on)
# against False (every activation a kernel of its own): three runs per side, alternating.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_fuse_auto_ab.txt; : > $OUT
for round in 1 2 3; do
  for sw in '"auto"' False; do
    echo "== FUSE_EPILOGUES=$sw, run $round" >> $OUT
    timeout -k 10 300 python tools/bench_with.py FUSE_EPILOGUES=$sw -- --steps 40 --no-cpu-baseline --no-strict --no-raster --no-gather --sustain-seconds 0 >> $OUT 2>&1 || exit 1
  done
done
