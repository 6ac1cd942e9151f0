#!/bin/bash
# Same-box A/B of the two-level accumulation in the dW kernel (csrc/gemm_h.hip: TN_FLUSH = 4 stages, shipped) against the single
# accumulator (bash tools/build_variant_lib.sh noflush -DDGDM_TN_FLUSH=1073741824), alternating, the headline step replayed:
#   gpurun -- bash tools/ab_tn_flush.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r06_tn_flush_ab.txt; : > $OUT
ARGS="--steps 40 --warmup 5 --no-cpu-baseline --no-gather --no-strict --no-raster --no-sample-loop --sustain-seconds 0"
for rep in 1 2 3; do
  for lib in shipped noflush; do
    if [ $lib = shipped ]; then CMD="python3 $R/bench.py $ARGS"; else CMD="python3 $R/tools/run_with_lib.py $R/dgdm_histopath_lab_amd/lib/noflush/libdgdm_hip.so $R/bench.py $ARGS"; fi
    $CMD 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rep $rep: %.3f ms/step  %.1f slides/s' % (d['ms_per_step'], d['value']))" | tee -a $OUT
  done
done
