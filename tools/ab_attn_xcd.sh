#!/bin/bash
# Same-box A/B of the XCD-aware workgroup order of the attention kernels (csrc/attn_h.hpp: DGDM_ATTN_XCD = 1, shipped) against the
# order of rounds 1-5 (bash tools/build_variant_lib.sh noxcd -DDGDM_ATTN_XCD=0), alternating, the headline step replayed; then the
# kernel durations of one eager step each (the bench line's roofline object).     gpurun -- bash tools/ab_attn_xcd.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r06_attn_xcd_ab.txt; : > $OUT
ARGS="--steps 40 --warmup 5 --no-cpu-baseline --no-gather --no-strict --no-raster --no-sample-loop --sustain-seconds 0"
for rep in 1 2 3; do
  for lib in shipped noxcd; do
    if [ $lib = shipped ]; then CMD="python3 $R/bench.py $ARGS"; else CMD="python3 $R/tools/run_with_lib.py $R/dgdm_histopath_lab_amd/lib/noxcd/libdgdm_hip.so $R/bench.py $ARGS"; fi
    $CMD 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$lib rep $rep: %.3f ms/step  %.1f slides/s | backward %.4f ms, forward %.4f ms, reduce %.4f ms' % (d['ms_per_step'], d['value'], r['ms_per_launch'], r['other_kernels_ms']['attn_fwd'], r['other_kernels_ms']['attn_bwd_dq_reduce']))" | tee -a $OUT
  done
done
