#!/bin/bash
# Same-box A/B of the product-block accumulation of the image GEMMs (csrc/gemm_img.hip: DGDM_IMG_FRESH = 1, shipped) against round 5's
# accumulation (bash tools/build_variant_lib.sh nofresh -DDGDM_IMG_FRESH=0), alternating, the headline step replayed:
#   gpurun -- bash tools/ab_img_fresh.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r06_img_fresh_ab.txt; : > $OUT
ARGS="--steps 40 --warmup 5 --no-cpu-baseline --no-gather --no-strict --no-raster --no-sample-loop --sustain-seconds 0"
for rep in 1 2 3; do
  for lib in shipped nofresh; do
    if [ $lib = shipped ]; then CMD="python3 $R/bench.py $ARGS"; else CMD="python3 $R/tools/run_with_lib.py $R/dgdm_histopath_lab_amd/lib/nofresh/libdgdm_hip.so $R/bench.py $ARGS"; fi
    $CMD 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rep $rep: %.3f ms/step  %.1f slides/s' % (d['ms_per_step'], d['value']))" | tee -a $OUT
  done
done
python3 $R/tools/microbench_gemm.py > $R/gpurun_out/r06_gemm_mb_fresh.txt 2>&1
python3 $R/tools/run_with_lib.py $R/dgdm_histopath_lab_amd/lib/nofresh/libdgdm_hip.so $R/tools/microbench_gemm.py > $R/gpurun_out/r06_gemm_mb_nofresh.txt 2>&1
