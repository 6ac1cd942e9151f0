#!/bin/bash
# Same-box A/B of the narrow image GEMM's activation prefetch: two register sets (stage s + 2 in flight, csrc/gemm_img.hip DGDM_IMG_DEPTH = 2)
# against one (bash tools/build_variant_lib.sh depth1 -DDGDM_IMG_DEPTH=1): the kernel alone (tools/microbench_gemm.py, f16x2 column), the
# in-kernel stamps, then the headline step replayed, alternating.       gpurun -- bash tools/ab_img_depth.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r06_img_depth_ab.txt; : > $OUT
for a in "128 160 128" "40000 160 128"; do echo "== stamps, DEPTH 2, M K N = $a" >> $OUT; $R/tools/ubench/gemm_img_stamps $a 2>&1 | tail -2 >> $OUT; done
echo "== microbench_gemm.py MATHS=f16x2 (us per launch), shipped (DEPTH 2)" >> $OUT
MATHS=f16x2 python3 $R/tools/microbench_gemm.py >> $OUT 2>&1
SHAPES=unet MATHS=f16x2 python3 $R/tools/microbench_gemm.py >> $OUT 2>&1
echo "== the same, DEPTH 1" >> $OUT
MATHS=f16x2 python3 $R/tools/run_with_lib.py $R/dgdm_histopath_lab_amd/lib/depth1/libdgdm_hip.so $R/tools/microbench_gemm.py >> $OUT 2>&1
SHAPES=unet MATHS=f16x2 python3 $R/tools/run_with_lib.py $R/dgdm_histopath_lab_amd/lib/depth1/libdgdm_hip.so $R/tools/microbench_gemm.py >> $OUT 2>&1
ARGS="--steps 40 --warmup 5 --no-cpu-baseline --no-gather --no-strict --no-raster --no-sample-loop --sustain-seconds 0"
for rep in 1 2 3; do
  for lib in shipped depth1; do
    if [ $lib = shipped ]; then CMD="python3 $R/bench.py $ARGS"; else CMD="python3 $R/tools/run_with_lib.py $R/dgdm_histopath_lab_amd/lib/depth1/libdgdm_hip.so $R/bench.py $ARGS"; fi
    $CMD 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rep $rep: %.3f ms/step  %.1f slides/s' % (d['ms_per_step'], d['value']))" | tee -a $OUT
  done
done
