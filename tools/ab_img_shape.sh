#!/bin/bash
# Same-box A/B of the narrow image-GEMM's workgroup shape (WM x WN waves, NTW tiles per wave): lib/g414 (shipped), g422, g222
# built by tools/build_variant_lib.sh -DDGDM_IMG_WM=.. -DDGDM_IMG_WN=.. -DDGDM_IMG_NTW=..; two rounds, alternating.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_img_shape_ab.txt; : > $OUT
for round in 1 2; do
  for v in g414 g422 g222; do
    echo "== $v round $round: microbench_gemm unet f16x2" >> $OUT
    SHAPES=unet MATHS=f16x2 timeout -k 10 200 python tools/run_with_lib.py dgdm_histopath_lab_amd/lib/$v/libdgdm_hip.so tools/microbench_gemm.py >> $OUT 2>&1 || exit 1
  done
done
for v in g414 g422 g222; do
  echo "== $v: microbench_epilogues" >> $OUT
  timeout -k 10 300 python tools/run_with_lib.py dgdm_histopath_lab_amd/lib/$v/libdgdm_hip.so tools/microbench_epilogues.py >> $OUT 2>&1 || exit 1
done
