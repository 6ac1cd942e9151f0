#!/bin/bash
# alternating same-box A/B of the headline step between the shipped library and a variant:  bash tools/attic/ab_step_libs.sh <variant> <reps> <out>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
V=$1; N=${2:-5}; OUT=$R/gpurun_out/${3:-ab_step_libs.txt}; : > $OUT
ARGS="--steps 60 --warmup 5 --no-cpu-baseline --no-gather --no-strict --no-raster --no-sample-loop --sustain-seconds 0"
for rep in $(seq 1 $N); do
  for lib in shipped $V; do
    if [ $lib = shipped ]; then CMD="python3 $R/bench.py $ARGS"; else CMD="python3 $R/tools/run_with_lib.py $R/dgdm_histopath_lab_amd/lib/$V/libdgdm_hip.so $R/bench.py $ARGS"; fi
    $CMD 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rep $rep: %.3f ms/step  %.1f slides/s' % (d['ms_per_step'], d['value']))" | tee -a $OUT
  done
done
