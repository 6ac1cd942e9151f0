#!/bin/bash
# PMC passes of the bench command on the GPU box (each counter set in a pass of its own, --kernel-trace only, as the guide and gpurun
# require), summarised into gpurun_out/<tag>_pmc_valu.json and <tag>_pmc_traffic.json:
#   gpurun -- 'bash tools/profile_pmc.sh r03 [extra bench.py flags, e.g. --precision fp32]'
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gather --no-strict --no-raster --no-sample-loop --sustain-seconds 0 --eager $@"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/valu -- python3 $B > $OUT/valu.log 2>&1
python3 $R/tools/pmc_valu.py $OUT/valu > $R/gpurun_out/${TAG}_pmc_valu.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $B > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $B > $OUT/write.log 2>&1
python3 $R/tools/pmc_traffic.py $OUT/fetch $OUT/write > $R/gpurun_out/${TAG}_pmc_traffic.json
rm -rf $OUT/*/*/*.csv $OUT/*/*.db
python3 - <<PY
import json
v = json.load(open("$R/gpurun_out/${TAG}_pmc_valu.json"))["kernels"]
t = json.load(open("$R/gpurun_out/${TAG}_pmc_traffic.json"))["kernels"]
for k in list(v)[:8]:
    print(k, {a: v[k][a] for a in ("valu_busy", "mfma_busy", "valu_insts_per_launch", "occupancy_waves_per_simd")}, t.get(k, {}).get("fabric_bytes_per_launch"))
PY
