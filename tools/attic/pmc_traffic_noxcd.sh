#!/bin/bash
# fabric traffic of the attention kernels with the workgroup order of rounds 1-5 (lib/noxcd), for the A/B row of DESIGN section 4
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_noxcd; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$R/tools/run_with_lib.py $R/dgdm_histopath_lab_amd/lib/noxcd/libdgdm_hip.so $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gather --no-strict --no-raster --no-sample-loop --sustain-seconds 0 --eager"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $B > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $B > $OUT/write.log 2>&1
python3 $R/tools/pmc_traffic.py $OUT/fetch $OUT/write > $R/gpurun_out/r06_noxcd_pmc_traffic.json
rm -rf $OUT/*/*/*.csv $OUT/*/*.db
