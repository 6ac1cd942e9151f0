#!/bin/bash
# rocprofv3 evidence for the north-star gather as it is NOW (core/graph_layers.py:92,99-110 of the reference): the
# 10k x 768 aggregation (k_spmm<64,3,4,false>), warm (back-to-back launches) and cold (1 GiB rewritten before every launch):
# kernel trace + the two traffic passes (FETCH_SIZE and WRITE_SIZE cannot share a pass), each in a run of its own with
# --kernel-trace only, the program directly after `--` (no env / bash -c hop).
#   gpurun -- 'bash tools/profile_gather.sh r04'   ->  gpurun_out/<tag>_gather_{kernel_stats.txt,pmc_traffic.json}
set -e -o pipefail
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
W=/tmp/gather_$TAG
rm -rf $W; mkdir -p $W $OUT
cd /tmp && export TMPDIR=/tmp
for MODE in warm cold; do
  if [ $MODE = warm ]; then ARGS="--widths 768 --iters 200"; else ARGS="--widths 768 --iters 24 --cold"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $W/${MODE}_trace -- python3 $R/tools/microbench_gather.py $ARGS > $W/${MODE}_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $W/${MODE}_fetch -- python3 $R/tools/microbench_gather.py $ARGS > $W/${MODE}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $W/${MODE}_write -- python3 $R/tools/microbench_gather.py $ARGS > $W/${MODE}_write.log 2>&1
  echo "$MODE passes done"
done
python3 $R/tools/gather_profile_summary.py $W $TAG > $OUT/${TAG}_gather_pmc_traffic.json
cp $W/summary_kernel_stats.txt $OUT/${TAG}_gather_kernel_stats.txt
cat $OUT/${TAG}_gather_kernel_stats.txt
