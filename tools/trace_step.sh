#!/bin/bash
# rocprofv3 kernel trace of the default bench command -> gpurun_out/<tag>_kernel_stats.txt (all kernels) + the bench line of that run
set -o pipefail
TAG=${1:-rXX}; shift
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$TAG && mkdir -p /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/trace -o bench -- python3 $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-gather --no-strict --no-raster "$@" > $OUT/${TAG}_trace_bench_line.json 2> /tmp/prof_$TAG/trace.err || { tail -5 /tmp/prof_$TAG/trace.err; exit 1; }
DB=$(find /tmp/prof_$TAG/trace -name "*.db" | head -1)
python3 $REPO/tools/kernel_stats_from_db.py $DB k_attn_h_bwd_dkv > $OUT/${TAG}_kernel_stats.txt && head -4 $OUT/${TAG}_kernel_stats.txt
python3 -c "import json;d=json.load(open('$OUT/${TAG}_trace_bench_line.json'));print(d['value'], d['ms_per_step'])"
