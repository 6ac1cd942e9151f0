#!/bin/bash
# Kernel trace of the default bench command on the GPU box, summarised into profiles/<tag>_bench_kernel_stats.txt (+ the launch sequence
# of the last step and the bench line of the profiled run).  Run through gpurun from the repository root:
#   gpurun -- 'bash tools/profile_bench.sh r03a [extra bench.py flags]'
# OPS="FUSE_EPILOGUES=False ..." profiles the same command with module attributes of ops overridden (tools/bench_with.py).
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ -n "$OPS" ]; then BENCH="$R/tools/bench_with.py $OPS --"; else BENCH="$R/bench.py"; fi
rocprofv3 --kernel-trace --stats -d $OUT -- python3 $BENCH --steps 16 --warmup 3 --no-cpu-baseline --no-gather --no-strict --no-raster --no-sample-loop --sustain-seconds 0 "$@" > $OUT/bench_line.json 2> $OUT/bench.err
DB=$(find $OUT -name "*results.db" | head -1)
python3 $R/tools/kernel_stats_from_db.py $DB "${MARKER:-k_attn_h_fwd}" 60 $R/gpurun_out/${TAG}_launch_sequence.txt > $R/gpurun_out/${TAG}_bench_kernel_stats.txt
cp $OUT/bench_line.json $R/gpurun_out/${TAG}_bench_kernel_stats_bench_line.json
[ -n "$KEEP_DB" ] || rm -f $DB
head -40 $R/gpurun_out/${TAG}_bench_kernel_stats.txt
