#!/bin/bash
# One GPU-box session that produces everything profiles/ holds for a round: the default bench line, the rocprofv3 kernel-trace
# summary of the same command, and the PMC passes (each in its own run, --kernel-trace only -- never with sys/hip traces).
#   gpurun -- 'bash tools/profile_round.sh r02a'      -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
set -o pipefail
TAG=${1:-rXX}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
BENCH_PROF="$REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-gather --no-strict --no-raster --no-sample-loop --sustain-seconds 0"
[ -n "$ONLY_TRACE" ] || python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || { tail -5 $OUT/${TAG}_bench.err; exit 1; }
[ -n "$ONLY_TRACE" ] || { echo "bench done"; cut -c1-400 $OUT/${TAG}_bench.json; }
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG && mkdir -p /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/trace -o bench -- python3 $BENCH_PROF > $OUT/${TAG}_trace_bench_line.json 2> /tmp/prof_$TAG/trace.err || { tail -5 /tmp/prof_$TAG/trace.err; exit 1; }
DB=$(find /tmp/prof_$TAG/trace -name "*.db" | head -1)
python3 $REPO/tools/kernel_stats_from_db.py $DB k_attn_h_fwd 400 $OUT/${TAG}_sequence.txt > $OUT/${TAG}_kernel_stats.txt && head -30 $OUT/${TAG}_kernel_stats.txt
echo "trace done"
[ -n "$ONLY_TRACE" ] && exit 0      # ONLY_TRACE=1: no PMC passes
PMCB="$REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gather --no-strict --no-raster --no-sample-loop --sustain-seconds 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
   --kernel-trace --output-format csv -d /tmp/prof_$TAG/valu -- python3 $PMCB > /dev/null 2> /tmp/prof_$TAG/valu.err || { tail -5 /tmp/prof_$TAG/valu.err; exit 1; }
python3 $REPO/tools/pmc_valu.py /tmp/prof_$TAG/valu > $OUT/${TAG}_pmc_valu.json && echo "valu pass done"
PMCT="$REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strict --no-raster --no-sample-loop --sustain-seconds 0"     # with the gather microbenchmark: its traffic is the calibration row
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_$TAG/fetch -- python3 $PMCT > /dev/null 2> /tmp/prof_$TAG/fetch.err || { tail -5 /tmp/prof_$TAG/fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_$TAG/write -- python3 $PMCT > /dev/null 2> /tmp/prof_$TAG/write.err || { tail -5 /tmp/prof_$TAG/write.err; exit 1; }
python3 $REPO/tools/pmc_traffic.py /tmp/prof_$TAG/fetch /tmp/prof_$TAG/write > $OUT/${TAG}_pmc_traffic.json && echo "traffic passes done"
