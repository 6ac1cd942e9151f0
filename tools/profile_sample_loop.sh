#!/bin/bash
# rocprofv3 kernel-trace summary of the denoise loop (bench.py's sample_loop leg: DiffusionLayer.sample at 10 000 x 128, 10 and 50 steps,
# graph-replayed): the per-launch duration of k_denoise_ddpm_step beside the loop time the bench line reports.
#   gpurun -- 'bash tools/profile_sample_loop.sh r06'   -> gpurun_out/<tag>_sample_loop_kernel_stats.txt
set -o pipefail
TAG=${1:-rXX}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sl_$TAG && mkdir -p /tmp/prof_sl_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_sl_$TAG/trace -o bench -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gather --no-strict --no-raster --sustain-seconds 0 \
   > $OUT/${TAG}_sample_loop_bench_line.json 2> /tmp/prof_sl_$TAG/trace.err || { tail -5 /tmp/prof_sl_$TAG/trace.err; exit 1; }
DB=$(find /tmp/prof_sl_$TAG/trace -name "*.db" | head -1)
python3 - $DB > $OUT/${TAG}_sample_loop_kernel_stats.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), avg(end - start), min(end - start), max(end - start) from kernels "
                  "where name like '%k_denoise_ddpm_step%' or name like '%normal%' or name like '%k_linear_small%' group by name").fetchall()
print("# kernels of the sample_loop leg (rocprofv3 --kernel-trace, every launch of the run: eager warm-up loops, the recording's replays)")
print("%-90s %8s %10s %10s %10s" % ("kernel", "calls", "avg_us", "min_us", "max_us"))
for n, c, a, lo, hi in rows:
    print("%-90s %8d %10.1f %10.1f %10.1f" % (n[:90], c, a / 1e3, lo / 1e3, hi / 1e3))
PY
cat $OUT/${TAG}_sample_loop_kernel_stats.txt
python3 -c "
import json,sys
d=json.loads(open('$OUT/${TAG}_sample_loop_bench_line.json').read().strip().splitlines()[-1])['sample_loop']
print('# bench line of the same run: steps10', d['steps10']['ms_per_loop'], 'ms per loop,', d['steps10']['us_per_step'], 'us per step; steps50', d['steps50']['ms_per_loop'], 'ms,', d['steps50']['us_per_step'], 'us per step')
" >> $OUT/${TAG}_sample_loop_kernel_stats.txt
tail -1 $OUT/${TAG}_sample_loop_kernel_stats.txt
