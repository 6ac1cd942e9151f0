#!/bin/bash
# The bench lines profiles/ holds for a round beside the default one: the default line again (final tree), the mixed-size stream
# (configs[4]), configs[3] on BASELINE's positions and on raster pixel positions.   gpurun -- bash tools/bench_round.sh r06
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
python3 $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python3 $R/bench.py --mixed --no-cpu-baseline --no-gather > $O/${TAG}_bench_mixed.json 2> $O/${TAG}_bench_mixed.err
python3 $R/bench.py --large --no-cpu-baseline --no-gather --steps 10 --warmup 3 > $O/${TAG}_bench_large.json 2> $O/${TAG}_bench_large.err
python3 $R/bench.py --large --pixel-positions 224 --no-cpu-baseline --no-gather --steps 10 --warmup 3 > $O/${TAG}_bench_large_raster.json 2> $O/${TAG}_bench_large_raster.err
for f in bench bench_mixed bench_large bench_large_raster; do
  python3 - "$O/${TAG}_$f.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], "slides/s", d["ms_per_step"], "ms |", r.get("kernel"), r.get("ms_per_launch"), "ms", r.get("achieved"), "TF frac", r.get("frac"),
          "pairs_live", (r.get("pairs_live") or {}).get("frac"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
