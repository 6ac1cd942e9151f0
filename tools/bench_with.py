"""Run bench.py with module attributes of dgdm_histopath_lab_amd.ops overridden first (same-box A/B of a switch that is not an
environment variable):  python tools/bench_with.py TN_GROUPED=False TN_GROUP_SORT=False -- --no-cpu-baseline --no-strict"""
import ast, os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dgdm_histopath_lab_amd import ops  # noqa: E402

args = sys.argv[1:]
split = args.index("--") if "--" in args else len(args)
for kv in args[:split]:
    k, v = kv.split("=", 1)
    if not hasattr(ops, k):
        raise SystemExit(f"ops has no attribute {k}")
    setattr(ops, k, ast.literal_eval(v))
sys.argv = [os.path.join(ROOT, "bench.py")] + args[split + 1:]
runpy.run_path(sys.argv[0], run_name="__main__")
