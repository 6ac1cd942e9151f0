"""Condense a rocprofv3 *_kernel_stats.csv into a short table (names trimmed, per-step times)."""
import csv, re, sys
path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    if n.startswith("Cijk_"):
        m = re.search(r"(Cijk_\w+?_S)_.*?(MT\d+x\d+x\d+)", n)
        return f"hipBLASLt {m.group(1)} {m.group(2)}" if m else n[:60]
    n = re.sub(r"at::native::", "", n)
    return n.split("(")[0][:90]
print(f"# total kernel time {tot/1e6:.2f} ms over {steps:g} steps = {tot/1e6/steps:.2f} ms/step ({path})")
print(f"{'kernel':92s} {'calls':>6s} {'avg_us':>9s} {'ms/step':>8s} {'%':>6s}")
agg = {}
for r in rows:
    k = short(r["Name"]); a = agg.setdefault(k, [0, 0.0]); a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{k:92s} {c:6d} {t/c/1e3:9.1f} {t/1e6/steps:8.3f} {100*t/tot:6.2f}")
