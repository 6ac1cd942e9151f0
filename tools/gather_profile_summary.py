#!/usr/bin/env python3
"""Summary of tools/profile_gather.sh: per case (warm / cold) the rocprofv3 kernel-trace duration of the gather kernel and its
counter traffic (FETCH_SIZE x 2 on gfx950 for 16 B/lane reads + WRITE_SIZE, MI355X_MICROARCH.md HBM section), beside the
algorithmic and unique byte counts.  JSON to stdout (-> profiles/rNN_gather_pmc_traffic.json, read by bench.py),
a text table into <workdir>/summary_kernel_stats.txt (-> profiles/rNN_gather_kernel_stats.txt)."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import per_kernel, short_name  # noqa: E402

N, E, C = 10000, 50000, 768
ALG = (E + N) * C * 4 + N * C * 4 + (E + N) * 8 + (N + 1) * 4
UNIQ = N * C * 4 + N * C * 4 + (E + N) * 8 + (N + 1) * 4


def trace_durations(dirname):
    """{short kernel name: sorted durations in us} from a --kernel-trace csv."""
    out = {}
    for path in glob.glob(f"{dirname}/**/*kernel_trace.csv", recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                out.setdefault(short_name(row["Kernel_Name"]), []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    return {k: sorted(v) for k, v in out.items()}


def main():
    work, tag = sys.argv[1], sys.argv[2]
    res = {"unit": "bytes per launch (FETCH_SIZE KB x2 gfx950 correction + WRITE_SIZE KB)", "command": "tools/profile_gather.sh " + tag,
           "workload": f"{N} nodes x {C} feat, {E}+{N} entries", "algorithmic_bytes": ALG, "unique_bytes": UNIQ, "kernels": {}, "cases": {}}
    lines = []
    for mode in ("warm", "cold"):
        dur = trace_durations(f"{work}/{mode}_trace")
        fetch, write = per_kernel(f"{work}/{mode}_fetch", "FETCH_SIZE"), per_kernel(f"{work}/{mode}_write", "WRITE_SIZE")
        names = [k for k in dur if k.startswith("k_spmm<")]
        lines.append(f"# {mode}: rocprofv3 --kernel-trace of tools/microbench_gather.py --widths 768" + (" --cold" if mode == "cold" else ""))
        lines.append(f"{'kernel':40s} {'calls':>6s} {'avg_us':>9s} {'median_us':>9s} {'min_us':>8s}")
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            lines.append(f"{k[:40]:40s} {len(v):6d} {sum(v) / len(v):9.2f} {v[len(v) // 2]:9.2f} {v[0]:8.2f}")
        for k in names:
            v = dur[k]
            nf, f = fetch.get(k, (0, 0.0))
            nw, w = write.get(k, (0, 0.0))
            rd, wr = 2.0 * f * 1024 / max(nf, 1), w * 1024 / max(nw, 1)
            us = v[len(v) // 2] if mode == "cold" else sum(v) / len(v)
            ent = {"launches": len(v), "us_per_launch": round(us, 2), "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                   "fabric_bytes_per_launch": round(rd + wr), "traffic_over_algorithmic": round((rd + wr) / ALG, 3),
                   "traffic_over_unique": round((rd + wr) / UNIQ, 3), "algorithmic_GBps": round(ALG / us / 1e3, 1),
                   "algorithmic_frac_of_8TBps": round(ALG / us / 1e3 / 8000, 4)}
            res["cases"].setdefault(mode, {})[k] = ent
            if mode == "warm":
                res["kernels"][k] = ent
            lines.append(f"# {mode} {k}: {us:.2f} us, counter traffic {(rd + wr) / 1e6:.1f} MB per launch = {(rd + wr) / ALG:.2f} x algorithmic "
                         f"({ALG / 1e6:.1f} MB) = {(rd + wr) / UNIQ:.2f} x unique ({UNIQ / 1e6:.1f} MB); {ALG / us / 1e3:.0f} GB/s algorithmic")
        lines.append("")
    open(os.path.join(work, "summary_kernel_stats.txt"), "w").write("\n".join(lines) + "\n")
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
