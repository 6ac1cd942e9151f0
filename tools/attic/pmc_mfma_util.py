#!/usr/bin/env python3
"""Matrix-pipe utilisation per kernel from one rocprofv3 --pmc pass:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA --kernel-trace --output-format csv -d OUT -- python3 bench.py ...
    python tools/pmc_mfma_util.py OUT > profiles/rNN_pmc_mfma_util.json

SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the chip's 1024 SIMDs (256 CUs x 4); GRBM_GUI_ACTIVE counts
cycles per XCD summed over the 8 XCDs (MI355X_MICROARCH.md, PMC section).  utilisation = busy / (1024 * active / 8).
"""
import csv, glob, json, re, sys, collections
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_traffic import short_name


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for path in glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path, newline="")):
            n = short_name(row["Kernel_Name"])
            acc[n][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cnt[n] += 1
    out = {}
    for n, c in acc.items():
        busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
        if busy <= 0 or act <= 0:
            continue
        out[n] = {"launches": cnt[n], "mfma_busy_cycles_per_launch": round(busy / cnt[n]), "kernel_cycles_per_launch": round(act / 8 / cnt[n]),
                  "mfma_pipe_utilisation": round(busy / (1024.0 * act / 8.0), 4), "mfma_insts_per_launch": round(c.get("SQ_INSTS_MFMA", 0.0) / cnt[n])}
    json.dump({"note": "fraction of SIMD-cycles in which the matrix pipe was busy, per kernel, averaged over launches",
               "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"] * kv[1]["launches"]))}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
