#!/usr/bin/env python3
"""VALU / issue-stall accounting per kernel from ONE rocprofv3 --pmc pass (8 SQ slots + 1 GRBM):

    (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY \
        SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
        --kernel-trace --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gather --eager)
    python tools/pmc_valu.py $OUT > profiles/rNN_pmc_valu.json

Units (MI355X_MICROARCH.md, per-instruction table): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count QUAD-cycles summed
over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE counts cycles summed over the 8
XCDs.  Reported per kernel, averaged over its launches:
  * valu_busy     = 4 * SQ_ACTIVE_INST_VALU / (1024 * GRBM_GUI_ACTIVE / 8): fraction of SIMD-cycles in which a vector
                    instruction of some wave was executing (a SIMD executes one VALU instruction at a time, so <= 1);
  * mfma_busy     = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8);
  * wave-cycle split: active (ACTIVE_INST_ANY) / issue-stalled (WAIT_INST_ANY) / parked on s_waitcnt or a barrier (WAIT_ANY),
    the three being disjoint and summing to ~ SQ_WAVE_CYCLES; valu_share_of_active = ACTIVE_INST_VALU / ACTIVE_INST_ANY;
  * valu_insts / mfma_insts per launch (wave-instructions).
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
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
        act, wc = c.get("GRBM_GUI_ACTIVE", 0.0), c.get("SQ_WAVE_CYCLES", 0.0)
        if act <= 0 or wc <= 0 or cnt[n] == 0:
            continue
        simd_cycles = 1024.0 * act / 8.0
        a_any = max(c.get("SQ_ACTIVE_INST_ANY", 0.0), 1.0)
        out[n] = {
            "launches": cnt[n],
            "kernel_cycles_per_launch": round(act / 8 / cnt[n]),
            "valu_busy": round(4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / simd_cycles, 4),
            "mfma_busy": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles, 4),
            "wave_cycles_active": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4),
            "wave_cycles_issue_stalled": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4),
            "wave_cycles_parked": round(c.get("SQ_WAIT_ANY", 0.0) / wc, 4),
            "valu_share_of_active": round(c.get("SQ_ACTIVE_INST_VALU", 0.0) / a_any, 4),
            "valu_insts_per_launch": round(c.get("SQ_INSTS_VALU", 0.0) / cnt[n]),
            "mfma_insts_per_launch": round(c.get("SQ_INSTS_MFMA", 0.0) / cnt[n]),
            "occupancy_waves_per_simd": round(4.0 * wc / simd_cycles, 3),
        }
    json.dump({"note": "per kernel, averaged over launches; see tools/pmc_valu.py for the definitions and units",
               "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["kernel_cycles_per_launch"] * kv[1]["launches"]))},
              sys.stdout, indent=1)


if __name__ == "__main__":
    main()
