#!/usr/bin/env python3
"""LDS accounting per kernel from one rocprofv3 --pmc pass:
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS \
        SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -- python3 bench.py ...
    python tools/pmc_lds.py $OUT > profiles/rNN_pmc_lds.json
Per kernel, averaged over launches: LDS-array cycles (SQ_LDS_IDX_ACTIVE) and the extra cycles bank conflicts cost
(SQ_LDS_BANK_CONFLICT), both as a fraction of CU-cycles (256 CUs x kernel cycles: one LDS per CU); the share of wave quad-cycles
stalled at LDS issue (SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES); LDS wave-instructions per launch."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import short_name

acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
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
    cu_cycles = 256.0 * act / 8.0
    out[n] = {"launches": cnt[n], "kernel_cycles_per_launch": round(act / 8 / cnt[n]),
              "lds_array_busy": round(c.get("SQ_LDS_IDX_ACTIVE", 0.0) / cu_cycles, 4),
              "lds_bank_conflict_share_of_cu_cycles": round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / cu_cycles, 4),
              "lds_bank_conflict_share_of_lds_cycles": round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0), 4),
              "wave_cycles_stalled_at_lds_issue": round(c.get("SQ_WAIT_INST_LDS", 0.0) / wc, 4),
              "wave_cycles_lds_inst_active": round(c.get("SQ_ACTIVE_INST_LDS", 0.0) / wc, 4),
              "lds_unaligned_stall": round(c.get("SQ_LDS_UNALIGNED_STALL", 0.0) / cnt[n]),
              "lds_insts_per_launch": round(c.get("SQ_INSTS_LDS", 0.0) / cnt[n]),
              "salu_insts_per_launch": round(c.get("SQ_INSTS_SALU", 0.0) / cnt[n])}
json.dump({"note": "see tools/pmc_lds.py", "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["kernel_cycles_per_launch"] * kv[1]["launches"])[:12])},
          sys.stdout, indent=1)
