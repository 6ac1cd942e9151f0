#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass).

    (cd /tmp && export TMPDIR=/tmp && \
     rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline && \
     rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline)
    python tools/pmc_traffic.py $OUT/fetch $OUT/write > profiles/rNN_pmc_traffic.json

Units and gfx950 corrections as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE / WRITE_SIZE are
kilobytes; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B/lane) coalesced reads at 64 bytes, so
it is doubled; WRITE_SIZE is exact for 16 B/lane stores.  Other access widths are uncalibrated (the SpMM gather,
whose algorithmic byte count is known, is the calibration row).
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short_name(raw):
    """`_ZN12_GLOBAL__N_1<len><name>I<template args>E...` -> name<args>; anything else: cut at the argument list."""
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", raw)
    if m:
        n = int(m.group(1))
        rest = raw[m.end():]
        name, tail = rest[:n], rest[n:]
        t = re.match(r"I((?:L[ib]\d+E)+)E", tail)
        if t:
            name += "<" + ",".join(re.findall(r"L[ib](\d+)E", t.group(1))) + ">"
        return name
    raw = raw.replace("void ", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", raw)[:100]


def per_kernel(dirname, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for path in glob.glob(f"{dirname}/**/*counter_collection.csv", recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                name = short_name(row["Kernel_Name"])
                a = acc[name]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return acc


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for name in sorted(set(fetch) | set(write)):
        nf, f = fetch.get(name, (0, 0.0))
        nw, w = write.get(name, (0, 0.0))
        rd = 2.0 * f * 1024 / max(nf, 1)   # gfx950 correction: x2
        wr = w * 1024 / max(nw, 1)
        out[name] = {"launches": max(nf, nw), "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                     "fabric_bytes_per_launch": round(rd + wr)}
    json.dump({"unit": "bytes per launch (FETCH_SIZE KB x2 gfx950 correction + WRITE_SIZE KB)", "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
