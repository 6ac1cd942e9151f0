"""Per-kernel summary of a rocprofv3 --kernel-trace run stored in rocpd (sqlite) format.
usage: python tools/kernel_stats_from_db.py <results.db> <steps_kernel_substring> [max_rows] > profiles/<name>.txt
The number of steps in the trace is taken as the launch count of the kernel whose name contains the given substring
(one launch per step, e.g. k_attn_h_bwd_dkv).  The header also gives launches per step, the share of launches shorter than
10 us, and the subtotal of kernels that are not this library's (torch element-wise / copy / fill / RNG / optimizer)."""
import sqlite3
import sys

db, marker = sys.argv[1], sys.argv[2]
max_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 400
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), avg(end-start)/1e3, sum(end-start)/1e6 from kernels group by name order by 4 desc").fetchall()
steps = next(r[1] for r in rows if marker in r[0])
total = sum(r[3] for r in rows)
launches = sum(r[1] for r in rows)
short = c.execute("select count(*), sum(end-start)/1e6 from kernels where end-start < 10000").fetchone()
foreign = [r for r in rows if r[0].startswith(("void at::", "at::", "__amd_rocclr", "void rocprim", "rocprim", "Cijk_", "void hipcub"))
           or "at::native" in r[0] or "Cijk_" in r[0]]
print(f"# total kernel time {total:.2f} ms over {steps} steps = {total / steps:.2f} ms/step ({db})")
print(f"# launches/step {launches / steps:.1f}; shorter than 10 us: {short[0] / steps:.1f} launches = {(short[1] or 0) / steps:.3f} ms/step; "
      f"not this library's (torch/rocclr/library): {sum(r[1] for r in foreign) / steps:.1f} launches = {sum(r[3] for r in foreign) / steps:.3f} ms/step")
# idle time between consecutive kernels over the LAST six steps (steady state: a replayed HIP graph in the default bench), the window
# running from the marker kernel's 7th-last launch to its last one
marks = [r[0] for r in c.execute("select start from kernels where name like ? order by start", (f"%{marker}%",)).fetchall()]
if len(marks) >= 7:
    t0, t1 = marks[-7], marks[-1]
    ks = c.execute("select start, end from kernels where start >= ? and start < ? order by start", (t0, t1)).fetchall()
    busy, idle, nidle, cur_end = 0, 0, 0, ks[0][0]
    for st, en in ks:
        if st > cur_end:
            idle += st - cur_end; nidle += 1
            busy += en - st
        else:
            busy += max(0, en - cur_end)
        cur_end = max(cur_end, en)
    print(f"# last 6 steps: wall {(t1 - t0) / 6e6:.3f} ms/step = GPU busy {busy / 6e6:.3f} + idle {idle / 6e6:.3f} ms/step over {nidle / 6:.0f} "
          f"gaps ({idle / max(1, nidle) / 1e3:.2f} us each); {len(ks) / 6:.0f} launches/step")
print(f"{'kernel':92s} {'calls':>6s} {'avg_us':>9s} {'ms/step':>8s} {'%':>6s}")
for name, calls, avg_us, ms in rows[:max_rows]:
    print(f"{name[:92]:92s} {calls:6d} {avg_us:9.1f} {ms / steps:8.3f} {100 * ms / total:6.2f}")

# optional 4th argument: a file that receives the launch sequence of the last step (start offset, duration, gap to the previous
# kernel's end, short name) -- what to read when hunting for chains of short launches worth fusing
if len(sys.argv) > 4 and len(marks) >= 2:
    import re
    t0, t1 = marks[-2], marks[-1]
    seq = c.execute("select start, end, name from kernels where start >= ? and start < ? order by start", (t0, t1)).fetchall()
    with open(sys.argv[4], "w") as f:
        prev_end = seq[0][0]
        for st, en, name in seq:
            short = re.sub(r"\(anonymous namespace\)::|void |_ZN12_GLOBAL__N_1\d+", "", name)[:70]
            f.write(f"{(st - t0) / 1e3:10.1f} {(en - st) / 1e3:8.1f} {(st - prev_end) / 1e3:7.1f}  {short}\n")
            prev_end = max(prev_end, en)
