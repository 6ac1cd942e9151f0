"""Per-kernel summary of a rocprofv3 --kernel-trace run stored in rocpd (sqlite) format.
usage: python tools/kernel_stats_from_db.py <results.db> <steps_kernel_substring> > profiles/<name>.txt
The number of steps in the trace is taken as the launch count of the kernel whose name contains the given substring
(one launch per step, e.g. k_attn_h_bwd_dkv)."""
import sqlite3
import sys

db, marker = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), avg(end-start)/1e3, sum(end-start)/1e6 from kernels group by name order by 4 desc").fetchall()
steps = next(r[1] for r in rows if marker in r[0])
total = sum(r[3] for r in rows)
print(f"# total kernel time {total:.2f} ms over {steps} steps = {total / steps:.2f} ms/step ({db})")
print(f"{'kernel':92s} {'calls':>6s} {'avg_us':>9s} {'ms/step':>8s} {'%':>6s}")
for name, calls, avg_us, ms in rows[:45]:
    print(f"{name[:92]:92s} {calls:6d} {avg_us:9.1f} {ms / steps:8.3f} {100 * ms / total:6.2f}")
