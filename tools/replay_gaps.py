"""Idle time inside and between REPLAYED steps of a rocprofv3 kernel trace of the default bench (tools/profile_bench.sh keeps the
database when KEEP_DB=1): the window runs over the middle launches of the marker kernel (one per step), which are graph replays -- the
first steps prime / record and the last five are the eager steps that carry the HIP-event timers.
usage: python tools/replay_gaps.py <results.db> [marker substring]"""
import sqlite3, sys
db = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_attn_h_fwd"
c = sqlite3.connect(db)
marks = [r[0] for r in c.execute("select start from kernels where name like ? order by start", (f"%{marker}%",)).fetchall()]
lo, hi = len(marks) // 3, len(marks) - 7
t0, t1, n = marks[lo], marks[hi], hi - lo
ks = c.execute("select start, end, name from kernels where start >= ? and start < ? order by start", (t0, t1)).fetchall()
busy = idle = 0
gaps = []
cur = ks[0][0]
for st, en, name in ks:
    if st > cur:
        idle += st - cur
        gaps.append(((st - cur) / 1e3, name[:60]))
        busy += en - st
    else:
        busy += max(0, en - cur)
    cur = max(cur, en)
print(f"{n} replayed steps: wall {(t1 - t0) / n / 1e6:.3f} ms/step = busy {busy / n / 1e6:.3f} + idle {idle / n / 1e6:.3f} ms/step; "
      f"{len(ks) / n:.0f} launches/step, {len(gaps) / n:.0f} gaps/step, mean gap {idle / max(1, len(gaps)) / 1e3:.2f} us")
gaps.sort(reverse=True)
print("largest gaps (us, kernel that follows):")
for g, name in gaps[:12]:
    print(f"  {g:8.1f}  {name}")
import collections
by = collections.Counter()
for g, name in gaps:
    by[name] += g
print("idle by following kernel (us per step):")
for name, g in by.most_common(12):
    print(f"  {g / n:8.2f}  {name}")
