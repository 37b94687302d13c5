#!/usr/bin/env python3
"""Timeline summary of the LAST complete training step in a rocprofv3 rocpd database (kernel-trace): wall time between two adam_kernel
launches, union of kernel intervals (GPU busy), idle time, time with >= 2 kernels in flight, per-stream busy time, the largest gaps.
usage: db_timeline.py results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start, end, name, stream_id from kernels order by start").fetchall()
adam = [i for i, r in enumerate(rows) if r[2].startswith("adam_kernel")]
a, b = adam[-3], adam[-2]
seg = rows[a + 1:b + 1]
t0, t1 = seg[0][0], seg[-1][1]
busy, cs, ce = 0, None, None
gaps = []
for s, e, n, st in seg:
    if ce is None or s > ce:
        if ce is not None:
            busy += ce - cs
            gaps.append((s - ce, n))
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
ev = sorted([(s, 1) for s, e, n, st in seg] + [(e, -1) for s, e, n, st in seg])
c, last, ov = 0, None, 0
for t, d in ev:
    if c >= 2:
        ov += t - last
    c += d; last = t
per = {}
for s, e, n, st in seg:
    per[st] = per.get(st, 0) + e - s
print(f"step wall {(t1 - t0) / 1e6:.2f} ms, GPU busy (union) {busy / 1e6:.2f}, idle {(t1 - t0 - busy) / 1e6:.2f}, >= 2 kernels in flight {ov / 1e6:.2f}, launches {len(seg)}")
print("per-stream kernel time (ms):", {k: round(v / 1e6, 2) for k, v in per.items()})
for g, n in sorted(gaps, reverse=True)[:8]:
    print(f"  gap {g / 1e3:8.1f} us before {n.split('(')[0][:80]}")
