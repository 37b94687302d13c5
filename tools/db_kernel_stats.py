#!/usr/bin/env python3
"""Reduce a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- python3 bench.py ...` writes
DIR/NAME_results.db on ROCm 7.2) to the per-kernel summary that `--output-format csv` calls kernel_stats.csv:
Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs.   usage: db_kernel_stats.py results.db out.csv [steps]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end - start), min(end - start), max(end - start) from kernels group by name").fetchall()
tot = sum(r[2] for r in rows) or 1
rows.sort(key=lambda r: -r[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else None
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, calls, total, mn, mx in rows:
        w.writerow([name, calls, total, round(total / calls, 1), round(100.0 * total / tot, 3), mn, mx])
print(f"{len(rows)} kernels, {tot / 1e6:.2f} ms of kernel time" + (f" = {tot / 1e6 / steps:.2f} ms per step over {steps} steps (warm-up included)" if steps else ""))
for name, calls, total, mn, mx in rows[:22]:
    print(f"{100.0 * total / tot:6.2f} %  {total / 1e6:9.2f} ms  {calls:6d} x {total / calls / 1e3:9.1f} us  {name[:110]}")
