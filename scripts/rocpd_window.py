#!/usr/bin/env python3
"""A window of consecutive kernel dispatches of a rocprofv3 (rocpd SQLite) trace, in time order, each with its duration
and the idle gap since the previous dispatch ended -- what a launch-bound phase looks like from the device (the per-batch
fixed cost of the column-sharded schedule, profiles/r06_rank_work.md).
Usage: rocpd_window.py results.db <anchor kernel substring> [occurrence=-1] [before=150] [after=5]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2]
occ = int(sys.argv[3]) if len(sys.argv) > 3 else -1
before = int(sys.argv[4]) if len(sys.argv) > 4 else 150
after = int(sys.argv[5]) if len(sys.argv) > 5 else 5
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
start, end = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
rows = db.execute(f"select {name_col}, {start}, {end} from kernels order by {start}").fetchall()
hits = [k for k, r in enumerate(rows) if anchor in r[0]]
if not hits:
    sys.exit(f"no dispatch matches {anchor!r}")
at = hits[occ]
lo, hi = max(0, at - before), min(len(rows), at + after + 1)
print("| # | kernel | us | gap before, us |")
print("|---|---|---|---|")
busy = idle = 0.0
for k in range(lo, hi):
    n, s, e = rows[k]
    gap = (s - rows[k - 1][2]) / 1e3 if k > 0 else 0.0
    busy += (e - s) / 1e3
    idle += max(gap, 0.0) if k > lo else 0.0
    print(f"| {k - at:+d} | `{n[:70]}` | {(e - s) / 1e3:.1f} | {gap:.1f} |")
print(f"\nwindow: {hi - lo} dispatches, {busy / 1e3:.3f} ms busy, {idle / 1e3:.3f} ms idle between them")
