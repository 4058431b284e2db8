#!/usr/bin/env python3
"""Per-dispatch durations of the kernels whose name contains a substring (rocprofv3 rocpd SQLite trace).
Usage: rocpd_dispatches.py results.db substring"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
start, end = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
extra = [c for c in ("grid_size", "grid_size_x", "workgroup_size", "lds_size", "vgpr_count", "sgpr_count", "scratch_size") if c in cols]
q = f"select {name_col}, {start}, {end}" + "".join(f", {c}" for c in extra) + f" from kernels order by {start}"
print("| # | kernel | ms | " + " | ".join(extra) + " |")
print("|---|---|---|" + "---|" * len(extra))
k = 0
for row in db.execute(q):
    if sys.argv[2] not in row[0]:
        continue
    k += 1
    print(f"| {k} | `{row[0][:60]}` | {(row[2] - row[1]) / 1e6:.3f} | " + " | ".join(str(x) for x in row[3:]) + " |")
