#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd SQLite) kernel trace: per-kernel calls / total / avg / min / max.
Usage: rocpd_summary.py results.db [out.md]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
start, end = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
rows = db.execute(
    f"select {name_col}, count(*), sum({end}-{start}), avg({end}-{start}), min({end}-{start}), max({end}-{start}) "
    f"from kernels group by {name_col} order by 3 desc").fetchall()
total = sum(r[2] for r in rows) or 1
lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
for n, c, t, a, mn, mx in rows:
    lines.append(f"| `{n[:110]}` | {c} | {t/1e6:.3f} | {a/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*t/total:.2f} |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
