#!/usr/bin/env python3
"""Per-kernel PMC counter totals from a rocprofv3 rocpd SQLite database.
Usage: pmc_summary.py results.db [kernel-name-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
filt = sys.argv[2] if len(sys.argv) > 2 else ""


def cols(t):
    return [r[1] for r in db.execute(f"pragma table_info({t})")]


view = "counters_collection"
c = cols(view)
name = "kernel_name" if "kernel_name" in c else ("name" if "name" in c else None)
cname = "counter_name" if "counter_name" in c else None
val = "value" if "value" in c else ("counter_value" if "counter_value" in c else None)
if not (name and cname and val):
    print("columns:", c)
    sys.exit(1)
disp = "dispatch_id" if "dispatch_id" in c else None
q = f"select {name}, {cname}, count(distinct {disp or 'rowid'}), sum({val}) from {view} group by 1, 2 order by 4 desc"
print("| kernel | counter | dispatches | sum | per dispatch |")
print("|---|---|---|---|---|")
for k, cn, n, s in db.execute(q):
    if filt and filt not in k:
        continue
    print(f"| `{k[:80]}` | {cn} | {n} | {s:.6g} | {s / max(n, 1):.6g} |")

# kernel durations (ns) for clock estimates
try:
    kc = cols("kernels")
    nm = "name" if "name" in kc else "kernel_name"
    st, en = ("start", "end") if "start" in kc else ("start_timestamp", "end_timestamp")
    print()
    print("| kernel | dispatches | total ms |")
    print("|---|---|---|")
    for k, n, t in db.execute(f"select {nm}, count(*), sum({en}-{st}) from kernels group by 1 order by 3 desc"):
        if filt and filt not in k:
            continue
        print(f"| `{k[:80]}` | {n} | {t / 1e6:.3f} |")
except Exception as e:  # noqa
    print("no kernel table:", e)
