#!/usr/bin/env python3
"""Per-kernel averages of the counters in a rocprofv3 --pmc rocpd database. usage: pmc_summary.py db [substr]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
    print([t for t in tabs if "pmc" in t.lower() or "counter" in t.lower()]); sys.exit(1)
cols = [d[0] for d in c.execute(f"select * from {view} limit 1").description]
kn = "kernel_name" if "kernel_name" in cols else "name"
rows = c.execute(f"select {kn}, counter_name, sum(value), count(distinct dispatch_id) from {view} group by {kn}, counter_name").fetchall()
for name, ctr, tot, n in sorted(rows):
    if sub in name:
        print(f"{name[:60]:60s} {ctr:32s} {tot / max(n, 1):16.1f}  x{n}")
