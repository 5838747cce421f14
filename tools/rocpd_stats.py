#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace): per kernel (name, grid) calls / avg / total.
usage: rocpd_stats.py results.db [out.csv] [--by-grid]"""
import re, sqlite3, sys
db = sys.argv[1]
out = next((a for a in sys.argv[2:] if not a.startswith("--")), None)
by_grid = "--by-grid" in sys.argv
c = sqlite3.connect(db)
rows = c.execute("select name, grid_x, grid_y, workgroup_x, duration, vgpr_count, accum_vgpr_count, lds_size, scratch_size from kernels").fetchall()
agg = {}
for name, gx, gy, wx, dur, vg, ag, lds, scr in rows:
    short = re.sub(r"^void ", "", name)
    short = re.sub(r"\(.*$", "", short)
    key = (short, gx // max(wx, 1), gy, wx) if by_grid else (short,)
    a = agg.setdefault(key, [0, 0, vg, ag, lds, scr])
    a[0] += 1; a[1] += dur
tot = sum(a[1] for a in agg.values())
lines = ["kernel,grid,calls,total_ms,avg_us,pct,vgpr,agpr,lds,scratch"]
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    grid = "x".join(str(v) for v in key[1:]) if by_grid else ""
    lines.append(f"\"{key[0]}\",{grid},{a[0]},{a[1]/1e6:.3f},{a[1]/a[0]/1e3:.1f},{100*a[1]/tot:.2f},{a[2]},{a[3]},{a[4]},{a[5]}")
txt = "\n".join(lines)
if out:
    open(out, "w").write(txt + "\n")
print(txt)
