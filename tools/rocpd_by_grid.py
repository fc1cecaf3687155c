#!/usr/bin/env python3
"""Per (kernel, grid size) dispatch durations of a rocprofv3 rocpd database: tells the launches of one multi-purpose
kernel apart (the fp32 MLP job-list kernel runs six differently shaped launches per step).
    python tools/rocpd_by_grid.py <results.db> [name-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
q = ("select s.kernel_name, d.grid_size_x / d.workgroup_size_x, count(*), avg(d.end - d.start) / 1e3, "
     "min(d.end - d.start) / 1e3 from %s d join %s s on d.kernel_id = s.id where s.kernel_name like ? "
     "group by s.kernel_name, d.grid_size_x order by s.kernel_name, d.grid_size_x" % (kd, ks))
print("%-60s %8s %8s %9s %9s" % ("kernel", "wgs", "calls", "avg_us", "min_us"))
for name, wgs, n, avg, mn in db.execute(q, ("%" + pat + "%",)):
    if n >= 20:
        print("%-60s %8d %8d %9.2f %9.2f" % (name[:60], wgs, n, avg, mn))
