#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (``*_results.db``) into a short text report.

    python tools/rocpd_summary.py gpurun_out/prof2/b_results.db [--steps 330] [--top 30] > profiles/x.txt

rocprofv3 on ROCm 7.2 writes SQLite by default; this prints the per-kernel table
(calls, total/avg/min/max microseconds, share, registers, LDS, grid) so that a
profile can be committed as text under profiles/.  With --steps the per-step
budget of every kernel is shown too (total / steps), which is the number to read
for a graph-replayed learner step.  PMC counter runs (``--pmc``) are summarised
per kernel and counter when the database holds any.
"""
import argparse
import re
import sqlite3
import sys


def short(name, width=96):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    if name.startswith("at::native") or "at::native::" in name[:60]:
        m = re.search(r"at::native::(?:\w+::)*(\w+)", name)
        inner = re.findall(r"(\w+_kernel\w*|\w+Functor\w*|index_put\w*|direct_copy\w*)", name)
        name = "torch::" + (m.group(1) if m else "op") + ("<" + inner[1] + ">" if len(inner) > 1 else "")
    else:
        name = re.sub(r"\(.*\)$", "", name)      # drop the argument list of our own kernels
    return name if len(name) <= width else name[:width - 3] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--steps", type=int, default=0, help="learner steps covered by the trace (per-step column)")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    con = sqlite3.connect(a.db)
    cur = con.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), "
        "max(grid_x), max(grid_y), max(grid_z), max(workgroup_x) from kernels group by name "
        "order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("# %s" % a.db)
    print("# kernels: %d distinct, %d dispatches, %.1f us of GPU kernel time" % (
        len(rows), sum(r[1] for r in rows), total / 1e3))
    hdr = "%-96s %7s %10s %8s %8s %8s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "share")
    if a.steps:
        hdr += " %9s" % "us/step"
    hdr += "  vgpr agpr sgpr   lds  grid(threads) x wg"
    print(hdr)
    for r in rows[:a.top]:
        name, calls, tot, avg, mn, mx, vg, ag, sg, lds, gx, gy, gz, wx = r
        line = "%-96s %7d %10.1f %8.2f %8.2f %8.2f %5.1f%%" % (
            short(name), calls, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total)
        if a.steps:
            line += " %9.2f" % (tot / 1e3 / a.steps)
        line += "  %4d %4d %4d %5d  %dx%dx%d / %d" % (vg or 0, ag or 0, sg or 0, lds or 0, gx or 0, gy or 0, gz or 0, wx or 0)
        print(line)
    if a.steps:
        print("# sum over all kernels: %.2f us/step" % (total / 1e3 / a.steps))
    # PMC counters, if any
    try:
        pmc = cur.execute(
            "select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
            "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
    except sqlite3.Error:
        pmc = []
    if pmc:
        print("\n# PMC counters (per kernel: dispatches, sum, mean per dispatch)")
        for name, cname, n, s, avg in pmc:
            print("%-80s %-28s %7d %16.0f %14.1f" % (short(name, 80), cname, n, s, avg))
    return 0


if __name__ == "__main__":
    sys.exit(main())
