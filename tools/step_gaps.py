#!/usr/bin/env python3
"""Where a graph-replayed learner step spends its time: kernel time vs the gaps between kernels.

    rocprofv3 --kernel-trace -d /tmp/gaps -o g -- python bench.py --steps 300 --warmup 50
    python tools/step_gaps.py /tmp/gaps/**/g_results.db

Takes the last 2000 kernel dispatches, splits them into steps at the first kernel of the step graph
(the most frequent kernel name that starts a repeating sequence) and prints the mean duration of each
kernel, the mean gap in front of it, and the totals per step."""
import glob
import sqlite3
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    dbs = glob.glob(path, recursive=True) if any(c in path for c in "*?") else [path]
    con = sqlite3.connect(dbs[0])
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    rows = rows[-2000:]
    names = [r[0] for r in rows]
    first = next((k for k in ("cnn_fwd_stack_x3_kernel", "cnn_fwd_stack_kernel") if any(k in n for n in names)), names[0])
    idx = [i for i, n in enumerate(names) if first in n]
    dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
    steps, total = 0, 0.0
    order = []
    for a, b in zip(idx[:-1], idx[1:]):
        steps += 1
        total += rows[b][1] - rows[a][1]
        for k in range(a, b):
            key = (k - a, rows[k][0].split("(")[0][-60:])
            if steps == 1:
                order.append(key)
            dur[key] += rows[k][2] - rows[k][1]
            gap[key] += rows[k][1] - rows[k - 1][2] if k > 0 else 0.0
            cnt[key] += 1
    print("%d steps, %.2f us per step" % (steps, total / steps / 1e3))
    ksum = gsum = 0.0
    for key in order:
        n = max(cnt[key], 1)
        print("  %-62s %7.2f us  (gap before: %5.2f us)" % (key[1], dur[key] / n / 1e3, gap[key] / n / 1e3))
        ksum += dur[key] / n / 1e3
        gsum += gap[key] / n / 1e3
    print("kernels %.2f us + gaps %.2f us per step" % (ksum, gsum))


if __name__ == "__main__":
    main()
