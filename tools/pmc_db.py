"""Per-kernel means of the counters in a rocprofv3 --pmc results database (sqlite).
    python tools/pmc_db.py gpurun_out/mlp/pmc1/p_results.db [kernel-substring]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = db.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id), avg(duration) "
                      "from counters_collection group by kernel_name, counter_name").fetchall()
    by = {}
    for k, c, v, n, d in rows:
        if pat in k:
            by.setdefault(k, {})[c] = (v / max(n, 1), n, d)
    for k, cs in by.items():
        n = max(x[1] for x in cs.values())
        print("%s  (%d dispatches)" % (k[:90], n))
        for c in sorted(cs):
            print("    %-28s %14.1f" % (c, cs[c][0]))


if __name__ == "__main__":
    main()
