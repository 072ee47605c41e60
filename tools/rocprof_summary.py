#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite output) results .db into the `--stats`-style
per-kernel summary we commit under profiles/.  Usage: rocprof_summary.py results.db [out.md]"""
import sqlite3
import sys


def short(name, n=110):
    name = name.replace("void ", "")
    if "(" in name and len(name) > n:
        name = name[: name.index("(")]
    return name[:n]


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    out.write("| kernel | calls | total_us | avg_us | % |\n|---|---|---|---|---|\n")
    for name, calls, tot, avg, pct in rows:
        if pct < 0.02:
            continue
        out.write("| `%s` | %d | %.1f | %.2f | %.2f |\n" % (short(name), calls, tot, avg, pct))
    try:
        pmc = list(c.execute("select name, counter_name, avg(value), count(*) from counters_collection "
                             "group by name, counter_name"))
        if pmc:
            out.write("\n| kernel | counter | avg value per dispatch | dispatches |\n|---|---|---|---|\n")
            for name, cn, v, n in pmc:
                out.write("| `%s` | %s | %.6g | %d |\n" % (short(name), cn, v, n))
    except sqlite3.Error:
        pass


if __name__ == "__main__":
    main()
