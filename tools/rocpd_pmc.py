#!/usr/bin/env python3
"""Per-kernel average of the PMC counters of a rocprofv3 --pmc run (rocpd sqlite): rocpd_pmc.py results.db [top]"""
import re
import sqlite3
import sys


def main(path, top=12):
    db = sqlite3.connect(path)
    cur = db.cursor()
    info = cur.execute("select name, description, expression from pmc_info").fetchall()
    for n, d, e in info:
        print(f"# counter {n}: {d} [{e}]")
    rows = cur.execute("select name, counter_name, counter_value, duration from pmc_events").fetchall()
    agg = {}
    for name, cn, v, dur in rows:
        k = (re.sub(r"\(.*", "", name).replace("void ", ""), cn)
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += v
        a[2] += (dur or 0)
    print(f"{'kernel':64s} {'counter':14s} {'calls':>6s} {'avg_value':>14s} {'avg_us':>9s}")
    for (k, cn), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{k[:64]:64s} {cn:14s} {a[0]:6d} {a[1]/a[0]:14.1f} {a[2]/a[0]/1e3:9.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12)
