#!/usr/bin/env python3
"""roctx ranges of a rocprofv3 (rocpd sqlite) marker trace (VRA_ROCTX=1: the engine brackets every step): message, count, total / avg ms.
    VRA_ROCTX=1 rocprofv3 --marker-trace --kernel-trace -d out -- python tools/prefill_once.py 1024;  python tools/rocpd_regions.py out/*/*.db"""
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(regions)")]
rows = cur.execute("select * from regions").fetchall()
i_start, i_end = cols.index("start"), cols.index("end")
agg = {}
for r in rows:
    msg = None
    for v in r:  # the message travels in a JSON column
        if isinstance(v, str) and v.startswith("{") and '"message"' in v:
            msg = json.loads(v).get("message")
            break
    msg = msg or str(r[cols.index("name")] if "name" in cols else r[3])
    a = agg.setdefault(msg, [0, 0.0])
    a[0] += 1
    a[1] += (r[i_end] - r[i_start]) / 1e6
print(f"{'range':60s} {'count':>6s} {'total_ms':>10s} {'avg_ms':>9s}")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{k[:60]:60s} {a[0]:6d} {a[1]:10.3f} {a[1] / a[0]:9.3f}")
