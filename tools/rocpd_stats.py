#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max duration) from a rocprofv3 rocpd .db (kernel trace)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   f"from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"{'kernel':90s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
for n, c, s, a, mn, mx in rows:
    print(f"{n[:90]:90s} {c:8d} {s / 1e6:10.3f} {a / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * s / tot:6.2f}")
