#!/usr/bin/env python3
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd .db (collected with --pmc <COUNTER>)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute("select name, counter_name, count(*), avg(counter_value), min(counter_value), max(counter_value), avg(duration) "
                   "from pmc_events group by name, counter_name order by 4 desc").fetchall()
print(f"{'kernel':70s} {'counter':12s} {'calls':>7s} {'avg':>14s} {'min':>14s} {'max':>14s} {'avg_us':>9s}")
for n, c, k, a, mn, mx, d in rows:
    print(f"{n[:70]:70s} {c:12s} {k:7d} {a:14.2f} {mn:14.2f} {mx:14.2f} {d / 1e3:9.2f}")
