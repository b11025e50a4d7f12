#!/usr/bin/env python3
"""Per-API-call statistics from a rocprofv3 --hip-trace rocpd database (run where the .db is)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
cand = [n for n in names if n in ("regions", "regions_and_samples", "top")]
print("views:", [n for n in names if not n.startswith("rocpd_")][:40])
for v in ("regions",):
    if v not in names:
        continue
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % v)]
    print(v, cols)
    q = "select name, count(*), avg(end-start)/1000.0, sum(end-start)/1e6 from %s group by name order by 4 desc limit 25" % v
    for r in cur.execute(q):
        print("%-40s calls %8d  avg %9.2f us  total %9.2f ms" % (r[0][:40], r[1], r[2], r[3]))
