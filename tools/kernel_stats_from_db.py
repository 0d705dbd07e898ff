#!/usr/bin/env python3
"""Per-kernel summary (the `--stats` table) from a rocprofv3 rocpd database: tools/kernel_stats_from_db.py results.db > out.csv"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name "
                  "order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for n, c, t, a, mn, mx in rows:
    print(f'"{n}",{c},{t},{a:.3f},{100.0 * t / tot:.2f},{mn},{mx}')
