#!/usr/bin/env python3
"""Per-kernel summary (the `--stats` table) from a rocprofv3 rocpd database: tools/kernel_stats_from_db.py results.db > out.csv
Extra column UnionNs: the time during which AT LEAST ONE launch of the kernel was running.  For a kernel whose launches overlap
(k_fast: the small pyramid levels run on a second stream beside the large ones) TotalDurationNs counts the overlapped time once per
launch, UnionNs once; UnionNs / steps (steps = calls of k_quadtree, one per batched step) is the wall time of the stage per step --
the figure bench.py measures with HIP events."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
names = [r[0] for r in db.execute("select distinct name from kernels")]
rows = []
for n in names:
    iv = db.execute("select start, end from kernels where name = ? order by start", (n,)).fetchall()
    tot = sum(e - s for s, e in iv)
    union, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            union += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    union += ce - cs
    d = [e - s for s, e in iv]
    rows.append((n, len(iv), tot, tot / len(iv), min(d), max(d), union))
rows.sort(key=lambda r: -r[2])
total = sum(r[2] for r in rows) or 1
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","UnionNs"')
for n, c, t, a, mn, mx, u in rows:
    print(f'"{n}",{c},{t},{a:.3f},{100.0 * t / total:.2f},{mn},{mx},{u}')

# how much of the run the GPU had at least one kernel in flight (all kernels, all streams): wall span of the trace vs union of the
# kernel intervals -- the difference is launch / dependency gaps, not kernel time
iv = db.execute("select start, end from kernels order by start").fetchall()
if iv:
    union, cs, ce = 0, iv[0][0], iv[0][1]
    for s_, e_ in iv[1:]:
        if s_ > ce:
            union += ce - cs
            cs, ce = s_, e_
        else:
            ce = max(ce, e_)
    union += ce - cs
    span = max(e_ for _, e_ in iv) - iv[0][0]
    print(f'"# all kernels: union busy ns {union}, trace span ns {span}, idle fraction inside the span {1 - union / span:.4f}"')

# template instances of one kernel (k_fast<40, 36>, k_fast<48, 36>: one per LDS carve-up) as ONE stage: the union over all of them
import re
groups = {}
for n in names:
    m = re.search(r"(\w+)<[^>]*>\(", n)
    if m:
        groups.setdefault(m.group(1), []).append(n)
for base, members in groups.items():
    if len(members) < 2:
        continue
    q = ",".join("?" * len(members))
    iv = db.execute(f"select start, end from kernels where name in ({q}) order by start", members).fetchall()
    union, cs, ce = 0, iv[0][0], iv[0][1]
    for s_, e_ in iv[1:]:
        if s_ > ce:
            union += ce - cs
            cs, ce = s_, e_
        else:
            ce = max(ce, e_)
    union += ce - cs
    steps = max((r[1] for r in rows if "k_brief" in r[0]), default=0)
    per = f", {union / steps / 1e6:.4f} ms per step ({steps} steps = calls of k_brief)" if steps else ""
    print(f'"# {base}: {len(members)} template instances, {len(iv)} launches, union ns {union}{per}"')
