#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv (per launch).  Usage: pmc_kernels.py <csv> [kernel-prefix]"""
import collections, csv, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(collections.Counter)
pre = sys.argv[2] if len(sys.argv) > 2 else "k_"
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
    if not k.startswith(pre):
        continue
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k][r["Counter_Name"]] += 1
for k in tot:
    print(k, n[k].most_common(1)[0][1], {c: round(v / n[k][c]) for c, v in sorted(tot[k].items())})
