#!/usr/bin/env python3
"""fetch_calib_summary.py <fetch counter csv> <write counter csv> <stdout of fetch_calib> <out.json>: counter bytes / true bytes per access shape."""
import collections, csv, json, sys


def collect(path, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].split("::")[-1]
        tot[name] += float(r["Counter_Value"])
        n[name] += 1
    return {k: tot[k] / n[k] * 1024.0 for k in tot}   # KiB -> bytes per launch


true = {}
for line in open(sys.argv[3]):
    f = line.split()
    if len(f) == 3 and f[0] == "TRUE":
        true[f[1]] = int(f[2])
fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"source": "tools/exp/fetch_calib.hip under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), 1 GiB buffer, MI355X",
       "note": "ratio = counter bytes / bytes the kernel really moved; multiply a kernel's raw FETCH_SIZE by 1 / ratio of its load shape",
       "loads": {}, "stores": {}}
for k, t in true.items():
    if "store" in k:
        out["stores"][k] = {"true_bytes": t, "WRITE_SIZE_bytes": write.get(k), "ratio": (write.get(k, 0) / t) if t else None}
    else:
        out["loads"][k] = {"true_bytes": t, "FETCH_SIZE_bytes": fetch.get(k), "ratio": (fetch.get(k, 0) / t) if t else None}
json.dump(out, open(sys.argv[4], "w"), indent=1, sort_keys=True)
for grp in ("loads", "stores"):
    for k, v in out[grp].items():
        print(f"{k:34s} ratio {v['ratio']:.3f}")
