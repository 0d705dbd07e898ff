#!/usr/bin/env python3
"""Per-wave SQ counter summary of a rocprofv3 --pmc pass -> profiles/r1_sq_counters.json layout.
Usage: sq_summary.py <counter_collection.csv> <out.json> "<source note>"
Counters expected: SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
(one pass; SQ_*_CYCLES are in units of 4 clocks).  A "step" = one launch of k_quadtree (one per batched step)."""
import collections, csv, json, sys

tot = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
    if not k.startswith("k_"):
        continue
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES":
        n[k] += 1
steps = max(n["k_quadtree"], 1)
out = {"source": sys.argv[3], "pairs_per_step": int(sys.argv[4]) if len(sys.argv) > 4 else None, "kernels": {}}
for k, c in sorted(tot.items()):
    w = max(c["SQ_WAVES"], 1.0)
    wc = max(c["SQ_WAVE_CYCLES"], 1.0)
    out["kernels"][k] = {
        "launches_per_step": round(n[k] / steps, 2), "waves_per_step": c["SQ_WAVES"] / steps,
        "valu_per_wave": round(c["SQ_INSTS_VALU"] / w, 1), "salu_per_wave": round(c["SQ_INSTS_SALU"] / w, 1),
        "lds_per_wave": round(c["SQ_INSTS_LDS"] / w, 1), "wave_quadcycles_per_wave": round(wc / w),
        "wait_any_frac": round(c["SQ_WAIT_ANY"] / wc, 3), "wait_inst_frac": round(c["SQ_WAIT_INST_ANY"] / wc, 3),
        "issue_active_frac": round(c["SQ_ACTIVE_INST_ANY"] / wc, 3)}
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
for k, v in out["kernels"].items():
    print(k, v)
