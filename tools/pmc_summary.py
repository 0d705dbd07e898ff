#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/pmc_traffic.json.
Usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <pairs_per_step> <out.json> <images_per_launch>
FETCH_SIZE / WRITE_SIZE are in KiB (x1024).  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports exactly half of the bytes of
a wide coalesced read stream, so reads are doubled ("corrected"); both raw and corrected figures are kept."""
import collections, csv, json, sys

STAGE = {"k_resize": "resize", "k_resize_fused": "resize", "k_resize_regions": "resize", "k_rowtable": "stereo", "k_kplist": "orient_brief", "k_blur": "blur", "k_fast": "fast", "k_quadtree": "quadtree", "k_ic_moments": "orient_brief",
         "k_orient": "orient_brief", "k_brief": "orient_brief", "k_stereo": "stereo", "k_load_level0": "load_level0"}


def collect(path, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]  # k_fast<48, 40> -> k_fast
        tot[name] += float(r["Counter_Value"])
        n[name] += 1
    return tot, n


def main():
    fetch, nf = collect(sys.argv[1], "FETCH_SIZE")
    write, nw = collect(sys.argv[2], "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 bench.py`, see profiles/",
           "pairs_per_step": int(sys.argv[3]), "images_per_launch": int(sys.argv[5]), "kernels": {}, "per_kernel": {}}
    stage = collections.defaultdict(lambda: [0.0, 0.0, 0])
    # a "launch" of a stage = one batched step (k_fast runs once per pyramid level, k_load_level0 once per eye): average per step
    steps_f, steps_w = max(nf["k_quadtree"], 1), max(nw["k_quadtree"], 1)
    for k in sorted(set(fetch) | set(write)):
        f = fetch[k] / steps_f * 1024.0
        w = write[k] / steps_w * 1024.0
        out["per_kernel"][k] = {"launches_per_step": nf[k] / steps_f, "fetch_bytes_raw": f, "fetch_bytes_corrected": 2 * f, "write_bytes": w}
        if k in STAGE:
            stage[STAGE[k]][0] += 2 * f
            stage[STAGE[k]][1] += w
    for s, (f, w, _) in stage.items():
        out["kernels"][s] = {"hbm_bytes_per_launch": f + w, "read_bytes_corrected": f, "write_bytes": w}
    json.dump(out, open(sys.argv[4], "w"), indent=1, sort_keys=True)
    for s, v in out["kernels"].items():
        print(f"{s:14s} read {v['read_bytes_corrected'] / 1e6:9.1f} MB  write {v['write_bytes'] / 1e6:9.1f} MB per launch")


if __name__ == "__main__":
    main()
