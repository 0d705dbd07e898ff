#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/pmc_traffic.json.
Usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <pairs_per_step> <out.json> <images_per_launch> [calibration.json]
FETCH_SIZE / WRITE_SIZE are in KiB (x1024).  The read counter is turned into bytes with the factor MEASURED for the load shape of each
kernel (profiles/r3_fetch_calibration.json, made by tools/exp/fetch_calib.sh: FETCH_SIZE / true bytes is 0.50 - 0.56 for every
streaming shape the pipeline uses -- aligned and byte-aligned dwords, aligned / 4-byte-aligned / byte-aligned 16-byte loads -- i.e.
the counter tallies 128-byte memory-side requests as 64 bytes; writes read 1.00).  The result is MEMORY-SIDE REQUEST bytes: Infinity-
Cache hits are counted (MI355X_MICROARCH.md), so it bounds the HBM bytes from above -- a kernel whose lines are re-fetched by several
XCDs' L2s (the keypoint windows of k_ic_moments / k_brief) shows more `traffic` than HBM could deliver in its run time."""
import collections, csv, json, sys

# load shape of each kernel -> entry of the calibration file
SHAPE = {"k_fast": "k_cal_dwordx4_4aligned_48", "k_blur": "k_cal_dword_aligned", "k_blur_mfma": "k_cal_dword_aligned", "k_resize_regions": "k_cal_dwordx4_byte_aligned",
         "k_resize": "k_cal_dwordx4_aligned", "k_load_level0": "k_cal_dwordx4_byte_aligned", "k_ic_moments": "k_cal_dword_byte_aligned",
         "k_brief": "k_cal_dword_aligned", "k_stereo": "k_cal_dword_aligned", "k_quadtree": "k_cal_dword_aligned", "k_kplist": "k_cal_dword_aligned",
         "k_orient": "k_cal_dword_aligned", "k_rowtable": "k_cal_dword_aligned", "k_stereo_rows": "k_cal_dwordx4_aligned", "k_stereo_sad": "k_cal_dword_aligned",
         "k_stereo4": "k_cal_dwordx4_aligned"}
STAGE = {"k_resize": "resize", "k_resize_fused": "resize", "k_resize_regions": "resize", "k_rowtable": "stereo", "k_kplist": "orient_brief", "k_blur": "blur", "k_blur_mfma": "blur", "k_fast": "fast", "k_quadtree": "quadtree", "k_ic_moments": "orient_brief",
         "k_orient": "orient_brief", "k_brief": "orient_brief", "k_stereo": "stereo", "k_stereo_rows": "stereo", "k_stereo_sad": "stereo", "k_stereo4": "stereo", "k_load_level0": "load_level0"}


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
    cal_path = sys.argv[6] if len(sys.argv) > 6 else None
    cal = json.load(open(cal_path))["loads"] if cal_path else {}

    def factor(k):   # counter bytes -> request bytes
        e = cal.get(SHAPE.get(k, ""), None)
        return 1.0 / e["ratio"] if e and e.get("ratio") else 2.0

    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 bench.py`, see profiles/",
           "read_factor_source": cal_path or "MI355X_MICROARCH.md (x2, uncalibrated)",
           "what": "memory-side request bytes per launch (Infinity-Cache hits included): an upper bound of the HBM bytes",
           "pairs_per_step": int(sys.argv[3]), "images_per_launch": int(sys.argv[5]), "kernels": {}, "per_kernel": {}}
    stage = collections.defaultdict(lambda: [0.0, 0.0, 0])
    # a "launch" of a stage = one batched step (k_fast runs once per pyramid level, k_load_level0 once per eye): average per step
    steps_f, steps_w = max(nf["k_quadtree"], 1), max(nw["k_quadtree"], 1)
    for k in sorted(set(fetch) | set(write)):
        f = fetch[k] / steps_f * 1024.0
        w = write[k] / steps_w * 1024.0
        out["per_kernel"][k] = {"launches_per_step": nf[k] / steps_f, "fetch_bytes_raw": f, "read_factor": factor(k),
                                "fetch_bytes_corrected": factor(k) * f, "write_bytes": w}
        if k in STAGE:
            stage[STAGE[k]][0] += factor(k) * f
            stage[STAGE[k]][1] += w
    for s, (f, w, _) in stage.items():
        out["kernels"][s] = {"hbm_bytes_per_launch": f + w, "read_bytes_corrected": f, "write_bytes": w}
    json.dump(out, open(sys.argv[4], "w"), indent=1, sort_keys=True)
    for s, v in out["kernels"].items():
        print(f"{s:14s} read {v['read_bytes_corrected'] / 1e6:9.1f} MB  write {v['write_bytes'] / 1e6:9.1f} MB per launch")


if __name__ == "__main__":
    main()
