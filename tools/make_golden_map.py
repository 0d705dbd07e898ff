#!/usr/bin/env python3
"""Writes tests/golden/map_small.pb (+ .json): a small `orbslam2.MapData` file serialised by the REAL protobuf runtime
(google.protobuf, schema of the reference's proto/*.proto declared in tests/map_pb_util.py) and the local-map graph the numpy
restatement of src/Optimizer.cc:232-330 builds on it.  host/map_pb.hpp must parse the file, re-encode it byte for byte and build the
same graph (tests/test_golden.py).  Run from the repo root: python tools/make_golden_map.py"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import map_pb_util as U  # noqa: E402


def main():
    md, cam = U.synth_map(seed=11, n_kf=6, n_pt=160, kf_id_step=3, mp_id0=500, extra_kps=2)
    for k in md.keyframes.keyframes:  # one BoW word per keyframe: the order of protobuf map entries is unspecified, one entry has no order
        words = sorted(k.bow_vector.words.items())[:1]
        k.bow_vector.words.clear()
        for w, v in words:
            k.bow_vector.words[w] = v
    pb = md.SerializeToString()
    out = os.path.join(ROOT, "tests", "golden")
    open(os.path.join(out, "map_small.pb"), "wb").write(pb)
    g = U.local_graph(md, 9)
    meta = {
        "generator": "tools/make_golden_map.py (google.protobuf %s)" % __import__("google.protobuf").protobuf.__version__,
        "sha256": hashlib.sha256(pb).hexdigest(),
        "summary": {"next_id": int(md.keyframes.next_id), "n_scale_factors": len(md.keyframes.scale_factors),
                    "n_keyframes": len(md.keyframes.keyframes), "n_mappoints": len(md.mappoints.mappoints),
                    "n_keypoints": sum(len(k.keypoints) for k in md.keyframes.keyframes),
                    "n_observations": sum(sum(1 for m in k.map_points if m >= 0) for k in md.keyframes.keyframes)},
        "graph_kf_id": 9,
        "graph": {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in g.items()},
        "camera": cam,
    }
    json.dump(meta, open(os.path.join(out, "map_small.json"), "w"))
    print(len(pb), meta["summary"], g["n_group"], len(g["edge_pose"]))


if __name__ == "__main__":
    main()
