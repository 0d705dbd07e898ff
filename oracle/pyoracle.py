"""ctypes binding of the CPU oracle (oracle/liborb_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package (orb_slam2_ros2_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

KP_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")]
)
assert KP_DTYPE.itemsize == 28


def build(fast: bool = False, out_dir: str | None = None) -> str:
    """Compile the oracle if needed and return the path of the .so."""
    name = "liborb_oracle_fast.so" if fast else "liborb_oracle.so"
    if out_dir is None:
        subprocess.check_call(["make", "-s", "-C", _HERE, name])
        return os.path.join(_HERE, name)
    # out-of-tree build (bench.py builds the -march=native flavour on the box it runs on)
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, name)
    flags = ["-O3", "-march=native"] if fast else ["-O2"]
    cmd = ["g++", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-pthread", *flags, "-o", out,
           os.path.join(_HERE, "orb_oracle.cpp"), os.path.join(_HERE, "ba_oracle.cpp"), os.path.join(_HERE, "pose_oracle.cpp"),
           os.path.join(_HERE, "lba_oracle.cpp"), os.path.join(_HERE, "glue_oracle.cpp")]
    subprocess.check_call(cmd)
    return out


_u8p = C.POINTER(C.c_uint8)


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


class Oracle:
    def __init__(self, path: str | None = None):
        self.path = path or build()
        L = self.lib = C.CDLL(self.path)
        L.orc_extractor_create.restype = C.c_void_p
        L.orc_extractor_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                           C.c_void_p, C.c_int, C.c_int]
        L.orc_extractor_destroy.argtypes = [C.c_void_p]
        L.orc_extractor_level_info.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4
        L.orc_extractor_plane.restype = C.c_void_p
        L.orc_extractor_plane.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_extractor_umax.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_extractor_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_extractor_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_extractor_splits.restype = C.c_long
        L.orc_extractor_splits.argtypes = [C.c_void_p, C.c_int]
        L.orc_extractor_thetas.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_extractor_describe.restype = C.c_double
        L.orc_extractor_describe.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_ba_eval_edges.argtypes = [C.c_int] + [C.c_void_p] * 8 + [C.c_double] * 5 + [C.c_void_p] * 6
        L.orc_se3_oplus.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_search_in_area.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 10
        L.orc_pose_only_optimize.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_double] * 5 + [C.c_void_p] * 2
        L.orc_cvt_gray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_cvt_gray_v.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_undistort_points.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_rgbd_lookup.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p,
                                      C.c_void_p]
        L.orc_ba_local_optimize.argtypes = [C.c_int] * 3 + [C.c_void_p] * 8 + [C.c_double] * 5 + [C.c_void_p] + [C.c_int] * 2 + [C.c_void_p] * 6
        L.orc_ba_build_system.argtypes = [C.c_int] * 3 + [C.c_void_p] * 8 + [C.c_double] * 5 + [C.c_void_p] * 7
        L.orc_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_gauss7_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_fast9_16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_quadtree_select.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_hamming256.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_best_match.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4
        L.orc_match_bruteforce.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_atan2.restype = C.c_double
        L.orc_atan2.argtypes = [C.c_double, C.c_double, C.c_int]
        L.orc_sincos.argtypes = [C.c_double, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_stereo_match.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_stereo_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                       C.c_int, C.c_float, C.c_float, C.c_int, C.c_int] + [C.c_void_p] * 8

    # ---- primitives ------------------------------------------------------------------------
    def resize(self, src: np.ndarray, dw: int, dh: int) -> np.ndarray:
        src = np.ascontiguousarray(src, dtype=np.uint8)
        dst = np.empty((dh, dw), np.uint8)
        self.lib.orc_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.shape[1], _p(dst), dw, dh, dw)
        return dst

    def gauss7(self, src: np.ndarray, variant: int = 0) -> np.ndarray:
        src = np.ascontiguousarray(src, dtype=np.uint8)
        dst = np.empty_like(src)
        self.lib.orc_gauss7_u8(_p(src), src.shape[1], src.shape[0], src.shape[1], _p(dst), src.shape[1], variant)
        return dst

    def fast(self, patch: np.ndarray, threshold: int, nonmax: bool = True) -> np.ndarray:
        patch = np.ascontiguousarray(patch, dtype=np.uint8)
        cap = patch.size
        out = np.empty((cap, 3), np.int32)
        n = self.lib.orc_fast9_16(_p(patch), patch.shape[1], patch.shape[1], patch.shape[0], threshold, int(nonmax), _p(out), cap)
        return out[:n].copy()

    def quadtree(self, w: int, h: int, xyr: np.ndarray, need: int):
        xyr = np.ascontiguousarray(xyr, dtype=np.float32).reshape(-1, 3)
        out = np.empty(max(need, 1) + 8, np.int32)
        ns = C.c_int64(0)
        n = self.lib.orc_quadtree_select(w, h, _p(xyr), xyr.shape[0], need, _p(out), C.byref(ns))
        return out[:n].copy(), ns.value

    def hamming(self, a: np.ndarray, b: np.ndarray) -> int:
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        return self.lib.orc_hamming256(_p(a), _p(b))

    def best_match(self, q: np.ndarray, train: np.ndarray, cand: np.ndarray):
        q = np.ascontiguousarray(q, np.uint8)
        train = np.ascontiguousarray(train, np.uint8)
        cand = np.ascontiguousarray(cand, np.int64)
        bi, bd, sd, ratio = C.c_int64(0), C.c_int(0), C.c_int(0), C.c_float(0)
        self.lib.orc_best_match(_p(q), _p(train), _p(cand), cand.size, C.byref(bi), C.byref(bd), C.byref(sd), C.byref(ratio))
        return bi.value, bd.value, sd.value, ratio.value

    def match_bruteforce(self, q: np.ndarray, t: np.ndarray):
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        nq, nt = q.shape[0], t.shape[0]
        bi = np.empty(nq, np.int32)
        bd = np.empty(nq, np.int32)
        sd = np.empty(nq, np.int32)
        self.lib.orc_match_bruteforce(_p(q), nq, _p(t), nt, _p(bi), _p(bd), _p(sd))
        return bi, bd, sd

    def atan2(self, y, x, mode=0):
        return self.lib.orc_atan2(float(y), float(x), mode)

    def sincos(self, t, mode=0):
        s, c = C.c_double(0), C.c_double(0)
        self.lib.orc_sincos(float(t), mode, C.byref(s), C.byref(c))
        return s.value, c.value

    def search_in_area_ex(self, kps, desc, bounds, qxy, radius, min_level, max_level, q_desc, exclude=None):
        """as search_in_area with the target frame's bounds (min_u, max_u, min_v, max_v); also returns the excluded-hit counts"""
        kps = np.ascontiguousarray(kps)
        desc = np.ascontiguousarray(desc, np.uint8)
        qxy = np.ascontiguousarray(qxy, np.float32).reshape(-1, 2)
        nq = qxy.shape[0]
        radius = np.ascontiguousarray(radius, np.float32)
        min_level = np.ascontiguousarray(min_level, np.int8)
        max_level = np.ascontiguousarray(max_level, np.int8)
        q_desc = np.ascontiguousarray(q_desc, np.uint8).reshape(-1, 32)
        bnd = np.ascontiguousarray(bounds, np.float32).reshape(4)
        ex = None if exclude is None else np.ascontiguousarray(exclude, np.uint8)
        out = [np.zeros(max(nq, 1), np.int32) for _ in range(4)]
        hits = np.zeros(max(kps.shape[0], 1), np.int32)
        self.lib.orc_search_in_area_ex(_p(kps), _p(desc), kps.shape[0], _p(bnd), nq, _p(qxy), _p(radius), _p(min_level), _p(max_level),
                                       _p(q_desc), _p(ex) if ex is not None else None, *[_p(o) for o in out], _p(hits))
        return tuple(o[:nq] for o in out) + (hits[:kps.shape[0]],)

    def search_in_area(self, kps, desc, width, height, qxy, radius, min_level, max_level, q_desc, exclude=None):
        kps = np.ascontiguousarray(kps)
        desc = np.ascontiguousarray(desc, np.uint8)
        qxy = np.ascontiguousarray(qxy, np.float32).reshape(-1, 2)
        nq = qxy.shape[0]
        radius = np.ascontiguousarray(radius, np.float32)
        min_level = np.ascontiguousarray(min_level, np.int8)
        max_level = np.ascontiguousarray(max_level, np.int8)
        q_desc = np.ascontiguousarray(q_desc, np.uint8).reshape(-1, 32)
        ex = None if exclude is None else np.ascontiguousarray(exclude, np.uint8)
        out = [np.zeros(max(nq, 1), np.int32) for _ in range(4)]
        self.lib.orc_search_in_area(_p(kps), _p(desc), kps.shape[0], width, height, nq, _p(qxy), _p(radius), _p(min_level), _p(max_level),
                                    _p(q_desc), _p(ex) if ex is not None else None, *[_p(o) for o in out])
        return tuple(o[:nq] for o in out)

    # ---- BA ------------------------------------------------------------------------------------
    def ba_eval_edges(self, poses, points, edge_pose, edge_point, meas, is_stereo, info, huber_delta, fx, fy, cx, cy, bf):
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
        points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
        edge_pose = np.ascontiguousarray(edge_pose, np.int32)
        edge_point = np.ascontiguousarray(edge_point, np.int32)
        meas = np.ascontiguousarray(meas, np.float64).reshape(-1, 3)
        is_stereo = np.ascontiguousarray(is_stereo, np.uint8)
        info = np.ascontiguousarray(info, np.float64)
        huber_delta = np.ascontiguousarray(huber_delta, np.float64)
        E = edge_pose.size
        out = dict(error=np.zeros((E, 3)), chi2=np.zeros(E), rho=np.zeros((E, 2)), j_point=np.zeros((E, 3, 3)),
                   j_pose=np.zeros((E, 3, 6)), depth_positive=np.zeros(E, np.uint8))
        self.lib.orc_ba_eval_edges(E, _p(poses), _p(points), _p(edge_pose), _p(edge_point), _p(meas), _p(is_stereo), _p(info),
                                   _p(huber_delta), fx, fy, cx, cy, bf, _p(out["error"]), _p(out["chi2"]), _p(out["rho"]),
                                   _p(out["j_point"]), _p(out["j_pose"]), _p(out["depth_positive"]))
        return out

    def ba_build_system(self, poses, points, edge_pose, edge_point, meas, is_stereo, info, huber_delta, fx, fy, cx, cy, bf, pose_fixed):
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
        points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
        edge_pose = np.ascontiguousarray(edge_pose, np.int32)
        edge_point = np.ascontiguousarray(edge_point, np.int32)
        meas = np.ascontiguousarray(meas, np.float64).reshape(-1, 3)
        is_stereo = np.ascontiguousarray(is_stereo, np.uint8)
        info = np.ascontiguousarray(info, np.float64)
        huber_delta = np.ascontiguousarray(huber_delta, np.float64)
        pose_fixed = np.ascontiguousarray(pose_fixed, np.uint8)
        nk, npt, E = poses.shape[0], points.shape[0], edge_pose.size
        out = dict(Hpp=np.zeros((nk, 6, 6)), bp=np.zeros((nk, 6)), Hll=np.zeros((npt, 3, 3)), bl=np.zeros((npt, 3)), Hpl=np.zeros((E, 6, 3)))
        tot = C.c_double(0)
        self.lib.orc_ba_build_system(nk, npt, E, _p(poses), _p(points), _p(edge_pose), _p(edge_point), _p(meas), _p(is_stereo), _p(info),
                                     _p(huber_delta), fx, fy, cx, cy, bf, _p(pose_fixed), _p(out["Hpp"]), _p(out["bp"]), _p(out["Hll"]),
                                     _p(out["bl"]), _p(out["Hpl"]), C.byref(tot))
        out["chi2_robust"] = tot.value
        return out

    def pose_only_optimize(self, Xw, meas, info, sigma2, pose, fx, fy, cx, cy, bf):
        Xw = np.ascontiguousarray(Xw, np.float64).reshape(-1, 3)
        meas = np.ascontiguousarray(meas, np.float64).reshape(-1, 3)
        info = np.ascontiguousarray(info, np.float64)
        sigma2 = np.ascontiguousarray(sigma2, np.float32)
        pose = np.ascontiguousarray(pose, np.float64)
        n = Xw.shape[0]
        out = np.zeros(7)
        inl = np.zeros(max(n, 1), np.uint8)
        r = self.lib.orc_pose_only_optimize(n, _p(Xw), _p(meas), _p(info), _p(sigma2), _p(pose), fx, fy, cx, cy, bf, _p(out), _p(inl))
        return r, out, inl[:n].astype(bool)

    # ---- frame glue -------------------------------------------------------------------------
    def cvt_gray(self, img: np.ndarray, order: int, variant: int = 0) -> np.ndarray:
        """variant 0: 14-bit coefficients (default), 1: the 15-bit ones of newer OpenCV 4.x builds"""
        img = np.ascontiguousarray(img, np.uint8)
        h, w, _ = img.shape
        out = np.empty((h, w), np.uint8)
        self.lib.orc_cvt_gray_v(_p(img), w, h, 3 * w, order, variant, _p(out), w)
        return out

    def undistort_points(self, xy, K, D):
        xy = np.array(xy, np.float32).reshape(-1, 2).copy()
        K = np.ascontiguousarray(K, np.float32)
        D = np.ascontiguousarray(D, np.float32)
        self.lib.orc_undistort_points(xy.shape[0], _p(xy), _p(K), _p(D))
        return xy

    def rgbd_lookup(self, xy, xy_u, depth, depth_scale, bf):
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        xy_u = np.ascontiguousarray(xy_u, np.float32).reshape(-1, 2)
        depth = np.ascontiguousarray(depth)
        assert depth.dtype in (np.uint16, np.float32)
        n = xy.shape[0]
        d, ru = np.zeros(max(n, 1)), np.zeros(max(n, 1))
        self.lib.orc_rgbd_lookup(n, _p(xy), _p(xy_u), _p(depth), 0 if depth.dtype == np.uint16 else 1, depth.strides[0], depth_scale, bf,
                                 _p(d), _p(ru))
        return d[:n], ru[:n]

    def project_map_points(self, pos, view_dir, max_dist, min_dist, Rcw, tcw, cam, bounds, scale_factor=1.2):
        """MapPoint::isInVision + predictLevel (MapPoint.cc:141-201): cam = fx fy cx cy, bounds = minU maxU minV maxV
        -> dict(uv, distance, cos_theta, level, visible)"""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        pos, view_dir, max_dist, min_dist = f32(pos).reshape(-1, 3), f32(view_dir).reshape(-1, 3), f32(max_dist), f32(min_dist)
        R, t, cam, bounds = f32(Rcw).reshape(9), f32(tcw).reshape(3), f32(cam), f32(bounds)
        n = pos.shape[0]
        m = max(n, 1)
        out = dict(uv=np.zeros((m, 2), np.float32), distance=np.zeros(m, np.float32), cos_theta=np.zeros(m, np.float32),
                   level=np.zeros(m, np.int8), visible=np.zeros(m, np.uint8))
        self.lib.orc_project_map_points.argtypes = [C.c_int] + [C.c_void_p] * 8 + [C.c_float] + [C.c_void_p] * 5
        self.lib.orc_project_map_points.restype = None
        self.lib.orc_project_map_points(n, _p(pos), _p(view_dir), _p(max_dist), _p(min_dist), _p(R), _p(t), _p(cam), _p(bounds),
                                        float(np.float32(scale_factor)), _p(out["uv"]), _p(out["distance"]), _p(out["cos_theta"]),
                                        _p(out["level"]), _p(out["visible"]))
        return {k: v[:n] for k, v in out.items()}

    def ba_local_optimize(self, prob, pose_fixed=None, iters1=5, iters2=10):
        """prob: dict as orb_slam2_ros2_amd.ba_synth.make_problem -> dict(poses, points, level, chi2, bad, iters)"""
        f64 = lambda a: np.ascontiguousarray(a, np.float64)
        poses, points, meas, info, delta = f64(prob["poses"]), f64(prob["points"]), f64(prob["meas"]), f64(prob["info"]), f64(prob["huber_delta"])
        ek = np.ascontiguousarray(prob["edge_pose"], np.int32)
        ep = np.ascontiguousarray(prob["edge_point"], np.int32)
        st = np.ascontiguousarray(prob["is_stereo"], np.uint8)
        nk, npnt, ne = poses.shape[0], points.shape[0], ek.shape[0]
        fixed = np.ascontiguousarray(pose_fixed if pose_fixed is not None else np.zeros(nk), np.uint8)
        po, xo = np.zeros((nk, 7)), np.zeros((npnt, 3))
        lvl, bad, chi2, its = np.zeros(max(ne, 1), np.uint8), np.zeros(max(ne, 1), np.uint8), np.zeros(max(ne, 1)), np.zeros(2, np.int32)
        self.lib.orc_ba_local_optimize(nk, npnt, ne, _p(poses), _p(points), _p(ek), _p(ep), _p(meas), _p(st), _p(info), _p(delta),
                                       prob["fx"], prob["fy"], prob["cx"], prob["cy"], prob["bf"], _p(fixed), iters1, iters2, _p(po), _p(xo),
                                       _p(lvl), _p(chi2), _p(bad), _p(its))
        return dict(poses=po, points=xo, level=lvl[:ne], chi2=chi2[:ne], bad=bad[:ne], iters=its)

    def se3_oplus(self, T, upd):
        T = np.ascontiguousarray(T, np.float64)
        upd = np.ascontiguousarray(upd, np.float64)
        out = np.zeros(7)
        self.lib.orc_se3_oplus(_p(T), _p(upd), _p(out))
        return out

    # ---- extractor --------------------------------------------------------------------------
    def extractor(self, img: np.ndarray, n_features=2000, n_levels=8, scale=1.2, th_hi=20, th_lo=7, pattern=None, blur_variant=0,
                  math_mode=0) -> "OracleExtractor":
        return OracleExtractor(self, img, n_features, n_levels, scale, th_hi, th_lo, pattern, blur_variant, math_mode)

    def stereo_frame(self, left, right, n_features=2000, n_levels=8, scale=1.2, th_hi=20, th_lo=7, fx=718.856, bf=386.1448,
                     math_mode=0, threads=2, want_outputs=True):
        left = np.ascontiguousarray(left, np.uint8)
        right = np.ascontiguousarray(right, np.uint8)
        h, w = left.shape
        if not want_outputs:
            return self.lib.orc_stereo_frame(_p(left), _p(right), w, h, w, n_features, n_levels, scale, th_hi, th_lo, fx, bf,
                                             math_mode, threads, *([None] * 8))
        lk = np.zeros(n_features, KP_DTYPE)
        rk = np.zeros(n_features, KP_DTYPE)
        ld = np.zeros((n_features, 32), np.uint8)
        rd = np.zeros((n_features, 32), np.uint8)
        nl, nr = C.c_int32(0), C.c_int32(0)
        ru = np.zeros(n_features, np.float64)
        dp = np.zeros(n_features, np.float64)
        m = self.lib.orc_stereo_frame(_p(left), _p(right), w, h, w, n_features, n_levels, scale, th_hi, th_lo, fx, bf, math_mode,
                                      threads, _p(lk), _p(ld), C.byref(nl), _p(rk), _p(rd), C.byref(nr), _p(ru), _p(dp))
        nl, nr = nl.value, nr.value
        return dict(n_matches=m, lk=lk[:nl], ld=ld[:nl], rk=rk[:nr], rd=rd[:nr], right_u=ru[:nl], depth=dp[:nl])


class OracleExtractor:
    """Mirrors ORBExtractor (ORBExtractor.cc:205-214): the pyramid and blur are built in the constructor."""

    def __init__(self, orc: Oracle, img, n_features, n_levels, scale, th_hi, th_lo, pattern, blur_variant, math_mode):
        self.orc = orc
        img = np.ascontiguousarray(img, np.uint8)
        self.n_features, self.n_levels = n_features, n_levels
        pat = None
        if pattern is not None:
            pat = np.ascontiguousarray(pattern, np.int8).reshape(256, 4)
        self._pat = pat
        self.h = orc.lib.orc_extractor_create(_p(img), img.shape[1], img.shape[0], img.shape[1], n_features, n_levels, scale,
                                              th_hi, th_lo, _p(pat) if pat is not None else None, blur_variant, math_mode)
        if not self.h:
            raise ValueError("ImageSizeError: a pyramid level is smaller than 2*19 px")

    def __del__(self):
        if getattr(self, "h", None):
            self.orc.lib.orc_extractor_destroy(self.h)
            self.h = None

    def level_info(self, level):
        w, h, q = C.c_int(0), C.c_int(0), C.c_int(0)
        sf = C.c_float(0)
        self.orc.lib.orc_extractor_level_info(self.h, level, C.byref(w), C.byref(h), C.byref(sf), C.byref(q))
        return w.value, h.value, sf.value, q.value

    def plane(self, level, blurred=False) -> np.ndarray:
        w, h, _, _ = self.level_info(level)
        ptr = self.orc.lib.orc_extractor_plane(self.h, level, int(blurred))
        buf = (C.c_uint8 * (w * h)).from_address(ptr)
        return np.frombuffer(buf, np.uint8).reshape(h, w).copy()

    def describe(self, level, x, y):
        desc = np.zeros(32, np.uint8)
        m10, m01 = C.c_int32(0), C.c_int32(0)
        th = self.orc.lib.orc_extractor_describe(self.h, level, float(x), float(y), _p(desc), C.byref(m10), C.byref(m01))
        return th, desc, m10.value, m01.value

    def umax(self):
        out = np.zeros(16, np.int32)
        self.orc.lib.orc_extractor_umax(self.h, _p(out))
        return out

    def extract(self):
        kps = np.zeros(self.n_features + 64, KP_DTYPE)
        desc = np.zeros((self.n_features + 64, 32), np.uint8)
        n = self.orc.lib.orc_extractor_extract(self.h, _p(kps), _p(desc), kps.shape[0])
        return kps[:n].copy(), desc[:n].copy()

    def candidates(self, level):
        n = self.orc.lib.orc_extractor_candidates(self.h, level, None, 0)
        out = np.zeros((max(n, 1), 3), np.float32)
        self.orc.lib.orc_extractor_candidates(self.h, level, _p(out), n)
        return out[:n]

    def splits(self, level):
        return self.orc.lib.orc_extractor_splits(self.h, level)

    def thetas(self):
        out = np.zeros(self.n_features + 64, np.float64)
        n = self.orc.lib.orc_extractor_thetas(self.h, _p(out), out.size)
        return out[:n]

    def stereo_match(self, right: "OracleExtractor", lk, ld, rk, rd, fx, bf):
        lk = np.ascontiguousarray(lk)
        rk = np.ascontiguousarray(rk)
        ld = np.ascontiguousarray(ld, np.uint8)
        rd = np.ascontiguousarray(rd, np.uint8)
        nl, nr = lk.shape[0], rk.shape[0]
        ru = np.zeros(max(nl, 1), np.float64)
        dp = np.zeros(max(nl, 1), np.float64)
        br = np.zeros(max(nl, 1), np.int32)
        bd = np.zeros(max(nl, 1), np.int32)
        m = self.orc.lib.orc_stereo_match(self.h, right.h, _p(lk), _p(ld), nl, _p(rk), _p(rd), nr, fx, bf, _p(ru), _p(dp), _p(br),
                                          _p(bd))
        return m, ru[:nl], dp[:nl], br[:nl], bd[:nl]
