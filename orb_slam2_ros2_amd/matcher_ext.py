"""Host logic of the remaining callers of ORBMatcher::getBestMatch (include/ORB_SLAM2/ORBMatcher.h:41-75) around the batched device
search: searchBySim3 x2 (src/ORBMatcher.cc:370-559), fuse x2 / processFuseMps (:623-734) and searchForTriangulation (:736-787).

Map state enters as arrays and the side effects on the map come back as data, as in frontend.ORBMatcher.  Float arithmetic follows the
reference's expressions in float32.  Where the reference multiplies cv::Mat objects (un-vendored OpenCV, cv::gemm), the convention of
csrc/k_guided.hip is used, a decision that could not be checked against OpenCV here: a 3x3 / 4x4 product element is the float sum
a0 b0 + a1 b1 + a2 b2 (+ a3 b3) taken left to right, and `alpha * A * x + t` is (float)((double)alpha * (double)sum + (double)t).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def _matvec(R, x):
    """rows of R (3x3 float32) times x (3,) in float32, left to right"""
    R = np.asarray(R, F32).reshape(3, 3)
    return np.array([F32(F32(F32(R[r, 0] * x[0]) + F32(R[r, 1] * x[1])) + F32(R[r, 2] * x[2])) for r in range(3)], F32)


def _affine(alpha, R, x, t):
    """(float)(alpha * (R x) + t) with the sum R x in float and the scale / shift in double (cv::gemm's alpha / beta)"""
    s = _matvec(R, x)
    return np.array([F32(np.float64(alpha) * np.float64(s[r]) + np.float64(np.asarray(t, F32).reshape(3)[r])) for r in range(3)], F32)


def _matmul(A, B):
    A, B = np.asarray(A, F32), np.asarray(B, F32)
    n, m, k = A.shape[0], B.shape[1], A.shape[1]
    out = np.zeros((n, m), F32)
    for i in range(n):
        for j in range(m):
            acc = F32(A[i, 0] * B[0, j])
            for q in range(1, k):
                acc = F32(acc + F32(A[i, q] * B[q, j]))
            out[i, j] = acc
    return out


def predict_level(max_dist, d, log_sf):
    """MapPoint::predictLevel (src/MapPoint.cc:188-199): cvRound(std::log(nMaxDis / distance) / std::log(mfScaledFactor)), clamped to [0, 7]"""
    lr = F32(np.log(np.float64(F32(F32(max_dist) / F32(d)))))
    lvl = int(np.rint(F32(lr / F32(log_sf))))
    return min(7, max(0, lvl))


def sim3_project(pos, max_dist, min_dist, Rcw, tcw, sim3, cam, bounds, log_sf):
    """The geometric half of ORBMatcher::SIM3Project (src/ORBMatcher.cc:370-392) for every point: world position -> source camera
    (Rcw, tcw; None: the point is already in the frame the similarity maps from) -> target camera through the similarity
    sim3 = (s, R, t) -> pixel.  Returns (ok [n], uv [n, 2], octave [n]): ok = in front, inside the image bounds (isInImage of mpCurr,
    :382-383), distance / s inside the map point's range (:388-390)."""
    pos = np.asarray(pos, F32).reshape(-1, 3)
    n = len(pos)
    s, R, t = sim3
    fx, fy, cx, cy = (F32(v) for v in cam)
    min_u, max_u, min_v, max_v = (F32(v) for v in bounds)
    ok = np.zeros(n, bool)
    uv = np.zeros((n, 2), F32)
    octave = np.zeros(n, np.int32)
    for i in range(n):
        pc = pos[i] if Rcw is None else _affine(1.0, Rcw, pos[i], tcw)
        pm = _affine(F32(s), R, pc, t)
        if pm[2] <= 0:
            continue
        x, y = F32(pm[0] / pm[2]), F32(pm[1] / pm[2])                                     # Camera::project (src/Camera.cc:14-22)
        u, v = F32(F32(fx * x) + cx), F32(F32(fy * y) + cy)
        if not (u < max_u and v < max_v and u > min_u and v > min_v):
            continue
        d = F32(np.sqrt(F32(F32(F32(pm[0] * pm[0]) + F32(pm[1] * pm[1])) + F32(pm[2] * pm[2]))) / F32(s))
        if not (d < F32(max_dist[i]) and d > F32(min_dist[i])):
            continue
        ok[i], uv[i], octave[i] = True, (u, v), predict_level(max_dist[i], d, log_sf)
    return ok, uv, octave


class MatcherExt:
    """mixed into frontend.ORBMatcher"""

    def _area_best(self, search, uv, octave, desc, th, scale_factors, exclude):
        """findFeaturesInArea(kp, th, octave - 1, octave + 1) + getBestMatch + `dist <= mnMinThreshold && ratio <= mfRatio` (:393-414)"""
        sf2 = np.asarray(scale_factors, F32) ** 2
        radius = (F32(th) * sf2[octave]).astype(F32)
        bi, bd, sd, nc = search(uv, radius, (octave - 1).astype(np.int8), (octave + 1).astype(np.int8), desc, exclude)
        ratio = bd.astype(F32) / sd.astype(F32)
        return bi, (nc > 0) & (bd <= self.mnMinThreshold) & (ratio <= F32(self.mfRatio))

    def searchBySim3Frames(self, ctx, matches, Scm, poseC, poseM, kfC, kfM, th, cam, bounds, scale_factors, search_in=None):
        """ORBMatcher::searchBySim3(mpCurr, mpMatch, matches, g2oScm, th) (src/ORBMatcher.cc:424-484).
        matches: [(queryIdx in C, trainIdx in M)] from the Sim3 solver; Scm = (s, R, t) with p_c = s R p_m + t; poseC / poseM = (Rcw, tcw).
        kfC / kfM: dict(kps [KP_DTYPE], desc [n, 32], pos [n, 3] world positions of the keyframe's map points (rows of features without one
        are ignored), good [n] = map point non-null and not bad, inmap [n], max_dist [n], min_dist [n]).
        Returns the extended match list [(queryIdx, trainIdx)]: the input followed by the new pairs in ascending queryIdx (std::map order)."""
        s, R, t = Scm
        R = np.asarray(R, F32).reshape(3, 3)
        s_inv = F32(F32(1.0) / F32(s))                                                      # Sim3Ret::inv (Sim3Solver.h:36-43)
        Rt = R.T.copy()
        Smc = (s_inv, Rt, (-s_inv * _matvec(Rt, np.asarray(t, F32).reshape(3))).astype(F32))
        log_sf = F32(np.log(F32(scale_factors[1])))
        nc, nm = len(kfC["kps"]), len(kfM["kps"])
        flagC, flagM = np.ones(nc, bool), np.ones(nm, bool)
        for q, tr in matches:
            flagC[q], flagM[tr] = False, False
        new = {}

        def direction(src, dst, flags, pose, sim3, need_inmap_src, forward, search):
            cand = np.flatnonzero(flags & np.asarray(src["good"], bool) & (np.asarray(src["inmap"], bool) if need_inmap_src else True))
            if cand.size == 0:
                return
            ok, uv, octave = sim3_project(np.asarray(src["pos"], F32)[cand], np.asarray(src["max_dist"], F32)[cand],
                                          np.asarray(src["min_dist"], F32)[cand], pose[0], pose[1], sim3, cam, bounds, log_sf)
            cand, uv, octave = cand[ok], uv[ok], octave[ok]
            if cand.size == 0:
                return
            usable_dst = np.asarray(dst["good"], bool) & np.asarray(dst["inmap"], bool)          # vGoodIndices (:396-405)
            bi, keep = self._area_best(search, uv, octave, np.asarray(src["desc"])[cand], th, scale_factors, (~usable_dst).astype(np.uint8))
            for k in np.flatnonzero(keep):
                key, val = (int(cand[k]), int(bi[k])) if forward else (int(bi[k]), int(cand[k]))
                new.setdefault(key, val)                                                           # std::map::insert keeps the first

        def target(dst, tag):   # search_in(tag, ...) lets the logic be exercised without a device
            if search_in is not None:
                return lambda *a: search_in(tag, *a)
            return lambda *a: ctx.search_in_area_features(dst["kps"], dst["desc"], *a)
        # C's map points into M (:448-462), then M's into C (:464-476)
        direction(kfC, kfM, flagC, poseC, Smc, True, True, target(kfM, "M"))
        direction(kfM, kfC, flagM, poseM, (F32(s), R, np.asarray(t, F32).reshape(3)), False, False, target(kfC, "C"))
        return list(matches) + sorted(new.items())

    def searchBySim3MapPoints(self, ctx, kf, loop_mps, matched, Scw, th, cam, bounds, scale_factors, search_in=None):
        """ORBMatcher::searchBySim3(pCurr, vLoopGroupMps, vMatchedMps, g2oScw, th) (src/ORBMatcher.cc:501-559).
        kf: dict(kps, desc) of pCurr; loop_mps: dict(pos [n, 3], view_dir [n, 3], desc [n, 32], max_dist, min_dist, usable [n] = non-null, not
        bad, in map, id [n] = identity of the map point); matched: [nKF] identity of the map point each feature of pCurr is matched
        with, -1 for none (vMatchedMps; entries whose map point is no longer usable do not count, the caller passes -1 for them).
        Returns (assignments [(featIdx, loop map point index)] in loop order -- vMatchedMps[bestMatch.first] = pMp --, nMatches)."""
        s, R, t = Scw
        R = np.asarray(R, F32).reshape(3, 3)
        t = np.asarray(t, F32).reshape(3)
        matched = np.asarray(matched, np.int64)
        already = set(int(v) for v in matched if v >= 0)
        n_matches = int((matched >= 0).sum())
        ids = np.asarray(loop_mps["id"], np.int64)
        cand = np.array([i for i in range(len(ids)) if loop_mps["usable"][i] and int(ids[i]) not in already], np.int64)
        if cand.size == 0:
            return [], n_matches
        log_sf = F32(np.log(F32(scale_factors[1])))
        pos = np.asarray(loop_mps["pos"], F32)[cand]
        ok, uv, octave = sim3_project(pos, np.asarray(loop_mps["max_dist"], F32)[cand], np.asarray(loop_mps["min_dist"], F32)[cand], None, None,
                                      (F32(s), R, t), cam, bounds, log_sf)
        # the viewing-angle test (:534-536): (Rqp * viewDirection) . p3dC >= 0.5 * |p3dC|, dot and norm accumulated in double (cv::Mat::dot, cv::norm)
        vd = np.asarray(loop_mps["view_dir"], F32)[cand]
        for j in np.flatnonzero(ok):
            pc = _affine(F32(s), R, pos[j], t)
            d_with_s = F32(np.sqrt(np.float64(pc[0]) ** 2 + np.float64(pc[1]) ** 2 + np.float64(pc[2]) ** 2))
            rv = _matvec(R, vd[j])
            dot = np.float64(rv[0]) * np.float64(pc[0]) + np.float64(rv[1]) * np.float64(pc[1]) + np.float64(rv[2]) * np.float64(pc[2])
            if dot < 0.5 * np.float64(d_with_s):
                ok[j] = False
        cand, uv, octave = cand[ok], uv[ok], octave[ok]
        if cand.size == 0:
            return [], n_matches
        search = search_in or (lambda *a: ctx.search_in_area_features(kf["kps"], kf["desc"], *a))
        bi, keep = self._area_best(search, uv, octave, np.asarray(loop_mps["desc"])[cand], th, scale_factors, None)
        out = [(int(bi[k]), int(cand[k])) for k in np.flatnonzero(keep)]
        return out, n_matches + len(out)

    def fuseMapPoints(self, ctx, kf, kf_state, mps, th=3.0, bLoop=False, scale_factors=None, search_in=None):
        """ORBMatcher::fuse(pkf1, mapPoints, map, bLoop, th) (src/ORBMatcher.cc:682-707): the map points already held by the keyframe are
        dropped (:687-701), the rest go through searchByProjection(pkf1, vMapPoints, th, matches, true) against the keyframe's features,
        then processFuseMps.  kf: dict(kps, desc); kf_state: dict(good [n], id [n], obs [n]) of the keyframe's map points; mps: dict(id,
        good, obs, usable [m] = in map, not bad, in vision of pkf1, uv [m, 2], level [m], cos_theta [m], desc [m, 32]) -- uv / level /
        cos_theta from ORBMatcher.projectMapPoints.  Returns (actions, nFuse) of processFuseMps; action indices refer to `mps`."""
        held = set(int(i) for i, g in zip(kf_state["id"], kf_state["good"]) if g)
        sel = np.array([i for i in range(len(mps["id"])) if int(mps["id"][i]) not in held], np.int64)
        if sel.size == 0:
            return [], 0
        search = search_in or (lambda *a: ctx.search_in_area_features(kf["kps"], kf["desc"], *a))
        matches, _ = self.searchByProjectionMapPoints(ctx, 0, np.asarray(mps["uv"], F32)[sel], np.asarray(mps["level"])[sel],
                                                      np.asarray(mps["cos_theta"], F32)[sel], np.asarray(mps["desc"])[sel],
                                                      np.asarray(mps["usable"], bool)[sel], th, np.zeros(len(kf["kps"]), bool), True,
                                                      scale_factors=scale_factors, area_search=search)
        matches = [(f, int(sel[m]), d) for f, m, d in matches]
        return self.processFuseMps(matches, kf_state["good"], kf_state["id"], mps["good"], mps["id"], kf_state["obs"], mps["obs"], bLoop)

    def fuseFrames(self, ctx, kf1, kf1_state, kf2_kps, kf2_desc, kf2_state, in_vision2, z, bl, scale_factors, search_in=None):
        """ORBMatcher::fuse(pkf1, pkf2, map) (src/ORBMatcher.cc:716-724): searchByProjection(pkf1, pkf2, matches, 3.0f, true) + processFuseMps.
        kf2_state: dict(good, id, obs) per feature of pkf2; in_vision2 [n2]: the map point of feature idx is in vision of pkf1 (:283-288)."""
        search = search_in or (lambda *a: ctx.search_in_area_features(kf1["kps"], kf1["desc"], *a))
        matches = self.searchByProjectionFrames(ctx, 0, scale_factors, kf2_kps, kf2_desc, kf2_state["good"], np.zeros(len(kf1["kps"]), bool), 3.0,
                                                z, bl, True, in_vision2, area_search=search)
        return self.processFuseMps(matches, kf1_state["good"], kf1_state["id"], kf2_state["good"], kf2_state["id"], kf1_state["obs"],
                                   kf2_state["obs"], False)

    def searchForTriangulation(self, ctx, bow_args, kps1, kps2, pose1, pose1_inv, pose2, pose2_inv, k_inv, scale_factors):
        """ORBMatcher::searchForTriangulation (src/ORBMatcher.cc:736-787): searchByBow(pkf1, pkf2, matches, true) -- bow_args are the
        keyword arguments of ORBMatcher.searchByBow for (frame = pkf1, keyframe = pkf2) -- then the mutual epipolar test."""
        matches = self.searchByBow(ctx, bAddMPs=True, **bow_args)
        if not matches:
            return []
        return self.epipolarFilter(matches, kps1, kps2, pose1, pose1_inv, pose2, pose2_inv, k_inv, scale_factors)

    @staticmethod
    def processFuseMps(matches, f_good, f_id, v_good, v_id, f_obs, v_obs, bLoop=False):
        """ORBMatcher::processFuseMps (src/ORBMatcher.cc:623-661): what to do with every match (queryIdx = feature of pkf1, trainIdx = map
        point index) given which features of pkf1 carry a good map point (f_good, identity f_id, observation count f_obs) and the same for
        the projected map points.  Returns (actions, nFuse): ("add", featIdx, mpIdx) = setMapPoint + addObservation,
        ("replace", keep_id, drop_id) = MapPoint::replace(keep, drop, map).  Within one call the map changes are NOT fed back into later
        decisions, exactly like the reference, which works on the copies taken before the loop (:705-706, :727-729)."""
        actions, n = [], 0
        for q, tr, *_ in matches:
            if not v_good[tr]:
                continue
            if not f_good[q]:
                actions.append(("add", int(q), int(tr)))
                n += 1
            elif f_id[q] != v_id[tr]:
                if bLoop:
                    actions.append(("replace", int(v_id[tr]), int(f_id[q])))
                elif f_obs[q] >= v_obs[tr]:
                    actions.append(("replace", int(f_id[q]), int(v_id[tr])))
                else:
                    actions.append(("replace", int(v_id[tr]), int(f_id[q])))
                n += 1
        return actions, n

    @staticmethod
    def epipolarFilter(matches, kps1, kps2, pose1, pose1_inv, pose2, pose2_inv, k_inv, scale_factors):
        """The second half of ORBMatcher::searchForTriangulation (src/ORBMatcher.cc:739-786): F21 = KInv^T [t21]x R21 KInv from the 4x4
        float poses (T21 = Tcw2 * Twc1), keep a match iff both point-to-epipolar-line distances are within 5.991 * scale^2 of the octave.
        matches: [(queryIdx in kf1, trainIdx in kf2, ...)] from searchByBow(pkf1, pkf2, matches, true)."""
        T21, T12 = _matmul(pose2, pose1_inv), _matmul(pose1, pose2_inv)
        K = np.asarray(k_inv, F32).reshape(3, 3)

        def fundamental(T):
            Rm, tv = T[:3, :3], T[:3, 3]
            ssm = np.array([[0, -tv[2], tv[1]], [tv[2], 0, -tv[0]], [-tv[1], tv[0], 0]], F32)
            return _matmul(_matmul(_matmul(K.T.copy(), ssm), Rm), K)
        F21, F12 = fundamental(T21), fundamental(T12)
        sf2 = np.asarray(scale_factors, F32) ** 2

        def dist(p_line, Fm, p):
            """point2LineDistance(pt_line^T * F, p) (:789-795): |param . point| / sqrt(a^2 + b^2), the dot in double (cv::Mat::dot)"""
            prm = _matmul(np.asarray(p_line, F32).reshape(1, 3), Fm)[0]
            dot = np.float64(prm[0]) * np.float64(p[0]) + np.float64(prm[1]) * np.float64(p[1]) + np.float64(prm[2]) * np.float64(p[2])
            return F32(F32(abs(dot)) / F32(np.sqrt(F32(F32(prm[0] * prm[0]) + F32(prm[1] * prm[1])))))
        out = []
        for m in matches:
            k1, k2 = kps1[m[0]], kps2[m[1]]
            p1, p2 = np.array([k1["x"], k1["y"], 1], F32), np.array([k2["x"], k2["y"], 1], F32)
            if dist(p2, F21, p1) > F32(5.991 * np.float64(sf2[k1["octave"]])):
                continue
            if dist(p1, F12, p2) > F32(5.991 * np.float64(sf2[k2["octave"]])):
                continue
            out.append(m)
        return out
