"""Host-side mirror of the reference's operator interface for the hot path (same names, argument
meaning and error behaviour), implemented on the C-ABI of include/orbfe.h.

Reference interfaces mirrored (relative to src/ORB_SLAM2/):
  ORBExtractor(image, nFeatures, pyramidLevels, scaleFactor, bfTemFp, maxThreshold, minThreshold) + extract()
      include/ORB_SLAM2/ORBExtractor.h:107-116
  ORBMatcher(ratio=0.6, checkOri=True).searchByStereo(frame) / descDistance(a, b)
      include/ORB_SLAM2/ORBMatcher.h:32-39, 77
  Optimizer -- the per-edge evaluation under OptimizeLocalMap            include/ORB_SLAM2/Optimizer.h:69-72
"""
from __future__ import annotations

import re

import numpy as np

from . import _lib
from ._lib import Context, ImageSizeError, OrbfeError  # noqa: F401


_ISTREAM_FLOAT = re.compile(r"\s*([+-]?(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?)")


def load_brief_template(path: str) -> np.ndarray:
    """Parse a BRIEF template file exactly as ORBExtractor::initBriefTemplate does (ORBExtractor.cc:242-267): skip the header line, then
    EVERY line is a pair `x1 y1 x2 y2` read with operator>> -- a value that does not parse, and every value after it, stays 0 (a blank
    line is the pair (0,0)-(0,0)); no count is checked there.  computeBRIEF (:426-456) walks the whole list but only the first 32 bytes
    = 256 pairs reach the descriptor (:405-406), so a longer file behaves like its first 256 lines; a shorter one makes the reference
    read past the end of a vector (undefined behaviour) and is refused here.  Raises FileNotFoundError like FileNotOpenError."""
    rows = []
    with open(path, "r") as f:
        text = f.read()
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()                       # std::getline yields no extra line after a final newline
    for ln in lines[1:]:
        vals = [0.0, 0.0, 0.0, 0.0]
        pos = 0
        for k in range(4):
            # operator>>(float): skip white space, then take the LONGEST prefix that reads as a number ("7x" is 7 and the NEXT extraction
            # fails on the 'x'); when nothing parses the stream is in a failed state: this and the remaining values keep their zeros
            m = _ISTREAM_FLOAT.match(ln, pos)
            if not m:
                break
            vals[k] = float(m.group(1))
            pos = m.end()
        if any(not (-129.0 < v < 128.0) for v in vals):   # (the C++ loader's cast to int8_t would be undefined: both loaders refuse)
            raise ValueError(f"{path}: BRIEF template value outside [-128, 127] in line {len(rows) + 2}")
        rows.append([int(v) for v in vals])               # truncation toward zero, like the (int8_t) cast
    if len(rows) < 256:
        raise ValueError(f"{path}: {len(rows)} BRIEF pairs; the descriptor needs 256 (the reference would index past its template)")
    return np.asarray(rows[:256], np.int8)


from .matcher_ext import MatcherExt  # searchBySim3 x2, processFuseMps, the epipolar half of searchForTriangulation

class _CtxCache:
    """One device context per (geometry, parameters): the reference builds a new extractor per image
    (Frame.cc:91-92); re-allocating device buffers per image would be absurd, so contexts are shared."""
    _cache: dict = {}

    @classmethod
    def get(cls, key, **kw) -> Context:
        ctx = cls._cache.get(key)
        if ctx is None or ctx.h is None:
            ctx = Context(**kw)
            cls._cache[key] = ctx
        return ctx

    @classmethod
    def clear(cls):
        for c in cls._cache.values():
            c.close()
        cls._cache.clear()


class ORBExtractor:
    """Per-image extractor, same construction/extract() split as the reference: the constructor takes the
    image (the reference builds pyramid + blur there, ORBExtractor.cc:205-214), extract() returns keypoints
    (structured array with cv::KeyPoint's fields) and an (N, 32) uint8 descriptor matrix."""

    mnBorderSize = 19  # ORBExtractor.cc:523

    def __init__(self, image, nFeatures, pyramidLevels, scaleFactor, bfTemFp=None, maxThreshold=20, minThreshold=7,
                 *, device_id=0, slot=0, max_images=2):
        image = np.ascontiguousarray(image, np.uint8)
        if image.ndim != 2:
            raise ValueError("ORBExtractor needs a single-channel 8-bit image (CV_8UC1)")
        pairs = load_brief_template(bfTemFp) if bfTemFp else None
        h, w = image.shape
        key = (w, h, nFeatures, pyramidLevels, float(scaleFactor), maxThreshold, minThreshold,
               pairs.tobytes() if pairs is not None else None, device_id, max_images)
        self.ctx = _CtxCache.get(key, width=w, height=h, n_features=nFeatures, n_levels=pyramidLevels, scale_factor=scaleFactor,
                                 fast_hi=maxThreshold, fast_lo=minThreshold, brief_pairs=pairs, device_id=device_id,
                                 max_images=max_images)
        self.image = image
        self.slot = slot
        self.mnLevels = pyramidLevels
        self.mfScaledFactor = float(scaleFactor)
        self._result = None

    def extract(self):
        if self.slot == 0:
            kps, desc = self.ctx.extract(self.image)
        else:  # fill the lower slots with this image too; only `slot` is read back
            kps, desc = self.ctx.extract_batch([self.image] * (self.slot + 1))[self.slot]
        self._result = (kps, desc)
        return kps, desc

    def getPyramid(self):
        """The 8 un-blurred level images (ORBExtractor.h:113); valid after extract()."""
        return [self.ctx.pyramid(self.slot, l, False) for l in range(self.mnLevels)]

    def getScaledFactors(self):
        return self.ctx.scale_factors()


class ORBMatcher(MatcherExt):
    mnMaxThreshold, mnMinThreshold, mnMeanThreshold = 100, 50, 75  # ORBMatcher.cc:1086-1088
    mnW, mnL, mnBinNum, mnBinChoose, mnFarParam = 5, 5, 30, 3, 35  # ORBMatcher.cc:1089-1093

    def __init__(self, ratio: float = 0.6, checkOri: bool = True):
        self.mfRatio, self.mbCheckOri = ratio, checkOri

    @staticmethod
    def descDistance(a, b) -> int:
        """Hamming distance of two 1x32 uint8 descriptors (ORBMatcher.cc:941-956); host-side helper for
        single pairs -- bulk matching goes through getBestMatches()."""
        a = np.ascontiguousarray(a, np.uint8).reshape(-1)
        b = np.ascontiguousarray(b, np.uint8).reshape(-1)
        if a.size != 32 or b.size != 32:
            raise ValueError("descriptors must be 1x32 uint8")
        return int(np.unpackbits(a ^ b).sum())

    @staticmethod
    def getBestMatches(ctx: Context, queries, train, cand_offsets=None, cand_idx=None):
        """Batched ORBMatcher::getBestMatch (ORBMatcher.cc:967-990): per query (best_idx, best_dist,
        second_dist) with the reference's scan-order semantics."""
        return ctx.match_bruteforce(queries, train, cand_offsets, cand_idx)

    @staticmethod
    def searchInArea(ctx: Context, slot, qxy, radius, min_level, max_level, q_desc, exclude=None):
        """Batched VirtualFrame::findFeaturesInArea + getBestMatch (Frame.cc:286-311, ORBMatcher.cc:967-990): the core of
        ORBMatcher::searchByProjection (ORBMatcher.cc:265-347, 561-612) against the device-resident features of `slot`."""
        return ctx.search_in_area(slot, qxy, radius, min_level, max_level, q_desc, exclude)

    # ---- the guided searches (host logic of the reference around the batched device calls) -------------------------------
    # Map state enters as plain arrays (which features already carry a good map point, which map points are usable ...);
    # the side effects on the map (setMapPoints, addMatchInTrack, ...) are returned as data and stay with the caller.
    # `area_search` / `best_match` default to the device calls; they are parameters so that this logic can be exercised on its own.

    def searchByProjectionFrames(self, ctx, slot1, scale_factors2, kps2, desc2, valid2, has_mp1, th, z, bl, bFuse=False, in_vision=None,
                                 area_search=None):
        """ORBMatcher::searchByProjection(pFrame1, pFrame2, matches, th, bFuse) (ORBMatcher.cc:265-347): every feature idx of frame 2
        that carries a usable map point (valid2[idx]) searches frame 1 (device slot `slot1`) around ITS OWN position with radius
        th * scale2[octave] (Frame.cc:289); octave window by the motion along z (`z` = tlc.z, `bl` = Camera::mfBl, :274-282).
        !bFuse: candidates that already have a good map point in frame 1 (has_mp1) are skipped (:316-330; the caller bumps
        addMatchInTrack for them).  Returns [(queryIdx in frame 1, trainIdx = idx, distance)] in idx order."""
        search = area_search or (lambda *a: ctx.search_in_area(slot1, *a))
        kps2 = np.asarray(kps2)
        idx = np.flatnonzero(np.asarray(valid2, bool) & (np.ones(len(kps2), bool) if not bFuse or in_vision is None else np.asarray(in_vision, bool)))
        if idx.size == 0:
            return []
        up, down = (abs(z) > bl and z > 0), (abs(z) > bl and not z > 0)
        octv = kps2["octave"][idx].astype(np.int32)
        lo = np.where(up, octv, np.where(down, 0, np.maximum(0, octv - 1))).astype(np.int8)
        hi = np.where(up, 7, np.where(down, octv, np.minimum(octv + 1, 7))).astype(np.int8)   # the reference hard-codes 7 (:302,:311)
        sf2 = np.asarray(scale_factors2, np.float32) ** 2             # getScaledFactor2(octave)
        radius = (np.float32(th) * sf2[octv]).astype(np.float32)
        qxy = np.stack([kps2["x"][idx], kps2["y"][idx]], 1).astype(np.float32)
        excl = None if bFuse else np.ascontiguousarray(has_mp1, np.uint8)
        bi, bd, sd, nc = search(qxy, radius, lo, hi, np.asarray(desc2)[idx], excl)
        ratio = bd.astype(np.float32) / sd.astype(np.float32)
        ok = (nc > 0) & (ratio < np.float32(self.mfRatio)) & (bd < self.mnMinThreshold)
        return [(int(bi[k]), int(idx[k]), int(bd[k])) for k in np.flatnonzero(ok)]

    @staticmethod
    def projectMapPoints(ctx, pos, view_dir, max_dist, min_dist, Rcw, tcw, cam, bounds):
        """MapPoint::isInVision + predictLevel (MapPoint.cc:141-201) for all candidate map points in one device call: returns
        dict(uv, distance, cos_theta, level, visible) -- the inputs searchByProjectionMapPoints takes (usable = visible & in map & not bad)."""
        return ctx.project_map_points(pos, view_dir, max_dist, min_dist, Rcw, tcw, cam, bounds)

    def searchByProjectionMapPoints(self, ctx, slot, uv, level, cos_theta, mp_desc, usable, th, frame_has_good_mp, bFuse=False, n_levels=8,
                                    scale_factors=None, area_search=None):
        """ORBMatcher::searchByProjection(pframe, mapPoints, th, matches, bFuse) (ORBMatcher.cc:561-612).  Per map point the caller
        supplies what MapPoint::isInVision / predictLevel produced (MapPoint.cc:141-201): projection uv, predicted level, cosTheta and
        usable = in map, not bad, in vision.  Radius (2.5 if cos > 0.998 else 4.0) * th * scale2[level], levels level-1..level+1.
        bFuse: returns (matches [(featIdx, mapPointIdx, dist)], nMatches).  Otherwise: (assignments [(featIdx, mapPointIdx)] the caller
        applies with setMapPoint / addMatchInTrack -- first map point wins a feature, features with a good map point are left alone
        (:588-594) --, nMatches including the frame's existing good map points (:566-572))."""
        search = area_search or (lambda *a: ctx.search_in_area(slot, *a))
        has = np.array(frame_has_good_mp, bool).copy()
        n_matches = 0 if bFuse else int(has.sum())
        idx = np.flatnonzero(np.asarray(usable, bool))
        if idx.size == 0:
            return [], n_matches
        lvl = np.asarray(level, np.int32)[idx]
        sf2 = np.asarray(scale_factors if scale_factors is not None else ctx.scale_factors(), np.float32) ** 2
        base = np.where(np.asarray(cos_theta, np.float32)[idx] > np.float32(0.998), np.float32(2.5), np.float32(4.0))
        radius = ((base * np.float32(th)).astype(np.float32) * sf2[lvl]).astype(np.float32)
        lo = np.maximum(0, lvl - 1).astype(np.int8)
        hi = np.minimum(n_levels - 1, lvl + 1).astype(np.int8)
        bi, bd, sd, nc = search(np.asarray(uv, np.float32)[idx], radius, lo, hi, np.asarray(mp_desc)[idx], None)
        ratio = bd.astype(np.float32) / sd.astype(np.float32)
        ok = (nc > 0) & (bd < self.mnMinThreshold) & (ratio < np.float32(self.mfRatio))
        out = []
        for k in np.flatnonzero(ok):
            f = int(bi[k])
            if bFuse:
                out.append((f, int(idx[k]), int(bd[k])))
                n_matches += 1
            elif not has[f]:
                has[f] = True                                           # pframe->setMapPoint(bestMatch.first, pMp)
                out.append((f, int(idx[k])))
                n_matches += 1
        return out, n_matches

    def searchByBow(self, ctx, desc_f, desc_kf, featvec_f, featvec_kf, good_f, inmap_f, good_kf, inmap_kf, angles_f=None, angles_kf=None,
                    bAddMPs=False, bLoop=False, best_match=None):
        """ORBMatcher::searchByBow (ORBMatcher.cc:170-253) given the two DBoW feature vectors as {node id: [feature ids]} (the BoW
        transform itself stays with DBoW3).  good_* = map point non-null and not bad, inmap_* = isInMap().  Returns the matches
        [(frame idx, keyframe idx, distance)] after the threshold / ratio test and verifyAngle (if mbCheckOri)."""
        match = best_match or (lambda q, t, off, cand: ctx.match_bruteforce(q, t, off, cand))
        good_f, inmap_f, good_kf, inmap_kf = (np.asarray(a, bool) for a in (good_f, inmap_f, good_kf, inmap_kf))
        q_ids, offs, cands = [], [0], []
        for node in sorted(set(featvec_f) & set(featvec_kf)):            # the merge walk over the two ordered maps (:183-189)
            if bAddMPs:
                cf = [p for p in featvec_f[node] if not (good_f[p] and inmap_f[p])]
            elif bLoop:
                cf = list(featvec_f[node])
            else:
                cf = [p for p in featvec_f[node] if not good_f[p]]
            for pk in featvec_kf[node]:
                g = good_kf[pk]
                if bAddMPs:
                    if g and inmap_kf[pk]:
                        continue
                elif not bLoop and not g:
                    continue
                if not cf:
                    continue
                q_ids.append(pk)
                cands += cf
                offs.append(len(cands))
        if not q_ids:
            return []
        bi, bd, sd = match(np.asarray(desc_kf)[q_ids], np.asarray(desc_f), np.asarray(offs, np.uint32), np.asarray(cands, np.uint32))
        ratio = bd.astype(np.float32) / sd.astype(np.float32)
        keep = ~((bd > self.mnMinThreshold) | (ratio > np.float32(self.mfRatio)))
        matches = [(int(bi[k]), int(q_ids[k]), int(bd[k])) for k in np.flatnonzero(keep)]
        if self.mbCheckOri and angles_f is not None and angles_kf is not None:
            matches = self.verifyAngle(matches, angles_f, angles_kf)
        return matches

    @classmethod
    def verifyAngle(cls, matches, angles1, angles2):
        """ORBMatcher::verifyAngle (ORBMatcher.cc:1013-1051): 30-bin histogram of angle differences (float arithmetic), the three
        largest bins survive (first maximum wins ties, empty bins never chosen); output ordered by bin id, then input order."""
        hist = [[] for _ in range(cls.mnBinNum)]
        for m in matches:
            diff = np.float32(angles1[m[0]]) - np.float32(angles2[m[1]])
            diff = diff if diff >= 0 else np.float32(360) + diff
            b = int(diff / np.float32(360 // cls.mnBinNum))             # 360 / mnBinNum is an int division (= 12)
            if b == 30:
                b = 0
            hist[b].append(m)
        good = set()
        for _ in range(cls.mnBinChoose):
            best, best_id = 0, None
            for i, h in enumerate(hist):
                if i not in good and len(h) > best:
                    best, best_id = len(h), i
            if best_id is not None:
                good.add(best_id)
        return [m for i in sorted(good) for m in hist[i]]

    def searchByStereo(self, frame: "StereoFrontEnd", fx: float, bf: float):
        """ORBMatcher::searchByStereo (ORBMatcher.cc:18-81) on the device-resident features of `frame`.
        Returns (n_matches, right_u, depth) with -1 where unmatched."""
        nm, ru, dp, _, _ = frame.ctx.stereo_match(0, 1, fx, bf)
        n = len(frame.left[0])
        return nm, ru[:n], dp[:n]


class StereoFrontEnd:
    """Frame::Frame(stereo) + Frame::createStereo (Frame.cc:85-111, Frame.h:313-322): both extractions and
    the stereo match for one pair, through one context."""

    def __init__(self, left, right, nFeatures=2000, nLevels=8, scale=1.2, maxThresh=20, minThresh=7, fx=718.856, bf=386.1448,
                 briefFp=None, device_id=0):
        left = np.ascontiguousarray(left, np.uint8)
        right = np.ascontiguousarray(right, np.uint8)
        if left.shape != right.shape:
            raise ValueError("left/right image sizes differ")
        pairs = load_brief_template(briefFp) if briefFp else None
        h, w = left.shape
        key = (w, h, nFeatures, nLevels, float(scale), maxThresh, minThresh, pairs.tobytes() if pairs is not None else None,
               device_id, 2)
        self.ctx = _CtxCache.get(key, width=w, height=h, n_features=nFeatures, n_levels=nLevels, scale_factor=scale,
                                 fast_hi=maxThresh, fast_lo=minThresh, brief_pairs=pairs, device_id=device_id, max_images=2)
        self.left, self.right = self.ctx.extract_batch([left, right])
        self.mnN, ru, dp = ORBMatcher().searchByStereo(self, fx, bf)
        self.mvFeatsRightU, self.mvDepths = ru, dp


class Frame:
    """The steps Frame::Frame / Tracking::grabFrame run either side of the extractor (src/Frame.cc:91-159, src/Tracking.cc:55-68)."""

    @staticmethod
    def grabColor(ctx: Context, img, color_type: int):
        """cv::cvtColor(COLOR_RGB2GRAY if Camera.Color == 1 else COLOR_BGR2GRAY) + ORBExtractor::extract on slot 0"""
        return ctx.extract_color(img, color_type)

    @staticmethod
    def finishRGBD(ctx: Context, slot: int, camera: dict, depth, depth_scale: float):
        """Camera::undistortPoints + the depth / rightU lookup of the RGB-D constructor (Frame.cc:145-158)"""
        return ctx.frame_rgbd(slot, camera, depth, depth_scale)


class Optimizer:
    """Edge evaluation of the graph Optimizer::OptimizeLocalMap builds (Optimizer.cc:296-330)."""
    deltaMono = float(np.float32(np.sqrt(5.991)))    # Optimizer.cc:1084 (stored as float)
    deltaStereo = float(np.float32(np.sqrt(7.815)))  # Optimizer.cc:1085

    @staticmethod
    def OptimizePoseOnly(ctx: Context, Xw, meas, info, sigma2, pose, fx, fy, cx, cy, bf):
        """The g2o part of Optimizer::OptimizePoseOnly (Optimizer.cc:33-178) on the device: returns (edges - nBad, pose, inliers)."""
        return ctx.pose_only_optimize(Xw, meas, info, sigma2, pose, fx, fy, cx, cy, bf)

    @staticmethod
    def OptimizeLocalMap(ctx: Context, problem, pose_fixed=None, iters_first=5, iters_second=10):
        """The g2o part of Optimizer::OptimizeLocalMap (Optimizer.cc:336-391) on the graph `problem` describes (the arrays of
        orbfe_ba_problem): optimize(5) with Huber, level-1 / kernel-off re-classification, optimize(10), final chi2 test.
        Returns dict(poses, points, level, chi2, bad, iters); the map bookkeeping of :393-441 stays with the caller."""
        return ctx.ba_local_optimize(problem, pose_fixed, iters_first, iters_second)

    @staticmethod
    def evalEdges(ctx: Context, poses, points, edge_pose, edge_point, meas, is_stereo, info, huber_delta, fx, fy, cx, cy, bf,
                  jacobians=True):
        return ctx.ba_eval_edges(poses, points, edge_pose, edge_point, meas, is_stereo, info, huber_delta, fx, fy, cx, cy, bf,
                                 jacobians)
