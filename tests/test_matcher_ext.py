"""The remaining callers of ORBMatcher::getBestMatch (include/ORB_SLAM2/ORBMatcher.h:41-75): searchBySim3 x2 (src/ORBMatcher.cc:370-559),
fuse x2 / processFuseMps (:623-734), searchForTriangulation (:736-787), as host logic (orb_slam2_ros2_amd/matcher_ext.py) around the
batched area search.  CPU: the batched formulation with the oracle's search injected, against loops written one query at a time after
the reference (explicit candidate lists, getBestMatch per query, std::map / std::set bookkeeping).  GPU: the same through
orbfe_search_in_area_features (a KeyFrame's features uploaded per call), which must reproduce the CPU result."""
import numpy as np
import pytest

from orb_slam2_ros2_amd import ORBMatcher
from orb_slam2_ros2_amd.matcher_ext import _affine, _matvec, predict_level
from test_guided_wrappers import H, SF, W, _best, _cands

F32 = np.float32
CAM = (718.856, 718.856, 607.1928, 185.2157)
BOUNDS = (0.0, float(W), 0.0, float(H))


def _keyframe(orc, img, seed, shift=(0.0, 0.0, 0.0)):
    """a 'KeyFrame': oracle features + a map point behind most of them (back-projected at a random depth, then shifted)"""
    k, d = orc.extractor(img).extract()
    r = np.random.default_rng(seed)
    n = len(k)
    z = r.uniform(4, 40, n).astype(F32)
    pos = np.stack([(k["x"] - F32(CAM[2])) / F32(CAM[0]) * z, (k["y"] - F32(CAM[3])) / F32(CAM[1]) * z, z], 1).astype(F32) + np.array(shift, F32)
    good = r.random(n) < 0.8
    return dict(kps=k, desc=d, pos=pos, good=good, inmap=good & (r.random(n) < 0.9), max_dist=(z * F32(1.2 ** 7) * F32(1.2)).astype(F32),
                min_dist=(z * F32(0.3)).astype(F32))


@pytest.fixture(scope="module")
def kfs(orc, kitti_pair):
    L, R = kitti_pair
    # the same image twice with different map state: a projected point lands near its own feature in the other keyframe, so the searches
    # do find matches; the right image (features displaced by the disparity) is used where a miss-heavy target is wanted
    return _keyframe(orc, L, 1), _keyframe(orc, L, 2)


def _sim3():
    a = 0.01
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], F32)
    return F32(1.03), R, np.array([0.02, -0.01, 0.05], F32)


def _loop_sim3_project(orc, src, i, pose, sim3, dst, th, ratio_max):
    """ORBMatcher::SIM3Project (src/ORBMatcher.cc:370-414) for ONE map point"""
    s, R, t = sim3
    pc = _affine(1.0, pose[0], src["pos"][i], pose[1])
    pm = _affine(s, R, pc, t)
    if pm[2] <= 0:
        return None
    u = F32(F32(F32(CAM[0]) * F32(pm[0] / pm[2])) + F32(CAM[2]))
    v = F32(F32(F32(CAM[1]) * F32(pm[1] / pm[2])) + F32(CAM[3]))
    if not (u < BOUNDS[1] and v < BOUNDS[3] and u > BOUNDS[0] and v > BOUNDS[2]):
        return None
    d = F32(np.sqrt(F32(F32(F32(pm[0] * pm[0]) + F32(pm[1] * pm[1])) + F32(pm[2] * pm[2]))) / s)
    if not (d < src["max_dist"][i] and d > src["min_dist"][i]):
        return None
    o = predict_level(src["max_dist"][i], d, F32(np.log(SF[1])))
    cand = [c for c in _cands(dst["kps"], u, v, F32(th) * SF[o] * SF[o], o - 1, o + 1) if dst["good"][c] and dst["inmap"][c]]
    if not cand:
        return None
    bi, bd, ratio = _best(orc, src["desc"][i], dst["desc"], cand)
    return bi if (bd <= 50 and ratio <= F32(ratio_max)) else None


def _inject(orc, kfC, kfM):
    def run(tag, *a):
        t = kfM if tag == "M" else kfC
        return orc.search_in_area(t["kps"], t["desc"], W, H, *a)
    return run


def _expected_sim3_frames(orc, kfC, kfM, matches, Scm, poseC, poseM, th, ratio):
    s, R, t = Scm
    Smc = (F32(F32(1) / s), R.T.copy(), (-F32(F32(1) / s) * _matvec(R.T.copy(), t)).astype(F32))
    flagC, flagM = np.ones(len(kfC["kps"]), bool), np.ones(len(kfM["kps"]), bool)
    for q, tr in matches:
        flagC[q], flagM[tr] = False, False
    new = {}
    for ic in range(len(kfC["kps"])):
        if flagC[ic] and kfC["good"][ic] and kfC["inmap"][ic]:
            b = _loop_sim3_project(orc, kfC, ic, poseC, Smc, kfM, th, ratio)
            if b is not None and ic not in new:
                new[ic] = b
    for im in range(len(kfM["kps"])):
        if flagM[im] and kfM["good"][im]:
            b = _loop_sim3_project(orc, kfM, im, poseM, Scm, kfC, th, ratio)
            if b is not None and b not in new:
                new[b] = im
    return list(matches) + sorted(new.items())


def test_search_by_sim3_frames(orc, kfs):
    kfC, kfM = kfs
    I = (np.eye(3, dtype=F32), np.zeros(3, F32))
    matches = [(5, 7), (40, 41), (100, 90)]
    m = ORBMatcher(0.9)
    got = m.searchBySim3Frames(None, matches, _sim3(), I, I, kfC, kfM, 7.5, CAM, BOUNDS, SF, search_in=_inject(orc, kfC, kfM))
    want = _expected_sim3_frames(orc, kfC, kfM, matches, _sim3(), I, I, 7.5, 0.9)
    assert got == want and len(got) > len(matches) + 10 and got[:3] == matches


def test_search_by_sim3_map_points(orc, kfs):
    kfC, kfM = kfs
    r = np.random.default_rng(5)
    n = len(kfM["kps"])
    vd = np.tile(np.array([0, 0, 1], F32), (n, 1)) + r.normal(0, 0.4, (n, 3)).astype(F32)
    loop = dict(pos=kfM["pos"], view_dir=vd.astype(F32), desc=kfM["desc"], max_dist=kfM["max_dist"], min_dist=kfM["min_dist"],
                usable=kfM["good"] & kfM["inmap"], id=np.arange(1000, 1000 + n))
    matched = np.full(len(kfC["kps"]), -1, np.int64)
    matched[[3, 9, 50]] = [1002, 1010, 1500]                          # three features of pCurr already carry a loop map point
    Scw = _sim3()
    m = ORBMatcher(0.9)
    got, n_got = m.searchBySim3MapPoints(None, kfC, loop, matched, Scw, 10.0, CAM, BOUNDS, SF,
                                         search_in=lambda *a: orc.search_in_area(kfC["kps"], kfC["desc"], W, H, *a))
    s, R, t = Scw
    want, n_want = [], 3
    for i in range(n):
        if not loop["usable"][i] or int(loop["id"][i]) in (1002, 1010, 1500):
            continue
        pc = _affine(s, R, loop["pos"][i], t)
        if pc[2] <= 0:
            continue
        u = F32(F32(F32(CAM[0]) * F32(pc[0] / pc[2])) + F32(CAM[2]))
        v = F32(F32(F32(CAM[1]) * F32(pc[1] / pc[2])) + F32(CAM[3]))
        if not (u < BOUNDS[1] and v < BOUNDS[3] and u > 0 and v > 0):
            continue
        dws = F32(np.sqrt(np.float64(pc[0]) ** 2 + np.float64(pc[1]) ** 2 + np.float64(pc[2]) ** 2))
        d = F32(dws / s)
        if not (d < loop["max_dist"][i] and d > loop["min_dist"][i]):
            continue
        rv = _matvec(R, loop["view_dir"][i])
        if float(rv[0]) * float(pc[0]) + float(rv[1]) * float(pc[1]) + float(rv[2]) * float(pc[2]) < 0.5 * float(dws):
            continue
        o = predict_level(loop["max_dist"][i], d, F32(np.log(SF[1])))
        cand = _cands(kfC["kps"], u, v, F32(10.0) * SF[o] * SF[o], o - 1, o + 1)
        if not cand:
            continue
        bi, bd, ratio = _best(orc, loop["desc"][i], kfC["desc"], cand)
        if bd <= 50 and ratio <= F32(0.9):
            want.append((bi, i))
            n_want += 1
    assert got == want and n_got == n_want and len(got) > 10


def test_process_fuse_mps_decisions():
    f_good = np.array([1, 0, 1, 1, 1, 0], bool)
    f_id = np.array([10, -1, 12, 13, 14, -1])
    f_obs = np.array([5, 0, 2, 9, 3, 0])
    v_good = np.array([1, 1, 0, 1, 1], bool)
    v_id = np.array([20, 21, 22, 13, 24])
    v_obs = np.array([3, 1, 7, 9, 8])
    matches = [(0, 0, 5), (1, 1, 9), (2, 2, 1), (3, 3, 0), (4, 4, 2), (5, 0, 3)]
    act, n = ORBMatcher.processFuseMps(matches, f_good, f_id, v_good, v_id, f_obs, v_obs)
    # 0: both good, obs 5 >= 3 -> keep the keyframe's; 1: keyframe has none -> add; 2: projected point bad -> skip; 3: the same point -> skip;
    # 4: obs 3 < 8 -> keep the projected one; 5: add
    assert act == [("replace", 10, 20), ("add", 1, 1), ("replace", 24, 14), ("add", 5, 0)] and n == 4
    act, n = ORBMatcher.processFuseMps(matches, f_good, f_id, v_good, v_id, f_obs, v_obs, bLoop=True)
    assert act == [("replace", 20, 10), ("add", 1, 1), ("replace", 24, 14), ("add", 5, 0)] and n == 4   # bLoop: the loop point always wins


def test_fuse_map_points_and_frames(orc, kfs):
    kf1, kf2 = kfs
    r = np.random.default_rng(8)
    n1, n2 = len(kf1["kps"]), len(kf2["kps"])
    st1 = dict(good=kf1["good"], id=np.where(kf1["good"], np.arange(n1), -1), obs=r.integers(1, 9, n1))
    st2 = dict(good=kf2["good"], id=np.where(kf2["good"], 5000 + np.arange(n2), -1), obs=r.integers(1, 9, n2))
    st2["id"][:40] = st1["id"][:40]                                     # some points are shared already
    inject = lambda *a: orc.search_in_area(kf1["kps"], kf1["desc"], W, H, *a)
    m = ORBMatcher(0.8)
    # fuse(pkf1, pkf2): every feature of kf2 with a usable, visible map point looks around its own position in kf1
    inv = r.random(n2) < 0.9
    act, n = m.fuseFrames(None, kf1, st1, kf2["kps"], kf2["desc"], st2, inv, 0.0, 0.54, SF, search_in=inject)
    want = []
    for idx in range(n2):
        if not st2["good"][idx] or not inv[idx]:
            continue
        o = int(kf2["kps"]["octave"][idx])
        cand = _cands(kf1["kps"], kf2["kps"]["x"][idx], kf2["kps"]["y"][idx], F32(3.0) * SF[o] * SF[o], max(0, o - 1), min(o + 1, 7))
        if cand:
            bi, bd, ratio = _best(orc, kf2["desc"][idx], kf1["desc"], cand)
            if ratio < F32(0.8) and bd < 50:
                want.append((bi, idx, bd))
    assert (act, n) == ORBMatcher.processFuseMps(want, st1["good"], st1["id"], st2["good"], st2["id"], st1["obs"], st2["obs"]) and n > 5
    # fuse(pkf1, mapPoints): the points kf1 already holds are dropped first
    mps = dict(id=st2["id"], good=st2["good"], obs=st2["obs"], usable=st2["good"] & inv,
               uv=np.stack([kf2["kps"]["x"], kf2["kps"]["y"]], 1), level=kf2["kps"]["octave"], cos_theta=r.uniform(0.6, 1, n2).astype(F32),
               desc=kf2["desc"])
    act2, n2f = m.fuseMapPoints(None, kf1, st1, mps, th=3.0, scale_factors=SF, search_in=inject)
    held = set(int(i) for i, g in zip(st1["id"], st1["good"]) if g)
    assert n2f > 5 and all(a[0] != "add" or int(mps["id"][a[2]]) not in held for a in act2)
    assert not any(a[0] == "replace" and (a[1] in held) and (a[2] in held) for a in act2)


def test_epipolar_filter_of_search_for_triangulation():
    """two cameras 0.5 m apart looking at the same points: true correspondences pass, a match moved 20 px off its epipolar line does not"""
    r = np.random.default_rng(3)
    n = 60
    P = np.stack([r.uniform(-5, 5, n), r.uniform(-1.5, 1.5, n), r.uniform(6, 30, n)], 1)
    K = np.array([[CAM[0], 0, CAM[2]], [0, CAM[1], CAM[3]], [0, 0, 1]], np.float64)
    T1, T2 = np.eye(4, dtype=F32), np.eye(4, dtype=F32)
    T2[0, 3] = -0.5                                                     # camera 2 sits 0.5 m to the right
    from orb_slam2_ros2_amd._lib import KP_DTYPE
    k1, k2 = np.zeros(n, KP_DTYPE), np.zeros(n, KP_DTYPE)
    for i in range(n):
        a, b = K @ P[i], K @ (P[i] + np.array([-0.5, 0, 0]))
        k1[i]["x"], k1[i]["y"] = a[0] / a[2], a[1] / a[2]
        k2[i]["x"], k2[i]["y"] = b[0] / b[2], b[1] / b[2]
        k1[i]["octave"], k2[i]["octave"] = i % 8, (i + 1) % 8
    k2["y"][10] += 20
    k1["y"][20] -= 60                                                    # (octave 4: the gate is 5.991 * 1.2^8 = 25.8 px)
    matches = [(i, i, 0) for i in range(n)]
    Kinv = np.linalg.inv(K).astype(F32)
    inv = lambda T: np.linalg.inv(T.astype(np.float64)).astype(F32)
    kept = ORBMatcher.epipolarFilter(matches, k1, k2, T1, inv(T1), T2, inv(T2), Kinv, SF)
    assert [m[0] for m in kept] == [i for i in range(n) if i not in (10, 20)]


@pytest.mark.gpu
def test_device_sim3_and_fuse_equal_the_injected_oracle_path(orc, kfs):
    from orb_slam2_ros2_amd._lib import Context
    kfC, kfM = kfs
    ctx = Context(W, H, max_images=2)
    I = (np.eye(3, dtype=F32), np.zeros(3, F32))
    m = ORBMatcher(0.9)
    matches = [(5, 7), (40, 41)]
    dev = m.searchBySim3Frames(ctx, matches, _sim3(), I, I, kfC, kfM, 7.5, CAM, BOUNDS, SF)
    cpu = m.searchBySim3Frames(None, matches, _sim3(), I, I, kfC, kfM, 7.5, CAM, BOUNDS, SF, search_in=_inject(orc, kfC, kfM))
    assert dev == cpu and len(dev) > 12
    # the raw device search against an uploaded feature set, with an exclusion mask and octave windows that leave the pyramid
    r = np.random.default_rng(2)
    nq = 300
    q = r.integers(0, len(kfM["kps"]), nq)
    qxy = np.stack([kfM["kps"]["x"][q], kfM["kps"]["y"][q]], 1) + r.normal(0, 3, (nq, 2)).astype(F32)
    rad = r.uniform(2, 40, nq).astype(F32)
    lo, hi = r.integers(-1, 6, nq).astype(np.int8), r.integers(2, 9, nq).astype(np.int8)
    ex = (r.random(len(kfC["kps"])) < 0.3).astype(np.uint8)
    a = ctx.search_in_area_features(kfC["kps"], kfC["desc"], qxy, rad, lo, hi, kfM["desc"][q], ex)
    b = orc.search_in_area(kfC["kps"], kfC["desc"], W, H, qxy, rad, lo, hi, kfM["desc"][q], ex)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and (a[3] > 0).sum() > 100
    st1 = dict(good=kfC["good"], id=np.where(kfC["good"], np.arange(len(kfC["kps"])), -1), obs=np.ones(len(kfC["kps"]), int))
    st2 = dict(good=kfM["good"], id=np.where(kfM["good"], 5000 + np.arange(len(kfM["kps"])), -1), obs=np.ones(len(kfM["kps"]), int))
    inv = np.ones(len(kfM["kps"]), bool)
    dev_f = m.fuseFrames(ctx, kfC, st1, kfM["kps"], kfM["desc"], st2, inv, 0.0, 0.54, SF)
    cpu_f = m.fuseFrames(None, kfC, st1, kfM["kps"], kfM["desc"], st2, inv, 0.0, 0.54, SF,
                         search_in=lambda *a: orc.search_in_area(kfC["kps"], kfC["desc"], W, H, *a))
    assert dev_f == cpu_f and dev_f[1] > 5
    # undistorted KeyFrame features may leave the image (Frame.cc:106) or be non-finite: the grid kernel must clamp them into the border
    # cells instead of writing outside its LDS counters / scratch lists (the reference indexes its cell vector unchecked there)
    wild = kfC["kps"].copy()
    wild["x"][::7] -= F32(200.0)
    wild["y"][::11] -= F32(90.0)
    wild["x"][3::13] += F32(W)
    wild["y"][5::17] += F32(2 * H)
    wild["x"][1::97] = np.nan
    wild["y"][2::101] = np.inf
    wild["x"][4::103] = -np.inf
    a = ctx.search_in_area_features(wild, kfC["desc"], qxy, rad * 3, lo, hi, kfM["desc"][q], ex)
    b = orc.search_in_area(wild, kfC["desc"], W, H, qxy, rad * 3, lo, hi, kfM["desc"][q], ex)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and (a[3] > 0).sum() > 100
    again = ctx.search_in_area_features(kfC["kps"], kfC["desc"], qxy, rad, lo, hi, kfM["desc"][q], ex)   # nothing was corrupted
    assert all(np.array_equal(x, y) for x, y in zip(again, orc.search_in_area(kfC["kps"], kfC["desc"], W, H, qxy, rad, lo, hi, kfM["desc"][q], ex)))
    # the extended form: undistorted frame bounds that differ from the image (a distorted camera: VirtualFrame::mfMinU..mfMaxV size the
    # grid and clip the box) and the per-feature counts of "excluded but in the window" (addMatchInTrack in searchByProjection)
    for bnd in ((0.0, float(W), 0.0, float(H)), (-23.7, W + 41.2, -11.3, H + 17.9), (5.5, W - 60.25, 2.0, H - 50.5)):
        a = ctx.search_in_area_features(kfC["kps"], kfC["desc"], qxy, rad, lo, hi, kfM["desc"][q], ex, bounds=bnd, want_hits=True)
        b = orc.search_in_area_ex(kfC["kps"], kfC["desc"], bnd, qxy, rad, lo, hi, kfM["desc"][q], ex)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), bnd
        assert a[4].sum() > 50 and (a[4][ex == 0] == 0).all()
    nohit = ctx.search_in_area_features(kfC["kps"], kfC["desc"], qxy, rad, lo, hi, kfM["desc"][q], None, want_hits=True)
    assert not nohit[4].any()
    from orb_slam2_ros2_amd._lib import OrbfeError
    with pytest.raises(OrbfeError):
        ctx.search_in_area_features(kfC["kps"], kfC["desc"], qxy, rad, lo, hi, kfM["desc"][q], ex, bounds=(10.0, 5.0, 0.0, 100.0))
    neg = kfC["kps"].copy()
    neg["octave"][10] = -1
    with pytest.raises(OrbfeError):
        ctx.search_in_area_features(neg, kfC["desc"], qxy[:4], rad[:4], lo[:4], hi[:4], kfM["desc"][q[:4]])
    # an empty feature set and an empty query list are fine
    e = ctx.search_in_area_features(kfC["kps"][:0], kfC["desc"][:0], qxy[:4], rad[:4], lo[:4], hi[:4], kfM["desc"][q[:4]])
    assert e[0].tolist() == [-1] * 4 and e[3].tolist() == [0] * 4
    ctx.close()
