"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called through the C-ABI, against the CPU
oracle on the same seeded inputs -- bit-exact for every integer/byte/index output (pyramid planes, FAST
candidates, keypoints, angles, descriptors, match indices, right_u/depth floats), <= 1e-9 relative for the fp64 BA
edge outputs -- plus size-independent properties at the full batch size."""
import hashlib
import json
import os

import numpy as np
import pytest

from orb_slam2_ros2_amd import ba_synth, synth

pytestmark = pytest.mark.gpu

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.json")))
FX, BF = 718.856, 718.856 * 0.537166


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def lib():
    from orb_slam2_ros2_amd import _lib
    return _lib


@pytest.fixture(scope="module")
def kitti_ctx(lib):
    ctx = lib.Context(1241, 376, max_images=12)
    yield ctx
    ctx.close()


def assert_image_parity(ctx, slot, ex, gk, gd, n_levels, check_planes=True):
    """every intermediate and final product of one image: GPU (ctx, slot) vs oracle extractor `ex`"""
    ok, od = ex.extract()
    for l in range(n_levels):
        if check_planes:
            for blurred in (False, True):
                a, b = ctx.pyramid(slot, l, blurred), ex.plane(l, blurred)
                assert a.shape == b.shape and np.array_equal(a, b), f"level {l} blurred={blurred}: {(a != b).sum()} px differ"
        gc, oc = ctx.debug_candidates(slot, l), ex.candidates(l)
        assert gc.shape == oc.shape and np.array_equal(gc, oc), f"level {l} FAST candidates differ ({gc.shape} vs {oc.shape})"
    assert len(gk) == len(ok), f"keypoint count {len(gk)} vs {len(ok)}"
    for fld in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(gk[fld].view(np.int32), ok[fld].view(np.int32)), f"keypoint field {fld}"
    assert np.array_equal(gd, od), "descriptors"
    return ok, od



@pytest.mark.parametrize("slots,env", [(2, "1"), (40, None), (2, None)])
def test_a_pair_with_and_without_sharded_candidate_lists(orc, lib, monkeypatch, slots, env):
    """A frame or two in a context of <= 16 slots: k_fast appends a level's candidates to sixteen shards and the small-launch quadtree reads
    them (r6).  ORBFE_FAST_SHARDS=1, or a context of more slots, keeps one list per level on the same launches -- all three against the oracle,
    candidate sets included (orbfe_debug_candidates gathers the shards)."""
    if env is not None:
        monkeypatch.setenv("ORBFE_FAST_SHARDS", env)
    ctx = lib.Context(1241, 376, max_images=slots)
    try:
        L, R = synth.stereo_pair(11)
        (lk, ld), (rk, rd) = ctx.extract_batch([L, R])
        assert_image_parity(ctx, 0, orc.extractor(L), lk, ld, 8, check_planes=False)
        assert_image_parity(ctx, 1, orc.extractor(R), rk, rd, 8, check_planes=False)
        k1, d1 = ctx.extract(R)   # one image alone, into the slot the left image used
        assert_image_parity(ctx, 0, orc.extractor(R), k1, d1, 8, check_planes=False)
    finally:
        ctx.close()

@pytest.mark.parametrize("f", [0, 1, 7])
def test_kitti_stereo_frame_bit_exact_and_golden(orc, kitti_ctx, f):
    L, R = synth.stereo_pair(f)
    (lk, ld), (rk, rd) = kitti_ctx.extract_batch([L, R])
    nm, ru, dp, br, bd = kitti_ctx.stereo_match(0, 1, FX, BF)
    exl, exr = orc.extractor(L), orc.extractor(R)   # libm mode: exactly what the reference calls
    okl, odl = assert_image_parity(kitti_ctx, 0, exl, lk, ld, 8)
    okr, odr = assert_image_parity(kitti_ctx, 1, exr, rk, rd, 8)
    om, oru, odp, obr, obd = exl.stereo_match(exr, okl, odl, okr, odr, FX, BF)
    n = len(okl)
    assert nm == om
    assert np.array_equal(ru[:n].view(np.int64), oru.view(np.int64)) and np.array_equal(dp[:n].view(np.int64), odp.view(np.int64))
    assert np.array_equal(br[:n], obr) and np.array_equal(bd[:n], obd)
    g = G["frames"][f"kitti_{f}"]
    assert (sha(lk), sha(ld), sha(rk), sha(rd)) == (g["lk_sha"], g["ld_sha"], g["rk_sha"], g["rd_sha"])
    assert sha(ru[:n]) == g["right_u_sha"] and sha(dp[:n]) == g["depth_sha"] and nm == g["n_matches"]


def test_tum_geometry_1000_features(orc, lib):
    img = synth.mono_image(0)
    ctx = lib.Context(640, 480, n_features=1000, max_images=1)
    k, d = ctx.extract(img)
    assert_image_parity(ctx, 0, orc.extractor(img, n_features=1000), k, d, 8)
    g = G["frames"]["tum_0"]
    assert (len(k), sha(k), sha(d)) == (g["n"], g["k_sha"], g["d_sha"])
    assert [ctx.level_info(l).quota for l in range(8)] == g["quotas"]
    ctx.close()


@pytest.mark.parametrize("w,h,nf,nl,sc,hi,lo", [(700, 300, 500, 5, 1.3, 20, 7), (333, 257, 300, 4, 1.2, 40, 10),
                                                (1241, 376, 2000, 8, 1.2, 7, 20), (512, 512, 1500, 6, 1.5, 25, 25),
                                                (1000, 200, 777, 3, 1.1, 12, 3), (1241, 376, 64, 8, 1.2, 20, 7)])
def test_other_geometries_and_thresholds(orc, lib, w, h, nf, nl, sc, hi, lo):
    img, _ = synth.stereo_pair(11, w, h, n_rect=160)
    ctx = lib.Context(w, h, n_features=nf, n_levels=nl, scale_factor=sc, fast_hi=hi, fast_lo=lo, max_images=1)
    k, d = ctx.extract(img)
    ex = orc.extractor(img, n_features=nf, n_levels=nl, scale=sc, th_hi=hi, th_lo=lo)
    assert_image_parity(ctx, 0, ex, k, d, nl)
    for l in range(nl):
        li = ctx.level_info(l)
        assert (li.width, li.height, li.quota) == (ex.level_info(l)[0], ex.level_info(l)[1], ex.level_info(l)[3])
        assert li.scale == ex.level_info(l)[2]
    ctx.close()


def test_widest_image_the_records_can_hold(orc, lib):
    """4096 columns: the largest x a candidate record (12 bits) and the matcher's packed patch centre can carry -- one frame and one stereo
    match against the oracle, with keypoints in the last columns the border allows"""
    w, h = 4096, 400
    L, R = synth.stereo_pair(21, w, h, n_rect=500)
    ctx = lib.Context(w, h, n_features=1500, n_levels=4, max_images=2)
    (lk, ld), (rk, rd) = ctx.extract_batch([L, R])
    ex = orc.extractor(L, n_features=1500, n_levels=4)
    assert_image_parity(ctx, 0, ex, lk, ld, 4)
    assert lk["x"].max() > 4000
    nm, ru, dp, _, _ = ctx.stereo_match(0, 1, 2000.0, 800.0)
    ref = orc.stereo_frame(L, R, n_features=1500, n_levels=4, fx=2000.0, bf=800.0)
    n = len(lk)
    assert nm == ref["n_matches"] and nm > 0 and np.array_equal(ru[:n].view(np.int64), ref["right_u"].view(np.int64))
    assert np.array_equal(dp[:n].view(np.int64), ref["depth"].view(np.int64))
    ctx.close()
    with pytest.raises(lib.OrbfeError):
        lib.Context(4097, 400)


@pytest.mark.parametrize("seed", range(10))
def test_fuzzed_geometries(orc, lib, seed):
    """random image sizes / feature counts / pyramid shapes / thresholds: every LDS layout, tile edge and quota path"""
    r = np.random.default_rng(1000 + seed)
    nl = int(r.integers(1, 9))
    sc = float(r.choice([1.1, 1.2, 1.25, 1.5, 2.0]))
    # smallest level must keep a FAST region of >= 30 px (and the reference's own 38-px check)
    min_side = int(np.ceil(70 * sc ** (nl - 1))) + 2
    w = int(r.integers(max(min_side, 80), max(min_side, 80) + 1300))
    h = int(r.integers(max(min_side, 80), max(min_side, 80) + 600))
    w, h = min(w, 2400), min(h, 1400)
    nf = int(r.choice([50, 300, 1000, 2500, 6000]))
    hi = int(r.integers(5, 60))
    lo = int(r.integers(0, 30))
    img, _ = synth.stereo_pair(200 + seed, w, h, n_rect=int(r.integers(20, 400)))
    try:
        ex = orc.extractor(img, n_features=nf, n_levels=nl, scale=sc, th_hi=hi, th_lo=lo)
    except ValueError:
        with pytest.raises(lib.ImageSizeError):
            lib.Context(w, h, n_features=nf, n_levels=nl, scale_factor=sc, fast_hi=hi, fast_lo=lo, max_images=1)
        return
    # (no capacity limit any more: a level whose quota exceeds one CU's LDS -- nl = 1 with 6000 features here -- keeps its node table
    #  in global memory)
    ctx = lib.Context(w, h, n_features=nf, n_levels=nl, scale_factor=sc, fast_hi=hi, fast_lo=lo, max_images=1)
    k, d = ctx.extract(img)
    assert_image_parity(ctx, 0, ex, k, d, nl)
    ctx.close()


@pytest.mark.parametrize("w,h,nf", [(1920, 1080, 1000), (1241, 376, 800), (640, 480, 1000)])
def test_other_camera_geometries_single_frame_and_batch(orc, lib, w, h, nf):
    """Sizes and feature counts around the quadtree's pre-partition limits (r3: the node table is sized for the pre-partition's scratch --
    a 1080p level 0 takes 735 entries, nFeatures = 800 on a KITTI frame 405 where the quota alone asks for 185): one frame through the
    host-pointer path (four waves per tree) and 40 images through the device-batch path (one wave per tree group), against the oracle."""
    import torch
    L, R = synth.stereo_pair(3, w, h)
    exl, exr = orc.extractor(L, n_features=nf), orc.extractor(R, n_features=nf)
    ctx = lib.Context(w, h, n_features=nf, max_images=40)
    (lk, ld), (rk, rd) = ctx.extract_batch([L, R])
    assert_image_parity(ctx, 0, exl, lk, ld, 8)
    assert_image_parity(ctx, 1, exr, rk, rd, 8, check_planes=False)
    n = 20
    dl = torch.from_numpy(np.stack([L if i % 2 == 0 else R for i in range(n)])).cuda()
    dr = torch.from_numpy(np.stack([R if i % 2 == 0 else L for i in range(n)])).cuda()
    ctx.stereo_batch_device(dl.data_ptr(), dr.data_ptr(), w, w * h, n, FX, BF)
    ok_l, od_l = exl.extract()
    ok_r, od_r = exr.extract()
    for pair in (0, 1, n - 1):
        for eye in (0, 1):
            k, d = ctx.fetch_features(2 * pair + eye)
            is_left_image = (pair % 2 == 0) == (eye == 0)
            ok, od = (ok_l, od_l) if is_left_image else (ok_r, od_r)
            assert np.array_equal(k, ok) and np.array_equal(d, od), (pair, eye)
    ctx.close()


@pytest.mark.parametrize("nf", [2000, 120, 24])
def test_sparse_image_levels_with_fewer_candidates_than_quota_yield_nothing(orc, lib, nf):
    img, _ = synth.stereo_pair(5, sparse=True)
    ctx = lib.Context(1241, 376, n_features=nf, max_images=1)
    k, d = ctx.extract(img)
    ok, _ = assert_image_parity(ctx, 0, orc.extractor(img, n_features=nf), k, d, 8, check_planes=False)
    if nf == 2000:
        assert len(k) == 0   # quirk Q3 on every level
    ctx.close()


def test_flat_and_noise_only_images(orc, lib):
    ctx = lib.Context(400, 300, n_features=500, n_levels=4, max_images=1)
    for img in (np.full((300, 400), 128, np.uint8), np.random.default_rng(0).integers(120, 127, (300, 400)).astype(np.uint8),
                np.random.default_rng(1).integers(0, 256, (300, 400)).astype(np.uint8),
                # salt and pepper / a checkerboard with jitter: almost every pixel passes FAST's necessary test for BOTH
                # polarities (stresses the dual-polarity list of k_fast and its overflow path)
                (np.random.default_rng(2).integers(0, 2, (300, 400)) * 255).astype(np.uint8),
                ((np.indices((300, 400)).sum(0) & 1) * 200 + np.random.default_rng(3).integers(0, 40, (300, 400))).astype(np.uint8)):
        k, d = ctx.extract(img)
        assert_image_parity(ctx, 0, orc.extractor(img, n_features=500, n_levels=4), k, d, 4)
    ctx.close()


@pytest.mark.parametrize("cpw", ["2", "5"])
def test_fast_cell_loop_on_noise_and_on_a_frame(orc, lib, cpw, monkeypatch):
    """k_fast's cell loop (several cells per wave: the next patch prefetched, a cell's records stored a cell later) is what large
    batches run; ORBFE_FAST_CPW (read at orbfe_create) forces it for single images, so that its rare branches see the pathological
    cells too: more than 64 kept maxima in a cell (stored at once, not deferred), the dual-polarity list and its overflow, empty
    cells between full ones."""
    monkeypatch.setenv("ORBFE_FAST_CPW", cpw)
    ctx = lib.Context(400, 300, n_features=500, n_levels=4, max_images=1)
    half = np.full((300, 400), 128, np.uint8)
    half[:, 200:] = (np.random.default_rng(7).integers(0, 2, (300, 200)) * 255).astype(np.uint8)  # flat cells beside salt-and-pepper ones
    for img in (np.random.default_rng(1).integers(0, 256, (300, 400)).astype(np.uint8),
                (np.random.default_rng(2).integers(0, 2, (300, 400)) * 255).astype(np.uint8),
                ((np.indices((300, 400)).sum(0) & 1) * 200 + np.random.default_rng(3).integers(0, 40, (300, 400))).astype(np.uint8), half):
        k, d = ctx.extract(img)
        assert_image_parity(ctx, 0, orc.extractor(img, n_features=500, n_levels=4), k, d, 4)
    ctx.close()
    ctx = lib.Context(1241, 376, max_images=2)
    L, R = synth.stereo_pair(6)
    (lk, ld), (rk, rd) = ctx.extract_batch([L, R])
    assert_image_parity(ctx, 0, orc.extractor(L), lk, ld, 8)
    assert_image_parity(ctx, 1, orc.extractor(R), rk, rd, 8)
    ctx.close()


def test_blur_variant_and_custom_template(orc, lib):
    img, _ = synth.stereo_pair(2, 640, 360, n_rect=150)
    r = np.random.default_rng(9)
    pat = r.integers(-13, 13, (256, 4)).astype(np.int8)
    ctx = lib.Context(640, 360, n_features=800, blur_variant=1, brief_pairs=pat, max_images=1)
    k, d = ctx.extract(img)
    assert_image_parity(ctx, 0, orc.extractor(img, n_features=800, blur_variant=1, pattern=pat), k, d, 8)
    ctx.close()


def test_batch_slots_are_independent_and_equal_to_single_calls(orc, kitti_ctx):
    frames = [synth.stereo_pair(f)[s] for f in (20, 21, 22) for s in (0, 1)]
    frames.insert(3, frames[0])  # a duplicate must give identical bytes in another slot
    res = kitti_ctx.extract_batch(frames)
    assert np.array_equal(res[0][0], res[3][0]) and np.array_equal(res[0][1], res[3][1])
    for i in (1, 5):
        ok, od = orc.extractor(frames[i]).extract()
        assert np.array_equal(res[i][0], ok) and np.array_equal(res[i][1], od)
    single = kitti_ctx.extract(frames[2])
    assert np.array_equal(single[0], res[2][0]) and np.array_equal(single[1], res[2][1])


def test_device_resident_stereo_batch_equals_host_api(orc, kitti_ctx):
    import torch
    pairs = [synth.stereo_pair(f) for f in (30, 31, 32, 33)]
    dl = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
    dr = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    kitti_ctx.stereo_batch_device(dl.data_ptr(), dr.data_ptr(), 1241, 1241 * 376, 4, FX, BF)
    kitti_ctx.sync()
    for p, (L, R) in enumerate(pairs):
        lk, ld = kitti_ctx.fetch_features(2 * p)
        rk, rd = kitti_ctx.fetch_features(2 * p + 1)
        nm, ru, dp, br, bd = kitti_ctx.fetch_stereo(p)
        r = orc.stereo_frame(L, R, fx=FX, bf=BF)
        assert np.array_equal(lk, r["lk"]) and np.array_equal(ld, r["ld"]) and np.array_equal(rk, r["rk"]) and np.array_equal(rd, r["rd"])
        n = len(lk)
        assert nm == r["n_matches"] and np.array_equal(ru[:n], r["right_u"]) and np.array_equal(dp[:n], r["depth"])
    # padded source rows (stride > width) give the same result
    pad = torch.zeros((4, 376, 1280), dtype=torch.uint8, device="cuda")
    pad[:, :, :1241] = dl
    padr = torch.zeros_like(pad)
    padr[:, :, :1241] = dr
    kitti_ctx.stereo_batch_device(pad.data_ptr(), padr.data_ptr(), 1280, 1280 * 376, 4, FX, BF)
    kitti_ctx.sync()
    lk2, ld2 = kitti_ctx.fetch_features(6)
    assert np.array_equal(lk2, lk) and np.array_equal(ld2, ld)


def test_cfg3_bruteforce_2000x2000_bit_exact(orc, kitti_ctx):
    q, t = synth.descriptors_cfg3()
    bi, bd, sd = kitti_ctx.match_bruteforce(q, t)
    obi, obd, osd = orc.match_bruteforce(q, t)
    assert np.array_equal(bi, obi) and np.array_equal(bd, obd) and np.array_equal(sd, osd)
    g = G["cfg3"]
    assert (sha(bi), sha(bd), sha(sd)) == (g["best_idx_sha"], g["best_dist_sha"], g["second_sha"])


def test_bruteforce_candidate_lists_order_and_edge_cases(orc, kitti_ctx):
    r = np.random.default_rng(4)
    t = r.integers(0, 256, (300, 32), dtype=np.uint8)
    q = t[r.integers(0, 300, 90)] ^ (r.integers(0, 256, (90, 32), dtype=np.uint8) & r.integers(0, 2, (90, 32), dtype=np.uint8) * 3)
    lens = r.integers(0, 200, 90)
    lens[:6] = [0, 1, 2, 63, 64, 65]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    cand = np.concatenate([r.permutation(300)[:n] for n in lens] + [np.zeros(0, np.int64)]).astype(np.uint32)
    bi, bd, sd = kitti_ctx.match_bruteforce(q, t, offs, cand)
    for i in range(90):
        c = cand[offs[i]:offs[i + 1]].astype(np.int64)
        if len(c) == 0:
            assert (bi[i], bd[i], sd[i]) == (-1, 2**31 - 1, 2**31 - 1)
        else:
            assert (bi[i], bd[i], sd[i]) == orc.best_match(q[i], t, c)[:3], i
    # descending-distance scan: second best stays INT_MAX (quirk Q6)
    tt = np.zeros((3, 32), np.uint8)
    tt[0, :7] = 255
    tt[1, :5] = 255
    tt[2, :3] = 255
    b = kitti_ctx.match_bruteforce(np.zeros((1, 32), np.uint8), tt)
    assert (b[0][0], b[1][0], b[2][0]) == (2, 24, 2**31 - 1)
    b = kitti_ctx.match_bruteforce(np.zeros((1, 32), np.uint8), tt[::-1].copy())
    assert (b[0][0], b[1][0], b[2][0]) == (0, 24, 40)
    e = kitti_ctx.match_bruteforce(np.zeros((2, 32), np.uint8), np.zeros((0, 32), np.uint8))
    assert e[0].tolist() == [-1, -1]


def test_ba_edges_match_oracle(orc, kitti_ctx):
    p = ba_synth.make_problem()
    out = kitti_ctx.ba_eval_edges(**p)
    ref = orc.ba_eval_edges(**p)
    for k in ("error", "chi2", "rho", "j_point", "j_pose"):
        a, b = out[k], ref[k]
        assert a.shape == b.shape
        assert np.allclose(a, b, rtol=1e-9, atol=1e-12), k  # tolerance of north_star (BA residuals <= 1e-4) is far looser
    assert np.array_equal(out["depth_positive"], ref["depth_positive"])
    assert out["chi2"].sum() == pytest.approx(G["cfg5_ba"]["chi2_sum"], rel=1e-12)
    lean = kitti_ctx.ba_eval_edges(**p, jacobians=False)
    assert np.array_equal(lean["chi2"], out["chi2"])


def test_ba_normal_equation_blocks_match_oracle(orc, kitti_ctx):
    p = ba_synth.make_problem()
    nk = p["poses"].shape[0]
    fixed = np.zeros(nk, np.uint8)
    fixed[0] = 1
    fixed[30:] = 1          # 29 free keyframes + KF0 fixed + 30 fixed observers
    out = kitti_ctx.ba_build_system(**p, pose_fixed=fixed)
    ref = orc.ba_build_system(**p, pose_fixed=fixed)
    for k in ("Hpp", "bp", "Hll", "bl", "Hpl"):
        scale = np.abs(ref[k]).max()
        assert np.allclose(out[k], ref[k], rtol=1e-9, atol=1e-12 * scale), k
    assert not out["Hpp"][0].any() and not out["Hpp"][30:].any() and out["Hpp"][1:30].any()
    again = kitti_ctx.ba_build_system(**p, pose_fixed=fixed)
    assert all(np.array_equal(out[k], again[k]) for k in out)   # segmented sums, no atomics: run-to-run identical
    free = kitti_ctx.ba_build_system(**p, pose_fixed=None, want_hpl=False)
    assert free["Hpp"][45].any() and "Hpl" not in free


def test_error_paths(lib, kitti_ctx):
    with pytest.raises(lib.ImageSizeError):
        lib.Context(130, 100)           # level 7 below 38 px (ORBExtractor.cc:310-314)
    with pytest.raises(lib.OrbfeError) as ei:
        kitti_ctx.extract_batch([np.zeros((376, 1241), np.uint8)] * 13)
    assert ei.value.status == 4         # ECAPACITY
    with pytest.raises(ValueError):
        kitti_ctx.extract(np.zeros((100, 100), np.uint8))
    with pytest.raises(lib.OrbfeError):
        kitti_ctx.stereo_match(0, 99, FX, BF)
    with pytest.raises(lib.OrbfeError):
        lib.Context(1241, 376, n_levels=0)


def test_full_batch_properties(kitti_ctx, lib):
    """BASELINE-size batch (64 pairs = 128 images): size-independent properties instead of an oracle run."""
    import torch
    base = [synth.stereo_pair(f) for f in range(40, 44)]
    B = 64
    ctx = lib.Context(1241, 376, max_images=2 * B)
    dl = torch.from_numpy(np.stack([base[i % 4][0] for i in range(B)])).cuda()
    dr = torch.from_numpy(np.stack([base[i % 4][1] for i in range(B)])).cuda()
    ctx.stereo_batch_device(dl.data_ptr(), dr.data_ptr(), 1241, 1241 * 376, B, FX, BF)
    ctx.sync()
    ref = {}
    for p in range(B):
        lk, ld = ctx.fetch_features(2 * p)
        nm, ru, dp, br, bd = ctx.fetch_stereo(p)
        key = p % 4
        if key not in ref:
            ref[key] = (lk, ld, nm, ru, dp)
            assert len(lk) == 2000 and np.all(np.diff(lk["octave"]) >= 0)      # level-major order
            assert np.all(lk["size"] == 7) and np.all(lk["class_id"] == -1)
            assert (ru[:2000] >= 0).sum() == nm and nm > 50
            ok = ru[:2000] >= 0
            assert np.all(lk["x"][ok] - ru[:2000][ok] > 0) and np.all(dp[:2000][ok] > 0)
        else:  # idempotence: the same image gives the same bytes in every slot of the batch
            r = ref[key]
            assert np.array_equal(lk, r[0]) and np.array_equal(ld, r[1]) and nm == r[2]
            assert np.array_equal(ru, r[3]) and np.array_equal(dp, r[4])
    ctx.close()


def test_frontend_mirror_classes(orc):
    from orb_slam2_ros2_amd import ORBExtractor, ORBMatcher, StereoFrontEnd
    L, R = synth.stereo_pair(0)
    e = ORBExtractor(L, 2000, 8, 1.2, None, 20, 7)
    k, d = e.extract()
    ok, od = orc.extractor(L).extract()
    assert np.array_equal(k, ok) and np.array_equal(d, od)
    pyr = e.getPyramid()
    assert len(pyr) == 8 and np.array_equal(pyr[0], L) and pyr[7].shape == (105, 346)
    assert ORBMatcher.descDistance(d[0], d[1]) == orc.hamming(d[0], d[1])
    fr = StereoFrontEnd(L, R, fx=FX, bf=BF)
    assert fr.mnN == G["frames"]["kitti_0"]["n_matches"] and sha(fr.mvDepths) == G["frames"]["kitti_0"]["depth_sha"]


def test_cpp_host_mirror_matches_oracle(orc, tmp_path):
    """The C++ classes of orb_slam2_ros2_amd/host/orbfe_shim.hpp (reference signatures over the C-ABI)."""
    import subprocess
    from test_abi_and_host import _build_shim

    def fnv1a(b):
        h = 1469598103934665603
        for x in b:
            h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h

    exe = _build_shim(tmp_path)
    L, R = synth.stereo_pair(0)
    L.tofile(tmp_path / "L.raw")
    R.tofile(tmp_path / "R.raw")
    out = subprocess.run([exe, str(tmp_path / "L.raw"), str(tmp_path / "R.raw"), "1241", "376"], capture_output=True, text=True, check=True)
    nl, nr, nm, hk, hd, d01, self_found, proj, sim3, ba_err, ba_bad, ba_iters = out.stdout.split()
    s_self, s_all = map(int, sim3.split("/"))
    assert s_self == s_all and s_all > 0.85 * int(nl)             # ORBMatcher::searchBySim3 (uploaded keyframe features): self matches
    found, asked = map(int, self_found.split("/"))
    assert found == asked == 200                                   # ORBMatcher::searchInArea: every keypoint finds itself
    selfm, n_proj, n_kept = map(int, proj.split("/"))
    assert selfm == n_proj == n_kept and n_proj > 0.9 * int(nl)     # ORBMatcher::searchByProjection(frame, frame) + verifyAngle
    assert float(ba_err) < 1e-6 and int(ba_bad) == 0 and 2 <= int(ba_iters) <= 15   # Optimizer::OptimizeLocalMap reaches the truth
    ref = orc.stereo_frame(L, R, fx=FX, bf=BF)
    assert (int(nl), int(nr), int(nm)) == (len(ref["lk"]), len(ref["rk"]), ref["n_matches"])
    assert int(hk, 16) == fnv1a(ref["lk"].tobytes()) and int(hd, 16) == fnv1a(ref["ld"].tobytes())
    assert int(d01) == orc.hamming(ref["ld"][0], ref["ld"][1])


@pytest.mark.gpu
@pytest.mark.parametrize("nf", [7, 13, 58])
def test_tiny_nfeatures_can_yield_more_keypoints_than_asked(orc, lib, nf):
    """ORBExtractor.cc:292-300 rounds the quota of every level and lets only the last one absorb the difference: for small nFeatures
    the first seven levels alone ask for more (nFeatures = 7 -> 2 per level = 14).  The arrays are sized by orbfe_get_capacity."""
    img = synth.stereo_pair(2)[0]
    ctx = lib.Context(1241, 376, n_features=nf, max_images=1)
    assert ctx.n_features >= nf and ctx.requested_features == nf
    k, d = ctx.extract(img)
    ok, _ = assert_image_parity(ctx, 0, orc.extractor(img, n_features=nf), k, d, 8, check_planes=False)
    assert len(ok) > nf and len(ok) <= ctx.n_features
    ctx.close()


@pytest.mark.gpu
def test_back_to_back_device_batches_overlap_safely(orc, lib):
    """Consecutive orbfe_stereo_batch_device calls are pipelined (the stereo match of batch k runs on its own stream under the front
    of batch k+1, two pyramid buffers): the results of the LAST batch must be those of that batch alone, whatever ran before, and
    every other entry point must see a quiesced context."""
    import torch
    B = 16
    sets = [[synth.stereo_pair(4 * s + i) for i in range(4)] for s in range(3)]

    def dev(pairs):
        l = torch.from_numpy(np.stack([pairs[i % 4][0] for i in range(B)])).cuda()
        r = torch.from_numpy(np.stack([pairs[i % 4][1] for i in range(B)])).cuda()
        return l, r
    bufs = [dev(p) for p in sets]
    ctx = lib.Context(1241, 376, max_images=2 * B)
    for _ in range(2):                                          # twice: both pyramid buffers get used in both roles
        for l, r in bufs:                                       # three batches back to back, no synchronisation in between
            ctx.stereo_batch_device(l.data_ptr(), r.data_ptr(), 1241, 1241 * 376, B, FX, BF)
        for i in (0, 3, B - 1):                                 # the last batch (sets[2]) is what the slots hold
            ref = orc.stereo_frame(*sets[2][i % 4], fx=FX, bf=BF)
            k, d = ctx.fetch_features(2 * i)
            nm, ru, dp, _, _ = ctx.fetch_stereo(i)
            n = len(ref["lk"])
            assert np.array_equal(k, ref["lk"]) and np.array_equal(d, ref["ld"]) and nm == ref["n_matches"]
            assert np.array_equal(ru[:n], ref["right_u"]) and np.array_equal(dp[:n], ref["depth"])
            assert np.array_equal(ctx.pyramid(2 * i, 1, False), orc.extractor(sets[2][i % 4][0]).plane(1, False))
    # a host-pointer call right after a queued batch (joins the stereo stream first), then another batch
    l, r = bufs[0]
    ctx.stereo_batch_device(l.data_ptr(), r.data_ptr(), 1241, 1241 * 376, B, FX, BF)
    (k0, d0), (k1, d1) = ctx.extract_batch([sets[1][0][0], sets[1][0][1]])
    ref = orc.stereo_frame(*sets[1][0], fx=FX, bf=BF)
    assert np.array_equal(k0, ref["lk"]) and np.array_equal(d1, ref["rd"])
    nm, ru, dp, _, _ = ctx.stereo_match(0, 1, FX, BF)
    assert nm == ref["n_matches"]
    nm5 = ctx.fetch_stereo(5)[0]
    assert nm5 == orc.stereo_frame(*sets[0][5 % 4], fx=FX, bf=BF)["n_matches"]   # pair 5 of the batch is untouched by the host call
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nf", [2000, 600])
def test_clustered_corner_distributions_for_the_batched_quadtree(orc, lib, nf):
    """Corner distributions that bend the quadtree's pop order: texture confined to one corner of the frame (a child of an early pop
    outnumbers every other node: the batched pops must cut their head there), two clusters of very different density, a dense
    column on a strip boundary, and isolated dots (single-point nodes that are halved until they vanish, ties everywhere).
    Selection, order, angles and descriptors must equal the CPU restatement's; with ORBFE_QT_BATCH=0 the same holds."""
    W, H = 1241, 376
    rng = np.random.default_rng(21)
    noise = lambda h, w, a=0, b=256: rng.integers(a, b, (h, w)).astype(np.uint8)
    imgs = []
    a = np.full((H, W), 110, np.uint8)
    a[40:200, 60:360] = noise(160, 300)                      # everything in the first root strip, upper half
    imgs.append(a)
    b = np.full((H, W), 90, np.uint8)
    b[30:340, 700:1200] = noise(310, 500)                    # dense cluster on the right ...
    b[60:120, 40:400:7] = 255                                # ... and a sparse comb of bright columns on the left
    imgs.append(b)
    c = np.full((H, W), 128, np.uint8)
    c[20:356, 300:322] = noise(336, 22)                      # a dense column around the boundary of the first two root strips
    c[180:196, 30:1210] = noise(16, 1180)                    # and a dense row through the middle (the first horizontal split line)
    imgs.append(c)
    d = np.full((H, W), 100, np.uint8)
    ys, xs = rng.integers(30, H - 30, 2600), rng.integers(30, W - 30, 2600)
    for y, x in zip(ys, xs):                                  # isolated 3x3 dots: FAST corners a few pixels apart at most
        d[y - 1:y + 2, x - 1:x + 2] = 230
    imgs.append(d)
    ctx = lib.Context(W, H, n_features=nf, max_images=1)
    total = 0
    for img in imgs:
        k, dsc = ctx.extract(img)
        ok, _ = assert_image_parity(ctx, 0, orc.extractor(img, n_features=nf), k, dsc, 8, check_planes=False)
        total += len(ok)
    assert total > nf  # the cases do select keypoints (not four empty results)
    ctx.close()


@pytest.mark.gpu
def test_host_calls_between_device_batches_use_the_current_pyramid_buffer(orc, lib):
    """A pipelined device batch swaps the context's two pyramid buffers; a host-pointer extract that replays a cached hipGraph must
    write the buffer the following stereo_match / get_pyramid read (the graph cache is keyed on the buffer)."""
    import torch
    B = 16
    pairs = [synth.stereo_pair(60 + i) for i in range(4)]
    l = torch.from_numpy(np.stack([pairs[i % 4][0] for i in range(B)])).cuda()
    r = torch.from_numpy(np.stack([pairs[i % 4][1] for i in range(B)])).cuda()
    ctx = lib.Context(1241, 376, max_images=2 * B)
    probes = [synth.stereo_pair(70 + i) for i in range(4)]
    for step, (L, R) in enumerate(probes):
        if step:  # an odd number of swaps between two host calls, then an even one
            for _ in range(1 if step % 2 else 2):
                ctx.stereo_batch_device(l.data_ptr(), r.data_ptr(), 1241, 1241 * 376, B, FX, BF)
        (k0, d0), (k1, d1) = ctx.extract_batch([L, R])
        ref = orc.stereo_frame(L, R, fx=FX, bf=BF)
        n = len(ref["lk"])
        assert np.array_equal(k0, ref["lk"]) and np.array_equal(d0, ref["ld"]) and np.array_equal(k1, ref["rk"])
        nm, ru, dp, _, _ = ctx.stereo_match(0, 1, FX, BF)
        assert nm == ref["n_matches"] and np.array_equal(ru[:n], ref["right_u"]) and np.array_equal(dp[:n], ref["depth"])
        assert np.array_equal(ctx.pyramid(1, 2, False), orc.extractor(R).plane(2, False))
    ctx.close()


@pytest.mark.gpu
def test_host_images_with_padded_rows(orc, lib):
    """orbfe_extract_batch takes the caller's row stride: a view into a wider buffer (a cv::Mat ROI) gives the result of the packed image."""
    L, R = synth.stereo_pair(9)
    big = np.random.default_rng(0).integers(0, 256, (376 + 6, 1241 + 123), dtype=np.uint8)
    big2 = big.copy()
    big[3:379, 50:1291] = L
    big2[3:379, 50:1291] = R
    ctx = lib.Context(1241, 376, max_images=2)
    (k0, d0), (k1, d1) = ctx.extract_batch([big[3:379, 50:1291], big2[3:379, 50:1291]])
    ref = orc.stereo_frame(L, R, fx=FX, bf=BF)
    assert np.array_equal(k0, ref["lk"]) and np.array_equal(d0, ref["ld"]) and np.array_equal(k1, ref["rk"]) and np.array_equal(d1, ref["rd"])
    assert ctx.stereo_match(0, 1, FX, BF)[0] == ref["n_matches"]
    ctx.close()


@pytest.mark.gpu
def test_stereo_row_table_edge_cases(orc, lib):
    """searchByStereo on inputs that stress the right image's row table (createRowIndexDB, ORBMatcher.cc:915-932): every keypoint
    inside one narrow horizontal band (rows with hundreds of candidates, most rows empty), an empty right or left image, and a right
    image whose content lies to the RIGHT of the left one (every candidate fails the u-range)."""
    W, H, NF, NL = 640, 240, 800, 6
    rng = np.random.default_rng(11)
    tex = rng.integers(0, 256, (H, W + 64)).astype(np.uint8)
    band = np.full((H, W + 64), 90, np.uint8)
    band[100:150] = tex[100:150]                                  # texture in 50 rows only
    blank = np.full((H, W), 77, np.uint8)
    cases = [(band[:, 20:20 + W], band[:, 32:32 + W]),           # dense band, disparity 12
             (tex[:, 0:W], tex[:, 9:9 + W]),                      # texture everywhere, disparity 9
             (band[:, 20:20 + W], blank), (blank, band[:, 20:20 + W]),
             (tex[:, 40:40 + W], tex[:, 0:W])]                    # negative disparity: nothing may match
    ctx = lib.Context(W, H, n_features=NF, n_levels=NL, max_images=2)
    for ci, (L, R) in enumerate(cases):
        L, R = np.ascontiguousarray(L), np.ascontiguousarray(R)
        (lk, ld), (rk, rd) = ctx.extract_batch([L, R])
        nm, ru, dp, br, bd = ctx.stereo_match(0, 1, 500.0, 50.0)
        assert (nm > 50) if ci < 2 else (nm == 0 if ci < 4 else nm < 5), (ci, nm)   # (random texture: a stray match can pass)
        exl, exr = orc.extractor(L, n_features=NF, n_levels=NL), orc.extractor(R, n_features=NF, n_levels=NL)
        okl, odl = exl.extract()
        okr, odr = exr.extract()
        assert len(lk) == len(okl) and len(rk) == len(okr)
        om, oru, odp, obr, obd = exl.stereo_match(exr, okl, odl, okr, odr, 500.0, 50.0)
        n = len(okl)
        assert nm == om
        assert np.array_equal(ru[:n].view(np.int64), oru.view(np.int64)) and np.array_equal(dp[:n].view(np.int64), odp.view(np.int64))
        assert np.array_equal(br[:n], obr) and np.array_equal(bd[:n], obd)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("w", [80, 81, 82, 83, 245, 246, 247, 248, 249, 250, 251, 252, 253, 309, 495, 496, 497, 498, 499, 744, 745, 1237, 1238, 1239, 1240])
def test_row_widths_around_the_blur_strip_and_word_boundaries(orc, lib, w):
    """k_blur rebuilds the pixels past the image border from neighbouring lanes (BORDER_REFLECT_101): every residue of the width
    mod 4, rows narrower than one 62-word strip, rows that end exactly at / one word past a strip boundary (the right-aligned last
    strip), and the resize regions' last column for the same widths; every plane of every level against the oracle."""
    h, nl = (96 if w < 1000 else 160), (2 if w < 100 else 3)  # (the coarsest level must keep one 30-px FAST cell; at most 16 root strips)
    img, _ = synth.stereo_pair(300 + w, w, h, n_rect=60)
    ctx = lib.Context(w, h, n_features=300, n_levels=nl, max_images=1)
    k, d = ctx.extract(img)
    assert_image_parity(ctx, 0, orc.extractor(img, n_features=300, n_levels=nl), k, d, nl)
    ctx.close()


@pytest.mark.gpu
def test_device_batch_whose_resize_reads_the_callers_images(orc, lib):
    """Batches of >= 32 images: k_resize_regions reads level 0 straight from the caller's device images while the copy-in runs beside it.
    Exact-size contiguous buffers (the last 16-byte unit of the last row of the last image must not be read past the buffer's end),
    rows padded to an odd stride, and a pitch with spare rows between the images; first, middle and last pair against the oracle."""
    import torch
    B = 16
    pairs = [synth.stereo_pair(400 + (f % 5)) for f in range(B)]
    ref = {f: orc.stereo_frame(*pairs[f], fx=FX, bf=BF) for f in (0, 7, B - 1)}
    ctx = lib.Context(1241, 376, max_images=2 * B)
    layouts = [(1241, 376 * 1241), (1244, 376 * 1244), (1251, 380 * 1251)]
    for stride, pitch in layouts:
        buf_l = torch.full((B * pitch,), 201, dtype=torch.uint8, device="cuda")
        buf_r = torch.full((B * pitch,), 202, dtype=torch.uint8, device="cuda")
        for p, (L, R) in enumerate(pairs):
            for buf, img in ((buf_l, L), (buf_r, R)):
                view = buf[p * pitch:p * pitch + 376 * stride].view(376, stride)
                view[:, :1241] = torch.from_numpy(img).cuda()
        # the strided copies above are kernels on torch's stream; the context works on streams of its own and takes the images as READY
        # (include/orbfe.h, orbfe_stereo_batch_device): without this the last image written can still be in flight when it is read
        torch.cuda.synchronize()
        ctx.stereo_batch_device(buf_l.data_ptr(), buf_r.data_ptr(), stride, pitch, B, FX, BF)
        ctx.sync()
        for f, r in ref.items():
            lk, ld = ctx.fetch_features(2 * f)
            rk, rd = ctx.fetch_features(2 * f + 1)
            nm, ru, dp, br, bd = ctx.fetch_stereo(f)
            assert np.array_equal(lk, r["lk"]) and np.array_equal(ld, r["ld"]) and np.array_equal(rk, r["rk"]) and np.array_equal(rd, r["rd"]), (stride, f)
            n = len(lk)
            assert nm == r["n_matches"] and np.array_equal(ru[:n], r["right_u"]) and np.array_equal(dp[:n], r["depth"]), (stride, f)
    ctx.close()


def test_no_limit_on_n_features_20000(orc):
    """The reference has no bound on nFeatures (ORBExtractor.cc:291-301: quotas are just rounded shares).  At 20 000 features the quotas of
    levels 0-2 (4340 / 3617 / 3014) exceed the node table one CU's LDS can hold: those trees keep their tables in global memory.  A dense
    frame (many more FAST corners than the quotas) so that the big trees really select; and the standard frame, where levels with fewer
    candidates than their quota come out empty (quirk Q3)."""
    from orb_slam2_ros2_amd._lib import Context
    W, H = 1241, 376
    ctx = Context(W, H, n_features=20000, max_images=2)
    assert ctx.n_features >= 20000
    for n_rect in (1500, 260):
        L, R = synth.stereo_pair(11, W, H, n_rect=n_rect)
        (lk, ld), (rk, rd) = ctx.extract_batch([L, R])
        ek, ed = orc.extractor(L, n_features=20000).extract()
        assert np.array_equal(lk, ek) and np.array_equal(ld, ed), n_rect
        if n_rect == 1500:
            assert len(lk) > 12000 and (lk["octave"] == 0).sum() > 4000   # level 0 really selected its ~4340
        ref = orc.stereo_frame(L, R, n_features=20000, fx=FX, bf=BF)
        nm, ru, dp, _, _ = ctx.stereo_match(0, 1, FX, BF)
        assert np.array_equal(rk, ref["rk"]) and nm == ref["n_matches"] and np.array_equal(ru[:len(lk)], ref["right_u"])
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,nl", [(1241, 376, 8), (752, 480, 8), (640, 480, 8), (335, 200, 3), (336, 200, 3), (337, 200, 3), (351, 193, 3), (352, 207, 3),
                                    (353, 208, 3), (383, 209, 3), (385, 191, 3), (1920, 1080, 8)])
def test_device_batch_blur_on_the_matrix_cores(orc, lib, w, h, nl):
    """Batches of >= 32 images blur on the integer matrix cores (k_blur_mfma.hip: both passes as banded v_mfma_i32_16x16x32_i8 products,
    BORDER_REFLECT_101 folded into the bands): every blurred plane of every level of the first, a middle and the last image against the
    oracle's cv::GaussianBlur restatement -- widths around the 48-column strips and 16-column blocks, heights around the 16-row tiles
    (a last tile of 1 ... 16 rows), flat-white and flat-black images among them (the byte planes' signed offsets at their extremes)."""
    import torch
    B = 16
    nf = 300 if w < 600 else 1000
    pairs = [synth.stereo_pair_content(500 + f, ("rect", "camera", "saturated")[f % 3], w, h) for f in range(B)]
    pairs[3] = (np.full((h, w), 255, np.uint8), np.zeros((h, w), np.uint8))
    ctx = lib.Context(w, h, n_features=nf, n_levels=nl, max_images=2 * B)
    dl = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
    dr = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    ctx.stereo_batch_device(dl.data_ptr(), dr.data_ptr(), w, w * h, B, 0.58 * w, 0.58 * w * 0.11)
    ctx.sync()
    for slot in (0, 1, 6, 7, 2 * B - 1):
        ex = orc.extractor(pairs[slot >> 1][slot & 1], n_features=nf, n_levels=nl)
        for l in range(nl):
            a, b = ctx.pyramid(slot, l, True), ex.plane(l, True)
            assert a.shape == b.shape and np.array_equal(a, b), f"slot {slot} level {l}: {(a != b).sum()} blurred px differ (first at {np.argwhere(a != b)[:3].tolist()})"
    ctx.close()
