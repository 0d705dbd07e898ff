"""Frame-level glue: cvtColor RGB/BGR->gray, Camera::undistortPoints, the RGB-D depth / rightU lookup.
CPU: the oracle against independent numpy formulations and known answers.  GPU: the device kernels against the oracle, bit-exact
(bytes, and floats / doubles computed without FMA contraction on both sides)."""
import numpy as np
import pytest

from orb_slam2_ros2_amd import synth

# config/tum_config_f2.yaml (TUM freiburg2): intrinsics and distortion as float32, bf = fx * baseline
TUM = dict(fx=520.908620, fy=521.007327, cx=325.141442, cy=249.701764, k1=0.231222, k2=-0.784899, p1=-0.003257, p2=-0.000105, k3=0.917205,
           bf=40.0)


def _color_image(seed, w=640, h=480):
    g = synth.mono_image(seed, w, h)
    rng = np.random.default_rng(seed)
    img = np.stack([g, np.roll(g, 3, 1), (255 - g)], 2).astype(np.int32) + rng.integers(-6, 7, (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def _distort(xn, yn, c):
    r2 = xn * xn + yn * yn
    rad = 1 + c["k1"] * r2 + c["k2"] * r2 ** 2 + c["k3"] * r2 ** 3
    xd = xn * rad + 2 * c["p1"] * xn * yn + c["p2"] * (r2 + 2 * xn * xn)
    yd = yn * rad + c["p1"] * (r2 + 2 * yn * yn) + 2 * c["p2"] * xn * yn
    return xd, yd


def test_oracle_cvt_gray_known_answers_and_numpy(orc):
    img = _color_image(1, 64, 48)
    for order in (1, 2):
        r, g, b = (img[..., 0], img[..., 1], img[..., 2]) if order == 1 else (img[..., 2], img[..., 1], img[..., 0])
        want = ((r.astype(np.int64) * 4899 + g.astype(np.int64) * 9617 + b.astype(np.int64) * 1868 + 8192) >> 14).astype(np.uint8)
        assert np.array_equal(orc.cvt_gray(img, order), want)
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 20, 30]]], np.uint8)
    assert orc.cvt_gray(px, 1)[0].tolist() == [255, 0, 76, 150, 29, 18]     # 0.299 / 0.587 / 0.114 in 14-bit fixed point
    assert orc.cvt_gray(px, 2)[0].tolist() == [255, 0, 29, 150, 76, 22]
    # variant 1: the 15-bit coefficients (9798 / 19235 / 3735, >> 15) -- a selectable decision like the blur taps
    for order in (1, 2):
        r, g, b = (img[..., 0], img[..., 1], img[..., 2]) if order == 1 else (img[..., 2], img[..., 1], img[..., 0])
        want = ((r.astype(np.int64) * 9798 + g.astype(np.int64) * 19235 + b.astype(np.int64) * 3735 + 16384) >> 15).astype(np.uint8)
        assert np.array_equal(orc.cvt_gray(img, order, 1), want)
    assert orc.cvt_gray(px, 1, 1)[0].tolist() == [255, 0, 76, 150, 29, 18]
    big = _color_image(5, 320, 240)
    assert (orc.cvt_gray(big, 1, 0) != orc.cvt_gray(big, 1, 1)).any()        # the two do differ on real data (by one grey level)
    assert np.abs(orc.cvt_gray(big, 1, 0).astype(int) - orc.cvt_gray(big, 1, 1).astype(int)).max() <= 1


def test_oracle_undistort_inverts_the_distortion_model(orc):
    K = np.array([TUM[k] for k in ("fx", "fy", "cx", "cy")], np.float32)
    D = np.array([TUM[k] for k in ("k1", "k2", "p1", "p2", "k3")], np.float32)
    rng = np.random.default_rng(5)
    ideal = np.stack([rng.uniform(60, 580, 500), rng.uniform(60, 420, 500)], 1)       # undistorted pixels
    xn, yn = (ideal[:, 0] - TUM["cx"]) / TUM["fx"], (ideal[:, 1] - TUM["cy"]) / TUM["fy"]
    xd, yd = _distort(xn, yn, TUM)
    dist = np.stack([xd * TUM["fx"] + TUM["cx"], yd * TUM["fy"] + TUM["cy"]], 1).astype(np.float32)
    und = orc.undistort_points(dist, K, D)
    assert np.abs(und - ideal).max() < 0.05           # 5 fixed-point iterations: not exact, but far closer than the input
    assert np.abs(dist - ideal).max() > 1.0
    assert np.array_equal(orc.undistort_points(dist, K, np.zeros(5, np.float32)), dist)           # k1 == 0: early out (Camera.cc:31)
    centre = np.array([[TUM["cx"], TUM["cy"]]], np.float32)
    assert np.abs(orc.undistort_points(centre, K, D) - centre).max() < 1e-4                     # the principal point stays


def test_oracle_rgbd_lookup_truncates_indices_and_marks_missing_depth(orc):
    depth = np.zeros((480, 640), np.uint16)
    depth[100, 200] = 5000
    depth[101, 201] = 10000
    xy = np.array([[200.9, 100.9], [201.0, 101.0], [10.0, 10.0]], np.float32)     # (x, y): truncation, not rounding (Q10)
    xyu = xy + np.float32(0.5)
    d, ru = orc.rgbd_lookup(xy, xyu, depth, 5000.0, 40.0)
    assert d.tolist() == [1.0, 2.0, -1.0]
    assert ru[0] == float(np.float32(xyu[0, 0]) - np.float32(40.0) / np.float32(1.0)) and ru[1] == float(xyu[1, 0] - np.float32(20.0)) and ru[2] == -1.0
    df, _ = orc.rgbd_lookup(xy, xyu, (depth / np.float32(5000)).astype(np.float32), 1.0, 40.0)
    assert df.tolist() == [1.0, 2.0, -1.0]


@pytest.mark.gpu
@pytest.mark.parametrize("order,variant", [(1, 0), (2, 0), (1, 1), (2, 1)])
def test_device_color_frame_matches_oracle(orc, order, variant):
    from orb_slam2_ros2_amd import Frame
    from orb_slam2_ros2_amd._lib import Context
    img = _color_image(3)
    ctx = Context(640, 480, n_features=1000, max_images=1, gray_variant=variant)
    k, d = Frame.grabColor(ctx, img, order)
    gray = orc.cvt_gray(img, order, variant)
    assert np.array_equal(ctx.pyramid(0, 0, False), gray)
    ok, od = orc.extractor(gray, n_features=1000).extract()
    assert len(k) == len(ok) and np.array_equal(d, od)
    for f in ("x", "y", "angle", "response", "octave"):
        assert np.array_equal(k[f].view(np.int32), ok[f].view(np.int32))
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("depth_dtype", [np.uint16, np.float32])
def test_device_rgbd_tail_matches_oracle(orc, depth_dtype):
    from orb_slam2_ros2_amd import Frame
    from orb_slam2_ros2_amd._lib import Context
    gray = synth.mono_image(4, 640, 480)
    ctx = Context(640, 480, n_features=1000, max_images=1)
    k, _ = ctx.extract(gray)
    n = len(k)
    rng = np.random.default_rng(9)
    raw = rng.integers(0, 30000, (480, 640)).astype(np.uint16)
    raw[rng.random((480, 640)) < 0.2] = 0                                           # holes in the depth map
    depth, scale = (raw, 5000.0) if depth_dtype == np.uint16 else ((raw / np.float32(5000)).astype(np.float32), 1.0)
    ku, d, ru = Frame.finishRGBD(ctx, 0, TUM, depth, scale)
    xy = np.stack([k["x"], k["y"]], 1)
    K = np.array([TUM[q] for q in ("fx", "fy", "cx", "cy")], np.float32)
    D = np.array([TUM[q] for q in ("k1", "k2", "p1", "p2", "k3")], np.float32)
    xyu = orc.undistort_points(xy, K, D)
    od, oru = orc.rgbd_lookup(xy, xyu, depth, scale, TUM["bf"])
    assert np.array_equal(np.stack([ku["x"][:n], ku["y"][:n]], 1).view(np.int32), xyu.view(np.int32))        # bit-exact floats
    assert np.array_equal(d[:n], od) and np.array_equal(ru[:n], oru) and (d[n:] == -1).all() and (ru[n:] == -1).all()
    assert (od > 0).sum() > 0.6 * n and (od < 0).sum() > 0.1 * n
    k2, _ = ctx.fetch_features(0)
    assert np.array_equal(k2["x"], ku["x"][:n]) and np.array_equal(k2["octave"], k["octave"])   # undistorted in place, rest untouched
    # the feature grid of the guided search now sees undistorted positions: every keypoint finds itself there
    bi, bd, sd, nc = ctx.search_in_area(0, np.stack([ku["x"][:n], ku["y"][:n]], 1)[:100], np.full(100, 2.0, np.float32),
                                        k["octave"][:100].astype(np.int8), k["octave"][:100].astype(np.int8), ctx.fetch_features(0)[1][:100])
    assert (bd == 0).all() and (nc >= 1).all()
    # k1 == 0: keypoints untouched (Camera.cc:31), depth still looked up
    ctx.extract(gray)
    flat = dict(TUM, k1=0.0)
    ku0, d0, _ = ctx.frame_rgbd(0, flat, depth, scale)
    assert np.array_equal(ku0["x"][:n], k["x"]) and np.array_equal(d0[:n], od)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("color_order,depth_dtype", [(0, np.uint16), (2, np.uint16), (1, np.float32), (0, None)])
def test_device_one_call_rgbd_frame_equals_the_two_calls(orc, color_order, depth_dtype):
    """orbfe_frame_rgbd_image (Frame::createRGBD's device work as one launch sequence) against orbfe_extract_color / orbfe_extract followed by
    orbfe_frame_rgbd -- which the tests above pin to the oracle -- and, for the gray image, against the oracle directly; repeated so that the
    captured sequence is replayed, with changing images and a second slot on its own lane"""
    from orb_slam2_ros2_amd._lib import Context
    ctx, ref = Context(640, 480, n_features=1000, max_images=2), Context(640, 480, n_features=1000, max_images=1)
    rng = np.random.default_rng(11)
    try:
        for rep, f in enumerate((4, 4, 6, 4)):
            gray = synth.mono_image(f, 640, 480)
            img = gray if color_order == 0 else np.stack([gray, np.roll(gray, 3, 1), np.roll(gray, 2, 0)], 2).copy()
            depth, scale = None, 1.0
            if depth_dtype is not None:
                raw = rng.integers(0, 30000, (480, 640)).astype(np.uint16)
                raw[rng.random((480, 640)) < 0.2] = 0
                depth, scale = (raw, 5000.0) if depth_dtype == np.uint16 else ((raw / np.float32(5000)).astype(np.float32), 1.0)
            slot = rep & 1
            ku, d, dd, ru = ctx.frame_rgbd_image(img, TUM, depth, scale, color_order, slot=slot)
            k0, d0 = ref.extract_color(img, color_order) if color_order else ref.extract(img)
            ku0, dd0, ru0 = ref.frame_rgbd(0, TUM, depth, scale)
            n = len(k0)
            assert len(ku) == n and np.array_equal(d, d0)
            assert ku.tobytes() == ku0[:n].tobytes()
            assert np.array_equal(dd.view(np.int64), dd0.view(np.int64)) and np.array_equal(ru.view(np.int64), ru0.view(np.int64))
            if color_order == 0:
                ok, od = orc.extractor(gray, n_features=1000).extract()
                assert n == len(ok) and np.array_equal(d, od)
            # the slot's device-resident results are the frame's: a guided search against it sees the undistorted keypoints
            assert np.array_equal(ctx.pyramid(slot, 0, False), ref.pyramid(0, 0, False))
    finally:
        ctx.close(), ref.close()


@pytest.mark.gpu
def test_device_glue_error_paths():
    from orb_slam2_ros2_amd._lib import Context, OrbfeError
    ctx = Context(640, 480, n_features=500, max_images=1)
    img = np.zeros((480, 640, 3), np.uint8)
    with pytest.raises(OrbfeError):
        ctx.extract_color(img, 3)                                   # Camera.Color is 1 (RGB) or 2 (BGR)
    k, d = ctx.extract_color(img, 1)
    assert len(k) == 0                                              # a flat image has no corners, and that is not an error
    with pytest.raises(OrbfeError):
        ctx.frame_rgbd(5, TUM)                                      # slot out of range
    with pytest.raises(OrbfeError):
        ctx.frame_rgbd(0, TUM, np.zeros((480, 640), np.uint16), depth_scale=0.0)
    ku, dd, ru = ctx.frame_rgbd(0, TUM, np.zeros((480, 640), np.uint16), depth_scale=5000.0)
    assert (dd == -1).all() and (ru == -1).all()                    # no keypoints: every entry is "no depth"
    ctx.close()
