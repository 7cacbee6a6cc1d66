"""Known-answer tests that pin the ORACLE's ORB restatement to the published algorithm.
Parity is unpinned against OpenCV itself (not installed, no upstream golden vectors): these cases are analytic."""
import numpy as np
import pytest


def P(O, **kw):
    return O.default_params(**kw)


def test_level_sizes_and_quotas_match_survey(O):
    p = P(O)
    assert [O.level_size(640, 480, p, l)[:2] for l in range(4)] == [(640, 480), (533, 400), (444, 333), (370, 278)]
    assert [O.level_size(1242, 375, p, l)[:2] for l in range(4)] == [(1242, 375), (1035, 312), (862, 260), (719, 217)]
    assert O.feature_quota(p) == [2575, 2146, 1788, 1491]                      # nfeatures 8000
    assert O.feature_quota(P(O, nfeatures=2000)) == [644, 537, 447, 372]
    assert O.feature_quota(P(O, nfeatures=4000)) == [1288, 1073, 894, 745]
    assert abs(O.level_size(640, 480, p, 1)[2] - 1.2) < 1e-6


def test_gray_fixed_point_and_border_reflect101(O):
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (70, 80, 3)).astype(np.uint8)
    lv = O.pyramid_level(img, P(O, nlevels=1), 0)
    b, g, r = [img[:, :, i].astype(np.int64) for i in range(3)]
    gray = ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)
    assert np.array_equal(lv[32:-32, 32:-32], gray)
    assert np.array_equal(lv, np.pad(gray, 32, mode="reflect"))               # BORDER_REFLECT_101
    # pure colours: weights sum to 2^14
    one = np.full((70, 80, 3), 255, np.uint8)
    assert (O.pyramid_level(one, P(O, nlevels=1), 0) == 255).all()


def test_resize_constant_and_ramp(O):
    p = P(O)
    img = np.full((120, 160), 77, np.uint8)
    for l in range(4):
        assert (O.pyramid_level(img, p, l) == 77).all()
    ramp = np.tile(np.arange(160, dtype=np.uint8), (120, 1))
    l1 = O.pyramid_level(ramp, p, 1)[32:-32, 32:-32]
    assert l1.shape == (100, 133)
    # bilinear on a linear ramp: value ~ (dx+0.5)*1.2-0.5 (edges clamp)
    x = np.arange(133)
    expect = (x + 0.5) * (160 / 133) - 0.5
    assert np.abs(l1[50].astype(float) - expect)[1:-1].max() <= 0.76  # two truncating >>16 + one rounding >>2
    assert np.abs(l1.astype(int) - l1[0].astype(int)).max() <= 1       # truncation depends on the row weights


def test_blur_kernel_properties(O):
    p = P(O, nlevels=1)
    flat = np.full((80, 90), 131, np.uint8)
    assert (O.pyramid_level(flat, p, 0, blurred=True) == 131).all()           # kernel sums to 256 exactly
    imp = np.zeros((81, 91), np.uint8)
    imp[40, 45] = 255
    bl = O.pyramid_level(imp, p, 0, blurred=True)[32:-32, 32:-32].astype(int)
    k = np.array([18, 34, 48, 56, 48, 34, 18])
    expect = (np.outer(k, k) * 255 + 32768) >> 16
    assert np.array_equal(bl[37:44, 42:49], expect)
    assert bl.sum() == expect.sum()
    # frame stays unblurred (cv::ORB::compute blurs the level ROI in place)
    rng = np.random.RandomState(1)
    img = rng.randint(0, 256, (80, 90)).astype(np.uint8)
    raw, blur = O.pyramid_level(img, p, 0), O.pyramid_level(img, p, 0, blurred=True)
    m = np.ones_like(raw, bool)
    m[32:-32, 32:-32] = False
    assert np.array_equal(raw[m], blur[m]) and not np.array_equal(raw[~m], blur[~m])


def _fast_img(arc_len, start=0, center=100, ring=150, base=100, size=101):
    circle = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2),
              (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    img = np.full((size, size), base, np.uint8)
    c = size // 2
    img[c, c] = center
    for k in range(arc_len):
        dx, dy = circle[(start + k) % 16]
        img[c + dy, c + dx] = ring
    return img, c


@pytest.mark.parametrize("start", [0, 5, 11, 15])
def test_fast_9_arc_is_corner_8_arc_is_not(O, start):
    p = P(O, nlevels=1, fast_threshold=20)
    img, c = _fast_img(9, start)
    cand = O.candidates(img, p)
    hit = cand[(cand["x"] == c) & (cand["y"] == c)]
    assert len(hit) == 1 and hit[0]["fast_score"] == 49                     # |150-100| - 1
    img8, _ = _fast_img(8, start)
    cand8 = O.candidates(img8, p)
    assert not ((cand8["x"] == c) & (cand8["y"] == c)).any()


def test_fast_threshold_is_strict(O):
    img, c = _fast_img(9, 3, center=100, ring=120)                           # difference == threshold
    assert len(O.candidates(img, P(O, nlevels=1, fast_threshold=20))) == 0
    cand = O.candidates(img, P(O, nlevels=1, fast_threshold=19))
    assert len(cand) == 1 and cand[0]["fast_score"] == 19
    dark, c = _fast_img(9, 3, center=100, ring=40)
    cand = O.candidates(dark, P(O, nlevels=1, fast_threshold=20))
    assert len(cand) == 1 and cand[0]["fast_score"] == 59


def test_fast_border_31_and_row_major_order(O):
    rng = np.random.RandomState(5)
    from conftest import graft
    img = graft.load_package().synth.small_test_image(3, 200, 150, channels=1)
    cand = O.candidates(img, P(O, nlevels=2))
    assert len(cand) > 20
    for l in range(2):
        w, h, _ = O.level_size(200, 150, P(O, nlevels=2), l)
        c = cand[cand["level"] == l]
        assert c["x"].min() >= 31 and c["x"].max() < w - 31 and c["y"].min() >= 31 and c["y"].max() < h - 31
        key = c["y"].astype(int) * 4096 + c["x"]
        assert (np.diff(key) > 0).all()
    assert (np.diff(cand["level"]) >= 0).all()
    assert rng is not None


def test_nms_plateau_keeps_none_and_strict_max(O):
    # two adjacent identical corners: neither is a strict maximum -> both suppressed
    img, c = _fast_img(9, 0)
    p = P(O, nlevels=1)
    a = O.candidates(img, p)
    assert len(a) == 1
    img2 = np.full((101, 140), 100, np.uint8)
    img2[:, :101] = img
    img2[:, 39:140] = np.maximum(img2[:, 39:140], 0)
    # build a real plateau by duplicating the pattern one pixel to the right with equal scores
    plate = np.full((101, 101), 100, np.uint8)
    plate[44:58, 30:70] = 100
    cand = O.candidates(plate, p)
    assert len(cand) == 0


def test_harris_and_angle_on_analytic_patterns(O):
    p = P(O, nlevels=1)
    # vertical step edge through a FAST corner: angle of the intensity centroid
    img = np.full((101, 101), 60, np.uint8)
    img[:, 51:] = 200                      # bright half plane to the right (+x)
    img[50, 50] = 255
    img2, c = _fast_img(9, 0, center=255, ring=10, base=60)
    k = O.orb_detect(np.where(img2 != 60, img2, img), p)
    # centroid lies towards +x -> angle near 0/360
    if len(k):
        a = k[0]["angle"]
        assert a < 20 or a > 340
    # symmetric patch -> m10 = m01 = 0 -> fastAtan2(0,0) = 0
    sym, c = _fast_img(16, 0, center=100, ring=160)
    cand = O.candidates(sym, p)
    hit = cand[(cand["x"] == c) & (cand["y"] == c)]
    assert len(hit) == 1 and hit[0]["angle"] == 0.0
    # Harris of a flat region is 0
    flat, c = _fast_img(9, 0)
    h = O.candidates(flat, p)[0]["harris"]
    assert np.isfinite(h)


def test_fast_atan2_polynomial_accuracy(O):
    # IC angle of a linear ramp I = ax + by + c is atan2(b, a) up to the 0.3 deg polynomial error
    ys, xs = np.mgrid[0:101, 0:101]
    for ang in (10, 45, 100, 180, 225, 300, 359):
        a, b = np.cos(np.deg2rad(ang)), np.sin(np.deg2rad(ang))
        ramp = 128 + 2.0 * (a * (xs - 50) + b * (ys - 50))
        img = np.clip(np.rint(ramp), 0, 255).astype(np.uint8)
        img2, c = _fast_img(9, 0, center=255, ring=0, base=0)
        mask = np.zeros_like(img2, bool)
        mask[c, c] = True
        for k in range(9):
            pass
        patch = np.where(img2 != 0, img2, img)
        cand = O.candidates(patch, P(O, nlevels=1, fast_threshold=10))
        hit = cand[(cand["x"] == c) & (cand["y"] == c)]
        if len(hit):
            d = abs((hit[0]["angle"] - ang + 180) % 360 - 180)
            assert d < 8.0, (ang, hit[0]["angle"])


def test_grid_sampling_rules(O):
    p = P(O, max_keypoints=1500, grid_size=16, grid_max_per_cell=8)
    kp = np.zeros(12, O.KEYPOINT_DTYPE)
    kp["x"] = 5.0 + np.arange(12) * 0.5        # all in cell (0,0)
    kp["y"] = 7.0
    out = O.select_uniform_kpts_by_grid(kp, 30, 40, p)
    assert len(out) == 8 and np.array_equal(out["x"], kp["x"][:8])   # 9th point in a cell is rejected
    # max+1 cutoff: `cnt > max` (feature_match.cpp:77) lets max+1 keypoints through
    p2 = P(O, max_keypoints=20, grid_size=16, grid_max_per_cell=8)
    kp = np.zeros(300, O.KEYPOINT_DTYPE)
    kp["x"] = (np.arange(300) % 40) * 16 + 1
    kp["y"] = (np.arange(300) // 40) * 16 + 1
    out = O.select_uniform_kpts_by_grid(kp, 30, 40, p2)
    assert len(out) == 21 and out.tobytes() == kp[:21].tobytes()
    # (int) truncation of the float coordinates
    kp = np.zeros(2, O.KEYPOINT_DTYPE)
    kp["x"] = [15.99, 16.0]
    kp["y"] = [0.2, 0.2]
    assert len(O.select_uniform_kpts_by_grid(kp, 30, 40, P(O, grid_max_per_cell=1))) == 2


def test_detect_compute_pipeline_properties(O):
    from conftest import graft
    S = graft.load_package().synth
    img = S.small_test_image(7, 320, 240)
    p = P(O, max_keypoints=400)
    k = O.calc_keypoints(img, p)
    assert 50 < len(k) <= 401
    assert (np.diff(k["octave"]) >= 0).all()                         # level-major order survives the grid
    sc = np.float32(1.2) ** k["octave"]
    assert np.allclose(k["size"], 31 * sc, rtol=1e-6)
    assert ((k["angle"] >= 0) & (k["angle"] < 360.0001)).all()
    k2, d = O.calc_descriptors(img, k, p)
    assert len(k2) == len(k) and d.shape == (len(k), 32)
    bits = np.unpackbits(d, axis=1).mean()
    assert 0.3 < bits < 0.7
    # a keypoint within 31 px of the border is dropped by compute (feature_match.h:15-17)
    kb = k[:3].copy()
    kb["x"][0] = 10.0
    k3, d3 = O.calc_descriptors(img, kb, p)
    assert len(k3) == 2 and np.array_equal(d3, d[1:3])
    # rgb colour = BGR pixel at (floor x, floor y) reversed (frame.h:80-85)
    _, _, rgb = O.calc_descriptors(img, k, p, want_rgb=True)
    i = 5
    px = img[int(np.floor(k["y"][i])), int(np.floor(k["x"][i]))]
    assert tuple(rgb[i]) == (px[2], px[1], px[0])


def test_brief_rotation_consistency(O):
    """Descriptor of a pattern rotated by 90 deg with the angle rotated accordingly is identical (taps land on
    exact pixels for multiples of 90 deg)."""
    rng = np.random.RandomState(3)
    base = rng.randint(0, 256, (141, 141)).astype(np.uint8)
    p = P(O, nlevels=1)
    kp = np.zeros(1, O.KEYPOINT_DTYPE)
    kp["x"], kp["y"], kp["size"], kp["octave"], kp["class_id"] = 70, 70, 31, 0, -1
    descs = []
    for q in range(4):
        img = np.ascontiguousarray(np.rot90(base, -q))   # rotate image clockwise by 90q (image coords: +angle)
        kp["angle"] = (90.0 * q) % 360
        # blur commutes with 90-degree rotations (symmetric kernel)
        _, d = O.calc_descriptors(img, kp, p)
        descs.append(d[0])
    for q in range(1, 4):
        assert np.array_equal(descs[0], descs[q]), q


def test_blur_kernel_is_the_error_diffused_fixed_point_gaussian(O):
    """The 7-tap sigma-2 kernel is DERIVED like OpenCV's bit-exact GaussianBlur does (normalised Gaussian -> 8.8 fixed point
    with error diffusion from the tails, the centre takes the remainder): {18, 34, 48, 56, 48, 34, 18}; read back here as the
    impulse response of the blur (horizontal x vertical = outer product, (+2^15) >> 16)."""
    k = np.exp(-0.5 / 4.0 * (np.arange(7) - 3.0) ** 2)
    k = k / k.sum() * 256
    err, half = 0.0, []
    for i in range(3):
        adj = k[i] + err
        v = int(np.rint(adj))
        err = adj - v
        half.append(v)
    kern = np.array(half + [256 - 2 * sum(half)] + half[::-1])
    assert kern.tolist() == [18, 34, 48, 56, 48, 34, 18]
    imp = np.zeros((41, 41), np.uint8)
    imp[20, 20] = 255
    bl = O.pyramid_level(imp, P(O, nlevels=1), 0, blurred=True)[32:-32, 32:-32].astype(int)
    want = (255 * np.outer(kern, kern) + 32768) >> 16
    assert np.array_equal(bl[17:24, 17:24], want)


def test_pyramid_interpolation_flavours(O):
    """cv::ORB resamples its pyramid with INTER_LINEAR_EXACT from OpenCV 3.4 on (canonical here, the reference needs
    >= 3.4.5) and with INTER_LINEAR before.  Both are restated; they differ by at most one grey level, agree on constants and
    on exact 2:1 decimation grids, and EXACT equals its defining formula."""
    rng = np.random.RandomState(4)
    img = rng.randint(0, 256, (97, 131)).astype(np.uint8)
    pe, pl = P(O, nlevels=2), P(O, nlevels=2, pyramid_interpolation=0)
    le = O.pyramid_level(img, pe, 1)[32:-32, 32:-32].astype(int)
    ll = O.pyramid_level(img, pl, 1)[32:-32, 32:-32].astype(int)
    assert le.shape == ll.shape and np.abs(le - ll).max() <= 1 and (le != ll).any()
    # the defining formula of the exact flavour: coordinates in double, 8-bit coefficients, one rounding
    h, w = img.shape
    dh, dw = le.shape
    def tab(s, d):
        f = (s / d) * (np.arange(d) + 0.5) - 0.5
        o = np.floor(f).astype(int)
        f = f - o
        f[o < 0], o[o < 0] = 0, 0
        f[o >= s - 1], o[o >= s - 1] = 0, s - 1
        c1 = np.rint(f * 256).astype(int)
        return o, 256 - c1, c1
    ox, ax0, ax1 = tab(w, dw)
    oy, ay0, ay1 = tab(h, dh)
    I = img.astype(int)
    x1, y1 = np.minimum(ox + 1, w - 1), np.minimum(oy + 1, h - 1)
    hr = I[:, ox] * ax0 + I[:, x1] * ax1
    want = (hr[oy] * ay0[:, None] + hr[y1] * ay1[:, None] + 32768) >> 16
    assert np.array_equal(le, want)
    for p in (pe, pl):
        assert (O.pyramid_level(np.full((60, 80), 91, np.uint8), p, 1) == 91).all()
