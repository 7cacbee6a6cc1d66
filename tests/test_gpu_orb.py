"""HIP ORB path vs the oracle, stage by stage, through the C-ABI.  Integer / byte work: bit-exact."""
import numpy as np
import pytest

from conftest import assert_struct_equal

pytestmark = pytest.mark.gpu

SIZES = [(160, 120, 3), (320, 240, 3), (333, 251, 1), (640, 480, 3)]


DEFAULTS = dict(nfeatures=8000, scale_factor=1.2, nlevels=4, fast_threshold=20, max_keypoints=1500, grid_size=16,
                grid_max_per_cell=8,   # config/config.yaml:65-69,94-95
                pyramid_interpolation=1)   # cv::ORB of OpenCV >= 3.4 (INTER_LINEAR_EXACT)


def _cfg(mvo, O, ctx, **kw):
    """(Re)configures the shared ctx from the reference's defaults + overrides; returns the matching oracle params."""
    full = dict(DEFAULTS)
    full.update(kw)
    ctx.orb_configure(**full)
    return O.default_params(**ctx.params)


def _cand_cmp(mvo, gc, oc):
    assert len(gc) == len(oc), "candidate count %d vs %d" % (len(gc), len(oc))
    g = np.zeros(len(gc), oc.dtype)
    g["x"], g["y"] = gc["x"], gc["y"]
    g["level"] = gc["level_score"] >> 16
    g["fast_score"] = gc["level_score"] & 0xffff
    g["harris"], g["angle"] = gc["harris"], gc["angle"]
    assert_struct_equal(g, oc, "candidates")


@pytest.mark.parametrize("w,h,ch", SIZES)
def test_pyramid_blur_candidates_bit_exact(mvo, O, ctx, w, h, ch):
    p = _cfg(mvo, O, ctx, max_keypoints=2000)
    img = mvo.synth.small_test_image(w + h, w, h, channels=ch)
    ctx.calc_keypoints(img)
    for l in range(p.nlevels):
        for blurred in (False, True):
            g = ctx.debug_level(l, blurred)
            o = O.pyramid_level(img, p, l, blurred)
            assert g.shape == o.shape, (l, g.shape, o.shape)
            bad = np.argwhere(g != o)
            assert len(bad) == 0, "level %d blurred=%d: %d px differ, first %s (gpu %d oracle %d)" % (
                l, blurred, len(bad), bad[0], g[tuple(bad[0])], o[tuple(bad[0])])
    _cand_cmp(mvo, ctx.debug_candidates(), O.candidates(img, p))


@pytest.mark.parametrize("nlevels,sf", [(4, 1.2), (8, 1.2), (3, 2.0)])
def test_both_pyramid_kernels_bit_exact(mvo, O, ctx, nlevels, sf):
    """The LDS-tiled pyramid kernel and the per-pixel chain kernel (the fallback for level groups whose source regions
    do not fit the LDS pool) are two implementations of the same integers: both equal the oracle's levels, on a
    gray and on a BGR image, also for the second level group (levels 4-7 hang off level 3)."""
    for ch, seed in ((3, 5), (1, 6)):
        img = mvo.synth.small_test_image(seed, 333, 251, channels=ch)
        p = _cfg(mvo, O, ctx, nlevels=nlevels, scale_factor=sf, max_keypoints=2000, nfeatures=3000)
        for force in (0, 1):
            mvo.debug_set("pyr_force_chain", force)
            try:
                ctx.calc_keypoints(img, cap=8192)
                for l in range(p.nlevels):
                    assert np.array_equal(ctx.debug_level(l, False), O.pyramid_level(img, p, l, False)), (ch, force, l)
            finally:
                mvo.debug_set("pyr_force_chain", 0)


def test_legacy_inter_linear_pyramid_bit_exact(mvo, O, ctx):
    """pyramid_interpolation = 0: cv::INTER_LINEAR as cv::ORB of OpenCV < 3.4 resampled (11-bit coefficients, truncating
    vertical pass) -- the other flavour of the oracle; every level, the candidates and the descriptors follow."""
    p = _cfg(mvo, O, ctx, max_keypoints=2000, pyramid_interpolation=0)
    img = mvo.synth.small_test_image(99, 320, 240)
    k = ctx.calc_keypoints(img, cap=4096)
    for l in range(1, p.nlevels):
        assert np.array_equal(ctx.debug_level(l, False), O.pyramid_level(img, p, l, False)), l
    pe = O.default_params(max_keypoints=2000)
    assert (O.pyramid_level(img, p, 1) != O.pyramid_level(img, pe, 1)).any()      # really the other arithmetic
    _cand_cmp(mvo, ctx.debug_candidates(), O.candidates(img, p))
    ko = O.calc_keypoints(img, p)
    assert_struct_equal(k, ko.astype(k.dtype), "calcKeyPoints (INTER_LINEAR pyramid)")
    k2, d = ctx.calc_descriptors(img, k, reuse_pyramid=True)
    ko2, do = O.calc_descriptors(img, ko, p)
    assert np.array_equal(d, do)


@pytest.mark.parametrize("w,h,ch", SIZES)
def test_keypoints_and_descriptors_bit_exact(mvo, O, ctx, w, h, ch):
    p = _cfg(mvo, O, ctx, max_keypoints=2000)
    img = mvo.synth.small_test_image(7 * w + h, w, h, channels=ch)
    k = ctx.calc_keypoints(img, cap=4096)
    ko = O.calc_keypoints(img, p)
    assert_struct_equal(k, ko.astype(k.dtype), "calcKeyPoints")
    k2, d, rgb = ctx.calc_descriptors(img, k, reuse_pyramid=True, want_rgb=True)
    ko2, do, rgbo = O.calc_descriptors(img, ko, p, want_rgb=True)
    assert_struct_equal(k2, ko2.astype(k.dtype), "calcDescriptors keypoints")
    assert np.array_equal(d, do), "descriptors: %d rows differ" % (d != do).any(1).sum()
    assert np.array_equal(rgb, rgbo)
    # the non-reuse path rebuilds the pyramid and must agree
    k3, d3 = ctx.calc_descriptors(img, k, reuse_pyramid=False)
    assert np.array_equal(d3, do)


def test_quota_limited_selection_and_small_thresholds(mvo, O, ctx):
    """nfeatures small enough that retainBest (nth_element + ties) actually cuts, FAST threshold low."""
    img = mvo.synth.small_test_image(11, 480, 360)
    for nf, thr in ((300, 20), (1000, 7), (50, 40)):
        p = _cfg(mvo, O, ctx, nfeatures=nf, fast_threshold=thr, max_keypoints=5000)
        k = ctx.calc_keypoints(img, cap=8192)
        ko = O.calc_keypoints(img, p)
        assert_struct_equal(k, ko.astype(k.dtype), "nfeatures=%d thr=%d" % (nf, thr))
        _, d = ctx.calc_descriptors(img, k, reuse_pyramid=True)
        _, do = O.calc_descriptors(img, ko, p)
        assert np.array_equal(d, do)


def test_other_pyramid_shapes(mvo, O, ctx):
    img = mvo.synth.small_test_image(5, 400, 300)
    for nlevels, sf in ((1, 1.2), (2, 1.5), (6, 1.2), (8, 1.1)):
        p = _cfg(mvo, O, ctx, nlevels=nlevels, scale_factor=sf, max_keypoints=3000, nfeatures=4000)
        k = ctx.calc_keypoints(img, cap=8192)
        ko = O.calc_keypoints(img, p)
        assert_struct_equal(k, ko.astype(k.dtype), "nlevels=%d" % nlevels)
        _, d = ctx.calc_descriptors(img, k, reuse_pyramid=True)
        _, do = O.calc_descriptors(img, ko, p)
        assert np.array_equal(d, do)


def test_grid_latching_and_degenerate_images(mvo, O, ctx):
    _cfg(mvo, O, ctx, max_keypoints=1500)
    flat = np.full((240, 320, 3), 90, np.uint8)
    k = ctx.calc_keypoints(flat)
    assert len(k) == 0
    k, d = ctx.calc_descriptors(flat, k, reuse_pyramid=True)
    assert len(k) == 0
    # S640-shaped frame 0 of the bench sequence: full-size parity
    seq = mvo.synth.Sequence(640, 480, 4, seed=1234, tex_size=1024)
    p = _cfg(mvo, O, ctx, max_keypoints=2000)
    for i in range(2):
        img = seq.frame(i)
        k = ctx.calc_keypoints(img, cap=4096)
        ko = O.calc_keypoints(img, p)
        assert_struct_equal(k, ko.astype(k.dtype), "S640 frame %d" % i)
        k, d = ctx.calc_descriptors(img, k, reuse_pyramid=True)
        ko, do = O.calc_descriptors(img, ko, p)
        assert np.array_equal(d, do)
        assert 1500 < len(k) <= 2001


def test_config4_kitti_shape_4000_keypoints(mvo, O, ctx):
    """BASELINE configs[3]: 1242x375, 4000 keypoints."""
    seq = mvo.synth.Sequence(1242, 375, 2, seed=99, tex_size=2048)
    p = _cfg(mvo, O, ctx, max_keypoints=4000)
    img = seq.frame(1)
    k = ctx.calc_keypoints(img, cap=8192)
    ko = O.calc_keypoints(img, p)
    assert_struct_equal(k, ko.astype(k.dtype), "S1242")
    k, d = ctx.calc_descriptors(img, k, reuse_pyramid=True)
    ko, do = O.calc_descriptors(img, ko, p)
    assert np.array_equal(d, do) and len(k) > 3000
    idx, dist = ctx.match_knn2(d, d[::-1].copy())
    io, do_ = O.match_knn2(d, d[::-1].copy())
    assert np.array_equal(idx, io) and np.array_equal(dist, do_)


def test_device_resident_image_and_descriptors(mvo, O, ctx):
    import torch
    p = _cfg(mvo, O, ctx, max_keypoints=1000)
    img = mvo.synth.small_test_image(21, 320, 240)
    t = torch.from_numpy(img).cuda()
    torch.cuda.synchronize()
    k = ctx.calc_keypoints_dev(t.data_ptr(), 320, 240, 320 * 3, 3, cap=2048)
    ko = O.calc_keypoints(img, p)
    assert_struct_equal(k, ko.astype(k.dtype), "calc_keypoints_dev")
    k2, d, dptr = ctx.calc_descriptors_dev(k)
    _, do = O.calc_descriptors(img, ko, p)
    assert np.array_equal(d, do) and dptr
    # device descriptors feed the matcher without a host round trip
    idx, dist = ctx.match_knn2_dev(dptr, len(k2), dptr, len(k2))
    assert (idx[:, 0] == np.arange(len(k2))).all() or (dist[:, 0] == 0).all()


def test_error_paths(mvo, ctx):
    with pytest.raises(mvo.MvoError):
        ctx.orb_configure(nlevels=99)
    ctx.orb_configure(nlevels=4)
    img = mvo.synth.small_test_image(1, 160, 120)
    k = ctx.calc_keypoints(img)
    with pytest.raises(mvo.MvoError) as e:
        ctx.calc_keypoints(img, cap=1)
    assert e.value.code == mvo.MVO_ERR_CAPACITY
    ctx.orb_configure(nlevels=4)     # drops the cached pyramid
    with pytest.raises(mvo.MvoError) as e:
        ctx.calc_descriptors(img, k, reuse_pyramid=True)
    assert e.value.code == mvo.MVO_ERR_STATE


def test_row_stride_bgra_and_large_frames(mvo, O, ctx):
    """Padded row strides (cv::Mat::step), 4-channel BGRA input, and a 1920x1080 frame."""
    import ctypes as C
    p = _cfg(mvo, O, ctx, max_keypoints=3000)
    img = mvo.synth.small_test_image(77, 318, 203)
    ko = O.calc_keypoints(img, p)
    ko, do = O.calc_descriptors(img, ko, p)
    # padded stride: 318 * 3 = 954 -> 1024 bytes per row
    pad = np.zeros((203, 1024), np.uint8)
    pad[:, :954] = img.reshape(203, 954)
    out = np.zeros(4096, mvo.KEYPOINT_DTYPE)
    n = C.c_int()
    r = ctx.lib.mvo_calc_keypoints(ctx.h, pad.ctypes.data_as(C.c_void_p), 318, 203, 1024, 3, out.ctypes.data_as(C.c_void_p),
                                   4096, C.byref(n))
    assert r == 0
    assert_struct_equal(out[:n.value], ko.astype(out.dtype), "padded stride")
    desc = np.zeros((n.value, 32), np.uint8)
    r = ctx.lib.mvo_calc_descriptors(ctx.h, pad.ctypes.data_as(C.c_void_p), 318, 203, 1024, 3, 0,
                                     out.ctypes.data_as(C.c_void_p), C.byref(n), desc.ctypes.data_as(C.c_void_p), None)
    assert r == 0 and np.array_equal(desc[:n.value], do)
    # BGRA: the alpha channel is ignored
    bgra = np.concatenate([img, np.full((203, 318, 1), 200, np.uint8)], axis=2)
    k4 = ctx.calc_keypoints(np.ascontiguousarray(bgra), cap=4096)
    assert_struct_equal(k4, ko.astype(k4.dtype), "BGRA")
    # full-HD frame
    seq = mvo.synth.Sequence(1920, 1080, 1, seed=5, tex_size=2048)
    big = seq.frame(0)
    p = _cfg(mvo, O, ctx, max_keypoints=8000, nfeatures=20000)
    k = ctx.calc_keypoints(big, cap=16384)
    kb = O.calc_keypoints(big, p)
    assert_struct_equal(k, kb.astype(k.dtype), "1920x1080")
    k, d = ctx.calc_descriptors(big, k, reuse_pyramid=True)
    kb, db = O.calc_descriptors(big, kb, p)
    assert np.array_equal(d, db) and len(k) > 4000


@pytest.mark.parametrize("w,h", [(640, 480), (1242, 375), (333, 251)])
def test_both_candidate_orderings_bit_exact(mvo, O, w, h):
    """The canonical (level, row, column) order of the candidates is restored either by the kernel (the last workgroup of
    a tile row, default) or by the host thread (a ctx in throughput mode leaves the GPU that time): both must deliver the
    oracle's list, element for element."""
    img = mvo.synth.small_test_image(5 * w + h, w, h)
    for mode in ("latency", "throughput"):
        c = mvo.Context(0, max_keypoints=3000)
        c.ba_set_mode(mode)
        p = O.default_params(**c.params)
        for rep in range(3):                         # (the arrival counters re-arm themselves)
            k = c.calc_keypoints(img, cap=8192)
            assert_struct_equal(k, O.calc_keypoints(img, p).astype(k.dtype), "calcKeyPoints, %s mode, run %d" % (mode, rep))
            _cand_cmp(mvo, c.debug_candidates(), O.candidates(img, p))
            # descriptors: keypoint windows blurred inside k_brief (latency mode) or whole levels blurred behind the detection
            # kernel and sampled by k_brief_sample (throughput mode) -- the oracle's bits either way, with and without reuse
            ko = O.calc_keypoints(img, p)
            k2, d = c.calc_descriptors(img, k, reuse_pyramid=True)
            ko2, do = O.calc_descriptors(img, ko, p)
            assert_struct_equal(k2, ko2.astype(k.dtype), "calcDescriptors keypoints, %s mode" % mode)
            assert np.array_equal(d, do), "descriptors, %s mode: %d rows differ" % (mode, (d != do).any(1).sum())
            k3, d3 = c.calc_descriptors(img, k, reuse_pyramid=False)
            assert np.array_equal(d3, do), "descriptors without reuse, %s mode" % mode
            assert np.array_equal(c.debug_level(1, True), O.pyramid_level(img, p, 1, True))
        c.close()
