"""HIP keyframe row (triangulation + culling) vs the oracle through the C-ABI: triangulated
points bit-exact (float outputs of an f64 4x4 Jacobi SVD in identical operation order), culling identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,seed,kw", [(600, 21, {}), (1, 22, {}), (257, 23, dict(outlier_frac=0.5)),
                                        (5000, 24, dict(pix_noise=1.0)), (64, 25, dict(baseline=0.01))])
def test_triangulate_points_bit_exact(mvo, O, ctx, n, seed, kw):
    kf = mvo.synth.keyframe_problem(n=n, seed=seed, **kw)
    T = kf["T_curr_to_prev"]
    pp, pc = ctx.triangulate_points(kf["kp_ref"], kf["kp_cur"], kf["K"], T[:3, :3], T[:3, 3])
    po, co = O.triangulate_points(kf["kp_ref"], kf["kp_cur"], kf["K"], T[:3, :3], T[:3, 3])
    assert np.array_equal(pp, po, equal_nan=True) and np.array_equal(pc, co, equal_nan=True)
    good = kf["inlier_gt"]
    if good.sum() > 10 and kw.get("baseline", 1) > 0.1:
        rel = np.linalg.norm(pp[good] - kf["p_ref"][good], axis=1) / kf["p_ref"][good][:, 2]
        assert np.median(rel) < 0.05


def test_triangulate_edge_cases(mvo, O, ctx):
    kf = mvo.synth.keyframe_problem(n=40, seed=26)
    T = kf["T_curr_to_prev"]
    pp, pc = ctx.triangulate_points(kf["kp_ref"][:0], kf["kp_cur"][:0], kf["K"], T[:3, :3], T[:3, 3])
    assert pp.shape == (0, 3) and pc.shape == (0, 3)
    # identical cameras (no baseline): the DLT is rank deficient -- whatever comes out must equal the oracle
    a, b = ctx.triangulate_points(kf["kp_ref"], kf["kp_ref"], kf["K"], np.eye(3), np.zeros(3))
    ao, bo = O.triangulate_points(kf["kp_ref"], kf["kp_ref"], kf["K"], np.eye(3), np.zeros(3))
    assert np.array_equal(a, ao, equal_nan=True) and np.array_equal(b, bo, equal_nan=True)


def test_retain_good_triangulation_matches_oracle(mvo, O, ctx):
    for seed, args in [(27, (1.0, 20.0)), (28, (3.0, 1.5)), (29, (0.0, 1e9))]:
        kf = mvo.synth.keyframe_problem(n=900, seed=seed)
        T = kf["T_curr_to_prev"]
        _, pc = ctx.triangulate_points(kf["kp_ref"], kf["kp_cur"], kf["K"], T[:3, :3], T[:3, 3])
        keep, ang = mvo.retain_good_triangulation(pc, kf["T_w_cur"], kf["T_w_ref"], *args)
        ko, ao = O.retain_good_triangulation(pc, kf["T_w_cur"], kf["T_w_ref"], *args)
        assert np.array_equal(keep, ko) and np.array_equal(ang, ao, equal_nan=True)
        if seed == 27:
            assert 0.5 * 900 < len(keep) < 900
    k0, a0 = mvo.retain_good_triangulation(np.zeros((0, 3), np.float32), kf["T_w_cur"], kf["T_w_ref"])
    assert len(k0) == 0 and len(a0) == 0


@pytest.mark.parametrize("n,seed,kw", [(500, 8, {}), (500, 9, dict(outlier_frac=0.5)), (300, 10, dict(outlier_frac=0.0)),
                                        (2000, 11, dict(outlier_frac=0.3)), (40, 12, dict(outlier_frac=0.7)),
                                        (800, 13, dict(outlier_frac=0.85))])
def test_find_essential_inliers_matches_the_oracle(mvo, O, ctx, n, seed, kw):
    """helperFindInlierMatchesByEpipolarCons: every evaluated hypothesis' candidate counts, the chosen candidate, the
    loop length and the inlier list are bit-exact (the device evaluates 256 or 1000 iterations, the sequential loop
    stops inside that range)."""
    kf = mvo.synth.keyframe_problem(n=n, seed=seed, **kw)
    got = ctx.find_essential_inliers(kf["kp_ref"], kf["kp_cur"], kf["K"], 0.999, 1.0)
    dbg = ctx.debug_essential()
    ref = O.find_essential_inliers(kf["kp_ref"], kf["kp_cur"], kf["K"], 0.999, 1.0)
    run = ref["iters_run"]
    assert dbg["iters_run"] == run and dbg["evaluated"] in (256, 1000) and dbg["evaluated"] >= run
    assert (dbg["evaluated"] == 256) == (run <= 256)
    assert np.array_equal(dbg["counts"][:run], ref["counts"][:run])
    assert (dbg["best_iter"], dbg["best_model"]) == (ref["best_iter"], ref["best_model"])
    assert np.array_equal(got, ref["inliers"])
    if kw.get("outlier_frac", 0.2) <= 0.5 and n >= 300:
        gt = kf["inlier_gt"]
        assert gt[got].mean() > 0.9 and len(got) > 0.9 * gt.sum()


def test_find_essential_inliers_edge_cases(mvo, O, ctx):
    kf = mvo.synth.keyframe_problem(n=60, seed=14, outlier_frac=0.0)
    a, b, K = kf["kp_ref"], kf["kp_cur"], kf["K"]
    assert len(ctx.find_essential_inliers(a[:4], b[:4], K)) == 0
    assert len(ctx.find_essential_inliers(a[:0], b[:0], K)) == 0
    five = ctx.find_essential_inliers(a[:5], b[:5], K)
    assert np.array_equal(five, O.find_essential_inliers(a[:5], b[:5], K)["inliers"])
    rng = np.random.RandomState(3)
    junk = rng.uniform(0, 480, (60, 2)).astype(np.float32)
    assert np.array_equal(ctx.find_essential_inliers(a, junk, K), O.find_essential_inliers(a, junk, K)["inliers"])
    same = ctx.find_essential_inliers(a, a, K)                      # no motion at all: degenerate five-point systems
    assert np.array_equal(same, O.find_essential_inliers(a, a, K)["inliers"])
    for prob, thr in [(0.5, 1.0), (0.999, 0.05), (0.9999999, 3.0)]:
        assert np.array_equal(ctx.find_essential_inliers(a, b, K, prob, thr), O.find_essential_inliers(a, b, K, prob, thr)["inliers"])
    with pytest.raises(mvo.MvoError):
        ctx.find_essential_inliers(a, b, K, prob=1.0)


def test_keyframe_insertion_end_to_end(mvo, O, ctx):
    """vo_addFrame.cpp:104-118 in one piece: epipolar inlier filter -> triangulation -> angle culling."""
    kf = mvo.synth.keyframe_problem(n=900, seed=15, outlier_frac=0.25)
    K, T = kf["K"], kf["T_curr_to_prev"]
    inl = ctx.find_essential_inliers(kf["kp_ref"], kf["kp_cur"], K)
    _, p_cur = ctx.triangulate_points(kf["kp_ref"][inl], kf["kp_cur"][inl], K, T[:3, :3], T[:3, 3])
    keep, ang = mvo.retain_good_triangulation(p_cur, kf["T_w_cur"], kf["T_w_ref"])
    inl_o = O.find_essential_inliers(kf["kp_ref"], kf["kp_cur"], K)["inliers"]
    _, p_cur_o = O.triangulate_points(kf["kp_ref"][inl_o], kf["kp_cur"][inl_o], K, T[:3, :3], T[:3, 3])
    keep_o, ang_o = O.retain_good_triangulation(p_cur_o, kf["T_w_cur"], kf["T_w_ref"])
    assert np.array_equal(inl, inl_o) and np.array_equal(p_cur, p_cur_o, equal_nan=True)
    assert np.array_equal(keep, keep_o) and np.array_equal(ang, ang_o, equal_nan=True)
    good = kf["inlier_gt"][inl][keep]
    assert good.mean() > 0.95 and len(keep) > 0.5 * kf["inlier_gt"].sum()
    err = np.linalg.norm(p_cur[keep][good] - kf["p_cur"][inl][keep][good], axis=1) / kf["p_cur"][inl][keep][good][:, 2]
    assert np.median(err) < 0.03
