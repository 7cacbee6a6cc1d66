"""HIP tracking rows vs the oracle through the C-ABI.

map points in view: indices, pixels and gathered descriptors bit-exact.
solvePnPRansac: every hypothesis (R, t), every inlier count, the chosen iteration and the inlier list bit-exact; the
refined (rvec, tvec) within 1e-8 (the refinement's sums are lane-partitioned on the device, sequential in the oracle;
tolerance stated by north_star for poses: 1e-4 relative)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same_models(a, b):
    return np.array_equal(a, b, equal_nan=True)


def _to_device(a):
    """A host array placed in HBM (what mvo_calc_descriptors_dev leaves there for the current frame)."""
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("seed,n_map", [(3, 4000), (4, 1), (5, 1023), (6, 1025), (7, 20000)])
def test_map_points_in_view_bit_exact(mvo, O, ctx, seed, n_map):
    pr = mvo.synth.tracking_problem(n_map=n_map, seed=seed)
    m = ctx.map_create()
    try:
        ctx.map_upload(m, pr["map_pos"], pr["map_desc"])
        idx, px, d_desc = ctx.map_points_in_view(m, pr["T_w_c"], pr["K"], pr["cols"], pr["rows"], cap=n_map)
        io, po = O.map_in_view(pr["map_pos"], pr["T_w_c"], pr["K"], pr["cols"], pr["rows"])
        assert np.array_equal(idx, io) and np.array_equal(px, po)
        if len(idx):
            # the gathered descriptors feed the matcher without leaving HBM (vo.cpp:281-289, method 1)
            cur = mvo.synth.match_inputs("perturbed", 300, 300, seed=seed)[1]
            t_cur = _to_device(cur)
            i2, d2 = ctx.match_knn2_dev(d_desc, len(idx), t_cur.data_ptr(), len(cur))
            i2o, d2o = O.match_knn2(pr["map_desc"][io], cur)
            assert np.array_equal(i2, i2o) and np.array_equal(d2, d2o)
        # bundle adjustment moves points: positions only
        moved = pr["map_pos"] + np.float32(0.25)
        ctx.map_update_positions(m, moved[n_map // 2:], first=n_map // 2)
        idx2, px2, _ = ctx.map_points_in_view(m, pr["T_w_c"], pr["K"], pr["cols"], pr["rows"], cap=n_map)
        half = pr["map_pos"].copy()
        half[n_map // 2:] = moved[n_map // 2:]
        io2, po2 = O.map_in_view(half, pr["T_w_c"], pr["K"], pr["cols"], pr["rows"])
        assert np.array_equal(idx2, io2) and np.array_equal(px2, po2)
    finally:
        ctx.map_release(m)


def test_map_points_in_view_errors(mvo, ctx):
    pr = mvo.synth.tracking_problem(n_map=500, seed=1)
    m = ctx.map_create()
    try:
        idx, px, d = ctx.map_points_in_view(m, pr["T_w_c"], pr["K"], 640, 480, cap=10)      # empty map
        assert len(idx) == 0
        ctx.map_upload(m, pr["map_pos"], pr["map_desc"])
        with pytest.raises(mvo.MvoError) as e:
            ctx.map_points_in_view(m, pr["T_w_c"], pr["K"], 640, 480, cap=3)
        assert e.value.code == mvo.MVO_ERR_CAPACITY
        with pytest.raises(mvo.MvoError) as e:
            ctx.map_points_in_view(m, np.zeros((4, 4)), pr["K"], 640, 480, cap=500)
        assert e.value.code == mvo.MVO_ERR_INVALID
        with pytest.raises(mvo.MvoError):
            ctx.map_update_positions(m, pr["map_pos"], first=10)                           # runs past the end
    finally:
        ctx.map_release(m)


@pytest.mark.parametrize("seed,kw", [(11, {}), (12, dict(outlier_frac=0.5)), (13, dict(outlier_frac=0.0)),
                                     (15, dict(planar=True)), (16, dict(n_map=400, pix_noise=0.0)),
                                     (18, dict(n_map=12000)), (19, dict(outlier_frac=0.8))])
def test_solve_pnp_ransac_matches_the_oracle(mvo, O, ctx, seed, kw):
    pr = mvo.synth.tracking_problem(seed=seed, **kw)
    p3, p2, K = pr["pts3d"], pr["pts2d"], pr["K"]
    got = ctx.solve_pnp_ransac(p3, p2, K)
    dbg = ctx.debug_pnp()
    ref = O.solve_pnp_ransac(p3, p2, K)
    assert got["ok"] == ref["ok"]
    # the device evaluates all 100 hypotheses; the oracle stops where the sequential loop stops
    run = ref["iters_run"]
    assert dbg["n_hyp"] == 100 and dbg["iters_run"] == run and dbg["best_iter"] == ref["best_iter"]   # 100: no correction
    assert np.array_equal(dbg["counts"][:run], ref["counts"][:run])
    assert _same_models(dbg["models"][:run], ref["models"][:run])
    assert np.array_equal(got["inliers"], ref["inliers"])
    if ref["ok"]:
        assert dbg["dlt"] == ref["dlt"] and dbg["lm_iters"] == ref["lm_iters"]
        assert np.abs(got["rvec"] - ref["rvec"]).max() < 1e-8 and np.abs(got["tvec"] - ref["tvec"]).max() < 1e-8
        Tcw = np.linalg.inv(pr["T_w_c"])
        assert np.abs(mvo.rodrigues(got["rvec"]) - Tcw[:3, :3]).max() < 2e-3
        assert np.abs(got["tvec"] - Tcw[:3, 3]).max() < 5e-3


@pytest.mark.parametrize("seed,kw", [(11, {}), (12, dict(outlier_frac=0.5)), (13, dict(outlier_frac=0.0)), (19, dict(outlier_frac=0.8)),
                                     (16, dict(n_map=400, pix_noise=0.0))])
def test_solve_pnp_ransac_in_chunks_matches_the_oracle(mvo, O, seed, kw):
    """Contexts that share the GPU (THROUGHPUT / SHARED mode) whose last RANSAC loop was short evaluate the first 32 hypotheses,
    replay the loop's bookkeeping and launch the other 68 only if the sequential loop would have gone on (vo.cpp:326-329: same
    model, same inliers, same loop length as cv::solvePnPRansac's loop either way -- here: the oracle's)."""
    pr = mvo.synth.tracking_problem(seed=seed, **kw)
    p3, p2, K = pr["pts3d"], pr["pts2d"], pr["K"]
    ref = O.solve_pnp_ransac(p3, p2, K)
    c = mvo.Context(0)
    try:
        c.ba_set_mode("shared")
        for rep in range(3):
            got = c.solve_pnp_ransac(p3, p2, K)
            dbg = c.debug_pnp()
            run = ref["iters_run"]
            assert got["ok"] == ref["ok"] and dbg["iters_run"] == run and dbg["best_iter"] == ref["best_iter"]
            # the first call of a ctx evaluates everything; after a short loop (<= 28 iterations) the next call starts with 32
            assert dbg["n_hyp"] == (32 if (rep > 0 and run <= 28) else 100), (rep, dbg["n_hyp"], run)   # (positive: the device's choice stood)
            assert np.array_equal(dbg["counts"][:run], ref["counts"][:run])
            assert _same_models(dbg["models"][:run], ref["models"][:run])
            assert np.array_equal(got["inliers"], ref["inliers"])
            if ref["ok"]:
                assert dbg["dlt"] == ref["dlt"] and dbg["lm_iters"] == ref["lm_iters"]
                assert np.abs(got["rvec"] - ref["rvec"]).max() < 1e-8 and np.abs(got["tvec"] - ref["tvec"]).max() < 1e-8
        if ref["iters_run"] <= 28:
            # ... and a frame whose loop then needs every iteration: the first 32 are not enough, the other 68 follow (two chunks)
            hard = mvo.synth.tracking_problem(seed=19, outlier_frac=0.8)
            refh = O.solve_pnp_ransac(hard["pts3d"], hard["pts2d"], hard["K"])
            assert refh["iters_run"] > 32
            goth = c.solve_pnp_ransac(hard["pts3d"], hard["pts2d"], hard["K"])
            dbg = c.debug_pnp()
            assert dbg["n_hyp"] == 100 and dbg["iters_run"] == refh["iters_run"] and dbg["best_iter"] == refh["best_iter"]
            assert np.array_equal(dbg["counts"][:refh["iters_run"]], refh["counts"][:refh["iters_run"]])
            assert np.array_equal(goth["inliers"], refh["inliers"])
            assert np.abs(goth["rvec"] - refh["rvec"]).max() < 1e-8 and np.abs(goth["tvec"] - refh["tvec"]).max() < 1e-8
    finally:
        c.close()


def test_all_hypotheses_bit_exact(mvo, O, ctx):
    """Beyond the sequential stopping point: every one of the 100 hypotheses equals the oracle's EPnP + score."""
    pr = mvo.synth.tracking_problem(seed=21, outlier_frac=0.35)
    p3, p2, K = pr["pts3d"], pr["pts2d"], pr["K"]
    ctx.solve_pnp_ransac(p3, p2, K)
    dbg = ctx.debug_pnp()
    sub = O.pnp_subsets(len(p3), 100)
    for h in range(100):
        R, t = O.epnp(p3, p2, sub[h], K)
        assert _same_models(np.concatenate([R.ravel(), t]), dbg["models"][h]), "hypothesis %d" % h
        assert O.pnp_score(p3, p2, K, R, t)[0] == dbg["counts"][h]


def test_solve_pnp_ransac_edge_cases(mvo, O, ctx):
    pr = mvo.synth.tracking_problem(seed=14, outlier_frac=0.0, pix_noise=0.0)
    p3, p2, K = pr["pts3d"], pr["pts2d"], pr["K"]
    few = ctx.solve_pnp_ransac(p3[:4], p2[:4], K)
    assert not few["ok"] and len(few["inliers"]) == 0
    none = ctx.solve_pnp_ransac(p3[:0], p2[:0], K)
    assert not none["ok"]
    five, five_o = ctx.solve_pnp_ransac(p3[:5], p2[:5], K), O.solve_pnp_ransac(p3[:5], p2[:5], K)
    assert five["ok"] and five["inliers"].tolist() == [0, 1, 2, 3, 4]
    assert np.abs(five["rvec"] - five_o["rvec"]).max() < 1e-12 and np.array_equal(five["tvec"], five_o["tvec"])
    rng = np.random.RandomState(0)
    junk2d = rng.uniform(0, 480, (60, 2)).astype(np.float32)
    junk, junk_o = ctx.solve_pnp_ransac(p3[:60], junk2d, K), O.solve_pnp_ransac(p3[:60], junk2d, K)
    assert junk["ok"] == junk_o["ok"] and np.array_equal(junk["inliers"], junk_o["inliers"])
    one, one_o = ctx.solve_pnp_ransac(p3, p2, K, iterations=1), O.solve_pnp_ransac(p3, p2, K, iters=1)
    assert one["ok"] == one_o["ok"] and np.array_equal(one["inliers"], one_o["inliers"])
    tight, tight_o = (ctx.solve_pnp_ransac(p3, p2, K, reprojection_error=0.05, confidence=0.5),
                      O.solve_pnp_ransac(p3, p2, K, reproj=0.05, confidence=0.5))
    assert np.array_equal(tight["inliers"], tight_o["inliers"])
    with pytest.raises(mvo.MvoError) as e:
        ctx.solve_pnp_ransac(p3, p2, K, confidence=1.0)
    assert e.value.code == mvo.MVO_ERR_INVALID
    nan3 = p3[:50].copy()
    nan3[7] = np.nan                                                       # a NaN pair is never an inlier
    bad, bad_o = ctx.solve_pnp_ransac(nan3, p2[:50], K), O.solve_pnp_ransac(nan3, p2[:50], K)
    assert bad["ok"] == bad_o["ok"] and np.array_equal(bad["inliers"], bad_o["inliers"]) and 7 not in bad["inliers"]


def test_host_corrects_a_wrong_device_choice(mvo, O, ctx):
    """The device replays the RANSAC loop itself (one host round trip per solve) and the host re-checks the choice on
    the returned counts.  With the test hook the device replays with confidence 0.5, stops the loop early and picks
    an earlier hypothesis than the real loop would: the result must still be the oracle's."""
    pr = mvo.synth.tracking_problem(seed=12, outlier_frac=0.5)
    p3, p2, K = pr["pts3d"], pr["pts2d"], pr["K"]
    ref = O.solve_pnp_ransac(p3, p2, K)
    corrected = 0
    try:
        mvo.debug_set("pnp_replay_skew", 1)
        for conf in (0.999, 0.9999999):
            ref = O.solve_pnp_ransac(p3, p2, K, confidence=conf)
            got = ctx.solve_pnp_ransac(p3, p2, K, confidence=conf)
            dbg = ctx.debug_pnp()
            corrected += dbg["n_hyp"] < 0
            assert dbg["best_iter"] == ref["best_iter"] and np.array_equal(got["inliers"], ref["inliers"])
            assert np.abs(got["rvec"] - ref["rvec"]).max() < 1e-8 and np.abs(got["tvec"] - ref["tvec"]).max() < 1e-8
    finally:
        mvo.debug_set("pnp_replay_skew", 0)
    assert corrected >= 1, "the hook never made the device choose differently: the test does not test the correction"
    got = ctx.solve_pnp_ransac(p3, p2, K)
    assert ctx.debug_pnp()["n_hyp"] == 100                      # normal operation: no correction needed


def test_rodrigues_host(mvo, O):
    rng = np.random.RandomState(2)
    for r in [rng.normal(size=3) * s for s in (1e-9, 0.2, 2.0)] + [np.zeros(3)]:
        assert np.abs(mvo.rodrigues(r) - O.rodrigues(r)).max() < 1e-15


def test_tracking_step_end_to_end(mvo, O, ctx):
    """vo.cpp:270-357 in one piece: map in view -> match against the frame's descriptors -> 3D-2D pairs -> PnP."""
    pr = mvo.synth.tracking_problem(n_map=3000, seed=31, outlier_frac=0.0)
    rng = np.random.RandomState(5)
    m = ctx.map_create()
    try:
        ctx.map_upload(m, pr["map_pos"], pr["map_desc"])
        idx, px, d_desc = ctx.map_points_in_view(m, pr["T_w_c"], pr["K"], pr["cols"], pr["rows"], cap=3000)
        # the current frame sees the visible map points (descriptor with a few flipped bits) plus clutter
        seen = rng.permutation(len(idx))[: int(0.8 * len(idx))]
        bits = np.unpackbits(pr["map_desc"][idx[seen]], axis=1)
        bits ^= (rng.uniform(size=bits.shape) < 0.03).astype(np.uint8)
        cur_desc = np.concatenate([np.packbits(bits, axis=1), rng.randint(0, 256, (400, 32)).astype(np.uint8)])
        cur_px = np.concatenate([px[seen] + rng.normal(0, 0.3, (len(seen), 2)).astype(np.float32),
                                 rng.uniform(0, 480, (400, 2)).astype(np.float32)])
        t_cur = _to_device(cur_desc)
        matches = ctx.match_features_dev(d_desc, len(idx), t_cur.data_ptr(), len(cur_desc), method=1)
        ref_matches = O.match_features(pr["map_desc"][idx], cur_desc, method=1)
        assert matches.tobytes() == ref_matches.tobytes()
        p3 = pr["map_pos"][idx[matches["queryIdx"]]]
        p2 = cur_px[matches["trainIdx"]]
        got, ref = ctx.solve_pnp_ransac(p3, p2, pr["K"]), O.solve_pnp_ransac(p3, p2, pr["K"])
        assert got["ok"] and np.array_equal(got["inliers"], ref["inliers"]) and len(got["inliers"]) > 0.9 * len(seen)
        T_c_w = np.eye(4)
        T_c_w[:3, :3] = mvo.rodrigues(got["rvec"])
        T_c_w[:3, 3] = got["tvec"]
        assert np.abs(np.linalg.inv(T_c_w) - pr["T_w_c"]).max() < 2e-3        # curr_->T_w_c_ = convertRt2T(R, t).inv()
    finally:
        ctx.map_release(m)
