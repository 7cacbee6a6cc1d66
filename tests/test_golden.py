"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle): the oracle
must still reproduce them on CPU, and the HIP path must reproduce them on the GPU."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_struct_equal


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


def test_oracle_reproduces_orb_golden(O):
    g = _load("orb_176x144.npz")
    p = O.default_params(nlevels=3, max_keypoints=300)
    assert_struct_equal(O.candidates(g["image"], p), g["candidates"], "candidates")
    k = O.calc_keypoints(g["image"], p)
    k, d, rgb = O.calc_descriptors(g["image"], k, p, want_rgb=True)
    assert_struct_equal(k, g["keypoints"], "keypoints")
    assert np.array_equal(d, g["descriptors"]) and np.array_equal(rgb, g["rgb"])
    assert np.array_equal(O.pyramid_level(g["image"], p, 2, True), g["level2_blurred"])


def test_oracle_reproduces_match_and_ba_golden(O):
    g = _load("match_150x170.npz")
    idx, dist = O.match_knn2(g["q"], g["t"])
    assert np.array_equal(idx, g["idx"]) and np.array_equal(dist, g["dist"])
    for m in (1, 2):
        assert_struct_equal(O.match_features(g["q2"], g["t2"], m, 2.0, 1.0), g["m%d" % m], "matchFeatures %d" % m)
    b = _load("ba_3x40.npz")
    f, cx, cy = b["intr"]
    args = (b["poses0"], b["points0"], b["edge_pose"], b["edge_point"], b["edge_uv"], f, cx, cy)
    P1, _, st1 = O.bundle_adjustment(*args, fix_points=True)
    assert np.abs(P1 - b["pose_only_poses"]).max() < 1e-12
    P2, X2, st2 = O.bundle_adjustment(*args, fix_points=False, max_iterations=3)
    assert np.abs(P2 - b["full3_poses"]).max() < 1e-12 and np.abs(X2 - b["full3_points"]).max() < 1e-12


def _track_K(t):
    return dict(fx=t["K4"][0], fy=t["K4"][1], cx=t["K4"][2], cy=t["K4"][3])


def test_oracle_reproduces_tracking_golden(O):
    t = _load("track_300.npz")
    K = _track_K(t)
    idx, px = O.map_in_view(t["map_pos"], t["T_w_c"], K, int(t["size"][0]), int(t["size"][1]))
    assert np.array_equal(idx, t["view_idx"]) and np.array_equal(px, t["view_px"])
    assert np.array_equal(O.pnp_subsets(len(t["pts3d"]), 40), t["subsets"])
    res = O.solve_pnp_ransac(t["pts3d"], t["pts2d"], K, iters=40)
    run = int(t["iters_run"])
    assert res["best_iter"] == int(t["best_iter"]) and res["iters_run"] == run
    assert np.array_equal(res["counts"], t["counts"]) and np.array_equal(res["inliers"], t["inliers"])
    assert np.array_equal(res["models"][:run], t["models"][:run], equal_nan=True)
    assert np.abs(res["rvec"] - t["rvec"]).max() < 1e-13 and np.abs(res["tvec"] - t["tvec"]).max() < 1e-13


def test_oracle_reproduces_keyframe_golden(O):
    g = _load("keyframe_120.npz")
    K = _track_K(g)
    er = O.find_essential_inliers(g["kp_ref"], g["kp_cur"], K)
    assert np.array_equal(er["inliers"], g["inliers"]) and np.array_equal(er["counts"][:er["iters_run"]], g["counts"])
    assert [er["best_iter"], er["best_model"]] == g["best"].tolist()
    T = g["T_curr_to_prev"]
    pp, pc = O.triangulate_points(g["kp_ref"][g["inliers"]], g["kp_cur"][g["inliers"]], K, T[:3, :3], T[:3, 3])
    assert np.array_equal(pp, g["pts_prev"]) and np.array_equal(pc, g["pts_curr"])
    keep, ang = O.retain_good_triangulation(pc, g["T_w_cur"], g["T_w_ref"])
    assert np.array_equal(keep, g["keep"]) and np.abs(ang - g["angles"]).max() < 1e-12


@pytest.mark.gpu
def test_hip_reproduces_keyframe_golden(mvo, ctx):
    g = _load("keyframe_120.npz")
    K = _track_K(g)
    inl = ctx.find_essential_inliers(g["kp_ref"], g["kp_cur"], K)
    dbg = ctx.debug_essential()
    assert np.array_equal(inl, g["inliers"]) and np.array_equal(dbg["counts"][:dbg["iters_run"]], g["counts"])
    assert [dbg["best_iter"], dbg["best_model"]] == g["best"].tolist()
    T = g["T_curr_to_prev"]
    pp, pc = ctx.triangulate_points(g["kp_ref"][inl], g["kp_cur"][inl], K, T[:3, :3], T[:3, 3])
    assert np.array_equal(pp, g["pts_prev"]) and np.array_equal(pc, g["pts_curr"])
    keep, ang = mvo.retain_good_triangulation(pc, g["T_w_cur"], g["T_w_ref"])
    assert np.array_equal(keep, g["keep"]) and np.abs(ang - g["angles"]).max() < 1e-12


@pytest.mark.gpu
def test_hip_reproduces_tracking_golden(mvo, ctx):
    t = _load("track_300.npz")
    K = _track_K(t)
    m = ctx.map_create()
    try:
        ctx.map_upload(m, t["map_pos"], t["map_desc"])
        idx, px, _ = ctx.map_points_in_view(m, t["T_w_c"], K, int(t["size"][0]), int(t["size"][1]), cap=len(t["map_pos"]))
    finally:
        ctx.map_release(m)
    assert np.array_equal(idx, t["view_idx"]) and np.array_equal(px, t["view_px"])
    res = ctx.solve_pnp_ransac(t["pts3d"], t["pts2d"], K, iterations=40)
    dbg = ctx.debug_pnp()
    run = int(t["iters_run"])
    assert res["ok"] and dbg["best_iter"] == int(t["best_iter"]) and dbg["iters_run"] == run
    assert np.array_equal(dbg["counts"][:run], t["counts"][:run]) and np.array_equal(res["inliers"], t["inliers"])
    assert np.array_equal(dbg["models"][:run], t["models"][:run], equal_nan=True)
    assert np.abs(res["rvec"] - t["rvec"]).max() < 1e-8 and np.abs(res["tvec"] - t["tvec"]).max() < 1e-8


@pytest.mark.gpu
def test_hip_reproduces_golden(mvo, ctx):
    g = _load("orb_176x144.npz")
    ctx.orb_configure(nfeatures=8000, scale_factor=1.2, nlevels=3, fast_threshold=20, max_keypoints=300, grid_size=16,
                      grid_max_per_cell=8)
    k = ctx.calc_keypoints(g["image"])
    c = ctx.debug_candidates()
    assert np.array_equal(c["x"], g["candidates"]["x"]) and np.array_equal(c["harris"], g["candidates"]["harris"])
    assert np.array_equal(ctx.debug_level(2, True), g["level2_blurred"])
    k, d, rgb = ctx.calc_descriptors(g["image"], k, reuse_pyramid=True, want_rgb=True)
    assert_struct_equal(k, g["keypoints"].astype(k.dtype), "keypoints")
    assert np.array_equal(d, g["descriptors"]) and np.array_equal(rgb, g["rgb"])
    m = _load("match_150x170.npz")
    idx, dist = ctx.match_knn2(m["q"], m["t"])
    assert np.array_equal(idx, m["idx"]) and np.array_equal(dist, m["dist"])
    for meth in (1, 2):
        assert_struct_equal(ctx.match_features(m["q2"], m["t2"], meth, 2.0, 1.0), m["m%d" % meth].astype(mvo.DMATCH_DTYPE),
                            "matchFeatures %d" % meth)
    b = _load("ba_3x40.npz")
    f, cx, cy = b["intr"]
    args = (b["poses0"], b["points0"], b["edge_pose"], b["edge_point"], b["edge_uv"], f, cx, cy)
    P1, _, st1 = ctx.bundle_adjustment(*args, fix_points=True)
    assert np.abs(P1 - b["pose_only_poses"]).max() < 1e-4 * np.abs(b["pose_only_poses"]).max()
    assert abs(st1["chi2_final"] - b["pose_only_chi2"][1]) < 1e-6 * b["pose_only_chi2"][1]
    P2, X2, st2 = ctx.bundle_adjustment(*args, fix_points=False, max_iterations=3)
    assert np.abs(P2 - b["full3_poses"]).max() < 1e-8 and np.abs(X2 - b["full3_points"]).max() < 1e-8
