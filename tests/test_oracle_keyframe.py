"""Known-answer tests for the ORACLE's keyframe row: helperTriangulatePoints
(motion_estimation.cpp:214-247, cv::triangulatePoints) and retainGoodTriangulationResult_ (vo.cpp:181-244).
Parity unpinned: checked against numpy / analytic answers, not against OpenCV."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def S():
    from conftest import graft
    return graft.load_package().synth


def test_triangulation_recovers_the_scene_points(O, S):
    kf = S.keyframe_problem(n=500, seed=3, pix_noise=0.0, outlier_frac=0.0)
    T = kf["T_curr_to_prev"]
    p_prev, p_cur = O.triangulate_points(kf["kp_ref"], kf["kp_cur"], kf["K"], T[:3, :3], T[:3, 3])
    # float32 pixels / normalised coordinates: ~1e-5 relative on the rays, amplified by depth / baseline
    assert np.abs(p_prev - kf["p_ref"]).max() < 5e-3 and np.median(np.abs(p_prev - kf["p_ref"])) < 2e-4
    assert np.abs(p_cur - kf["p_cur"]).max() < 5e-3
    # independent DLT with numpy's SVD on the same float32 normalised coordinates
    K = kf["K"]
    for i in (0, 7, 123, 499):
        n1 = np.array([np.float32((kf["kp_ref"][i, 0] - K["cx"]) / K["fx"]), np.float32((kf["kp_ref"][i, 1] - K["cy"]) / K["fy"])], np.float64)
        n2 = np.array([np.float32((kf["kp_cur"][i, 0] - K["cx"]) / K["fx"]), np.float32((kf["kp_cur"][i, 1] - K["cy"]) / K["fy"])], np.float64)
        P1 = np.hstack([np.eye(3), np.zeros((3, 1))])
        P2 = T[:3]
        A = np.stack([n1[0] * P1[2] - P1[0], n1[1] * P1[2] - P1[1], n2[0] * P2[2] - P2[0], n2[1] * P2[2] - P2[1]])
        X = np.linalg.svd(A)[2][3]
        assert np.abs(p_prev[i] - X[:3] / X[3]).max() < 1e-5 * max(1, np.abs(X[:3] / X[3]).max())


def test_triangulation_with_noise_and_wrong_matches_stays_finite_where_it_should(O, S):
    kf = S.keyframe_problem(n=800, seed=4)
    T = kf["T_curr_to_prev"]
    p_prev, p_cur = O.triangulate_points(kf["kp_ref"], kf["kp_cur"], kf["K"], T[:3, :3], T[:3, 3])
    good = kf["inlier_gt"]
    rel = np.linalg.norm(p_prev[good] - kf["p_ref"][good], axis=1) / kf["p_ref"][good][:, 2]
    assert np.median(rel) < 0.02 and np.isfinite(p_prev[good]).all()
    assert O.triangulate_points(kf["kp_ref"][:0], kf["kp_cur"][:0], kf["K"], T[:3, :3], T[:3, 3])[0].shape == (0, 3)


def test_retain_good_triangulation_rules(O, S):
    kf = S.keyframe_problem(n=400, seed=5, outlier_frac=0.0)
    T = kf["T_curr_to_prev"]
    _, p_cur = O.triangulate_points(kf["kp_ref"], kf["kp_cur"], kf["K"], T[:3, :3], T[:3, 3])
    keep, ang = O.retain_good_triangulation(p_cur, kf["T_w_cur"], kf["T_w_ref"], 1.0, 20.0)
    # numpy restatement of vo.cpp:203-211 with the same float cast of the world point
    pw = (p_cur.astype(np.float64) @ kf["T_w_cur"][:3, :3].T + kf["T_w_cur"][:3, 3]).astype(np.float32).astype(np.float64)
    v1, v2 = kf["T_w_cur"][:3, 3] - pw, kf["T_w_ref"][:3, 3] - pw
    want = np.degrees(np.arccos((v1 * v2).sum(1) / (np.linalg.norm(v1, axis=1) * np.linalg.norm(v2, axis=1)))) * (np.pi / 3.1415926)
    assert np.abs(ang - want).max() < 1e-9
    med = np.sort(ang)[len(ang) // 2]
    assert np.array_equal(keep, np.nonzero(~((ang < 1.0) | (ang / med > 20.0)))[0])
    assert 0 < len(keep) <= len(ang) and 1.0 < med < 10.0
    # thresholds bite: a huge minimum angle removes everything, a tiny ratio removes the upper half
    assert len(O.retain_good_triangulation(p_cur, kf["T_w_cur"], kf["T_w_ref"], 90.0, 20.0)[0]) == 0
    k2, _ = O.retain_good_triangulation(p_cur, kf["T_w_cur"], kf["T_w_ref"], 0.0, 1.0)
    assert np.array_equal(k2, np.nonzero(ang / med <= 1.0)[0])
    assert len(O.retain_good_triangulation(p_cur[:0], kf["T_w_cur"], kf["T_w_ref"])[0]) == 0
