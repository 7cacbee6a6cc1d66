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


def _skew(t):
    return np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])


def test_real_roots_against_numpy(O):
    rng = np.random.RandomState(0)
    for trial in range(200):
        if trial % 3 == 0:      # prescribed real roots, some clustered, plus complex pairs
            k = rng.randint(0, 6) * 2
            rr = np.sort(rng.uniform(-5, 5, k))
            c = np.poly(np.concatenate([rr, (rng.normal(size=(10 - k) // 2) + 1j * rng.uniform(0.5, 2, (10 - k) // 2)),
                                        ])) if False else None
        c = rng.normal(size=11) * 10.0 ** rng.uniform(-3, 3, 11)
        got = O.real_roots_deg10(c)
        ref = np.roots(c)
        ref = np.sort(ref[np.abs(ref.imag) < 1e-9 * np.maximum(1, np.abs(ref))].real)
        assert len(got) == len(ref), (trial, got, ref)
        assert np.all(np.diff(got) > 0)
        assert np.abs(got - ref).max(initial=0) <= 1e-7 * np.maximum(1, np.abs(ref)).max(initial=1)
        res = np.abs(np.polyval(c, got))
        scale = np.polyval(np.abs(c), np.abs(got))
        assert (res <= 1e-12 * scale).all()                       # bisection ends on adjacent doubles
    # known roots
    c = np.poly([-3, -1, 0.5, 0.5000001, 2, 4, 7, 1 + 1j, 1 - 1j, -8])
    got = O.real_roots_deg10(c)
    assert len(got) == 8 and np.abs(got - np.array([-8, -3, -1, 0.5, 0.5000001, 2, 4, 7])).max() < 1e-6
    assert len(O.real_roots_deg10(np.r_[np.zeros(10), 1.0])) == 0    # constant
    assert np.allclose(O.real_roots_deg10(np.r_[np.zeros(9), 2.0, -3.0]), [1.5])   # leading zeros: 2 z - 3


def test_five_point_recovers_the_essential_matrix(O, S):
    kf = S.keyframe_problem(n=300, seed=7, pix_noise=0.0, outlier_frac=0.0)
    K = kf["K"]
    # exact normalised coordinates from the scene (double): x2^T E x1 = 0 with x1 in ref, x2 in cur
    x1 = kf["p_ref"][:, :2] / kf["p_ref"][:, 2:]
    x2 = kf["p_cur"][:, :2] / kf["p_cur"][:, 2:]
    T = kf["T_curr_to_prev"]                                   # p_cur = R p_ref + t
    Et = _skew(T[:3, 3]) @ T[:3, :3]
    Et /= np.linalg.norm(Et)
    rng = np.random.RandomState(1)
    hits = 0
    for _ in range(25):
        idx = rng.choice(300, 5, replace=False)
        E, dbg = O.five_point(x1[idx], x2[idx], want_dbg=True)
        assert 1 <= len(E) <= 10
        # the null space really is one, the reduced system starts with the identity
        Q = np.stack([np.r_[b[0] * a[0], b[0] * a[1], b[0], b[1] * a[0], b[1] * a[1], b[1], a[0], a[1], 1.0]
                      for a, b in zip(x1[idx], x2[idx])])
        assert np.abs(Q @ dbg["basis"].T).max() < 1e-12 and np.abs(dbg["basis"] @ dbg["basis"].T - np.eye(4)).max() < 1e-12
        assert np.array_equal(dbg["A"][:, :10], np.eye(10))
        for e in E:
            assert abs(np.linalg.norm(e) - 1) < 1e-12
            assert np.abs(np.einsum("ni,ij,nj->n", np.c_[x2[idx], np.ones(5)], e, np.c_[x1[idx], np.ones(5)])).max() < 1e-9
            # a valid essential matrix: two equal singular values and a zero one
            sv = np.linalg.svd(e, compute_uv=False)
            assert abs(sv[0] - sv[1]) < 1e-6 and sv[2] < 1e-6
        d = min(min(np.abs(e - Et).max(), np.abs(e + Et).max()) for e in E)
        hits += d < 1e-6
    assert hits >= 23          # a few 5-point samples are ill-conditioned


def test_essential_ransac_finds_the_inliers(O, S):
    for seed, out in [(8, 0.2), (9, 0.5), (10, 0.0)]:
        kf = S.keyframe_problem(n=500, seed=seed, outlier_frac=out)
        res = O.find_essential_inliers(kf["kp_ref"], kf["kp_cur"], kf["K"], 0.999, 1.0)
        gt = set(np.nonzero(kf["inlier_gt"])[0].tolist())
        got = set(res["inliers"].tolist())
        # a 1-px Sampson gate on 0.3-px noise; wrong matches that happen to lie on the epipolar line pass
        assert len(gt - got) <= 0.05 * len(gt) and len(got - gt) <= 0.1 * len(got) + 3
        assert np.all(np.diff(res["inliers"]) > 0) and 1 <= res["iters_run"] <= 1000
        c = res["counts"][:res["iters_run"]]
        assert c[res["best_iter"], res["best_model"]] == c.max() == len(res["inliers"])
        assert (res["counts"][res["iters_run"]:] == -2).all()
    assert len(O.find_essential_inliers(kf["kp_ref"][:4], kf["kp_cur"][:4], kf["K"])["inliers"]) == 0
    five = O.find_essential_inliers(kf["kp_ref"][:5], kf["kp_cur"][:5], kf["K"])
    assert five["inliers"].tolist() == [0, 1, 2, 3, 4]
