"""Known-answer tests for the ORACLE's tracking rows: getMappointsInCurrentView_
(vo.cpp:16-49) and cv::solvePnPRansac as called at vo.cpp:326-329.  Parity unpinned: checked against analytic
answers / numpy / scipy, not against OpenCV."""
import numpy as np
import pytest
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation


@pytest.fixture(scope="module")
def S():
    from conftest import graft
    return graft.load_package().synth


def _project(K, R, t, X):
    q = X @ R.T + t
    return np.stack([K["fx"] * q[:, 0] / q[:, 2] + K["cx"], K["fy"] * q[:, 1] / q[:, 2] + K["cy"]], 1)


def test_rng_subsets_are_the_mwc_sequence(O):
    # cv::RNG: state = (u32)state * 4164903690 + (state >> 32), seeded with (uint64)-1; uniform(0, n) = next % n
    st = 0xFFFFFFFFFFFFFFFF
    want = []
    for _ in range(4):
        row = []
        while len(row) < 5:
            st = ((st & 0xFFFFFFFF) * 4164903690 + (st >> 32)) & 0xFFFFFFFFFFFFFFFF
            v = (st & 0xFFFFFFFF) % 1000
            row.append(v)
            if v in row[:-1]:
                row.pop()      # getSubset redraws the SAME slot until it differs from the earlier ones
        want.append(row)
    got = O.pnp_subsets(1000, 4)
    assert got.tolist() == want
    small = O.pnp_subsets(6, 200)                      # heavy duplicate rejection
    assert all(len(set(r)) == 5 for r in small.tolist()) and small.min() >= 0 and small.max() < 6


def test_invert4x4_and_map_in_view(O, S):
    pr = S.tracking_problem(n_map=4000, seed=3)
    T = pr["T_w_c"]
    assert np.abs(O.invert4x4(T) - np.linalg.inv(T)).max() < 1e-14
    assert O.invert4x4(np.zeros((4, 4))) is None
    idx, px = O.map_in_view(pr["map_pos"], T, pr["K"], pr["cols"], pr["rows"])
    # numpy restatement with the same float/double casts (opencv_funcs.cpp:67-78, camera.cpp:23-28)
    Ti = O.invert4x4(T)
    p = pr["map_pos"].astype(np.float64)
    res = np.zeros((len(p), 3))
    for j in range(3):
        res += 0  # keep the accumulation order explicit below
    res = ((Ti[:3, 0] * p[:, :1] + Ti[:3, 1] * p[:, 1:2]) + Ti[:3, 2] * p[:, 2:3]) + Ti[:3, 3] * 1.0
    pc = res.astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        u = (pr["K"]["fx"] * pc[:, 0].astype(np.float64) / pc[:, 2].astype(np.float64) + pr["K"]["cx"]).astype(np.float32)
        v = (pr["K"]["fy"] * pc[:, 1].astype(np.float64) / pc[:, 2].astype(np.float64) + pr["K"]["cy"]).astype(np.float32)
    keep = (pc[:, 2] >= 0) & (u > 0) & (v > 0) & (u < pr["cols"]) & (v < pr["rows"])
    assert np.array_equal(idx, np.nonzero(keep)[0])
    assert np.array_equal(px, np.stack([u[keep], v[keep]], 1))
    assert 0.2 * len(p) < len(idx) < 0.8 * len(p)       # the scene does exercise both outcomes
    assert set(pr["ids"].tolist()) == set(idx.tolist())


def test_rodrigues_matches_scipy_and_finite_differences(O):
    rng = np.random.RandomState(0)
    for r in [rng.normal(size=3) * s for s in (1e-3, 0.3, 1.0, 3.0)] + [np.zeros(3)]:
        R, J = O.rodrigues(r, want_jac=True)
        assert np.abs(R - Rotation.from_rotvec(r).as_matrix()).max() < 1e-14
        if 0 < np.linalg.norm(r) < np.pi:
            assert np.abs(O.rodrigues_inv(R) - r).max() < 1e-10
        assert np.abs(O.rodrigues(O.rodrigues_inv(R)) - R).max() < 1e-12      # |r| > pi maps to the short way round
        for i in range(3):
            d = np.zeros(3)
            d[i] = 1e-6
            fd = (O.rodrigues(r + d) - O.rodrigues(r - d)).reshape(9) / 2e-6
            assert np.abs(J[i] - fd).max() < 1e-8
    # the 180-degree branch of the inverse
    R = Rotation.from_rotvec([0, np.pi, 0]).as_matrix()
    assert abs(np.linalg.norm(O.rodrigues_inv(R)) - np.pi) < 1e-7


def test_epnp_is_exact_on_noise_free_points(O, S):
    pr = S.tracking_problem(seed=5, pix_noise=0, outlier_frac=0)
    Tcw = np.linalg.inv(pr["T_w_c"])
    rng = np.random.RandomState(1)
    ok = 0
    for _ in range(20):
        idx = rng.choice(len(pr["pts3d"]), 5, replace=False).astype(np.int32)
        R, t, ut = O.epnp(pr["pts3d"], pr["pts2d"], idx, pr["K"], want_ut=True)
        assert np.abs(ut @ ut.T - np.eye(12)).max() < 1e-12          # accumulated rotations stay orthonormal
        assert abs(np.linalg.det(R) - 1) < 1e-9
        # pts2d is float32 (cv::Point2f): ~3e-5 px of rounding, amplified by the 5-point geometry
        ok += np.abs(R - Tcw[:3, :3]).max() < 1e-3 and np.abs(t - Tcw[:3, 3]).max() < 3e-3
    assert ok >= 17                                                    # a few subsets are near-degenerate
    idx = np.arange(40, dtype=np.int32)
    R, t = O.epnp(pr["pts3d"], pr["pts2d"], idx, pr["K"])              # n > 5 goes through the same code
    assert np.abs(R - Tcw[:3, :3]).max() < 1e-5 and np.abs(t - Tcw[:3, 3]).max() < 1e-5


def test_score_counts_reprojection_errors(O, S):
    pr = S.tracking_problem(seed=6)
    Tcw = np.linalg.inv(pr["T_w_c"])
    cnt, mask = O.pnp_score(pr["pts3d"], pr["pts2d"], pr["K"], Tcw[:3, :3], Tcw[:3, 3], 2.0)
    uv = _project(pr["K"], Tcw[:3, :3], Tcw[:3, 3], pr["pts3d"].astype(np.float64))
    e2 = ((uv - pr["pts2d"]) ** 2).sum(1)
    sure = np.abs(e2 - 4.0) > 1e-3
    assert np.array_equal(mask[sure] != 0, (e2 <= 4.0)[sure]) and cnt == mask.sum()
    assert np.array_equal(mask != 0, pr["inlier_gt"]) or (mask != 0)[pr["inlier_gt"]].mean() > 0.99
    # NaN / behind-camera models yield zero inliers instead of crashing
    assert O.pnp_score(pr["pts3d"], pr["pts2d"], pr["K"], np.full((3, 3), np.nan), np.zeros(3))[0] == 0


def test_iterative_refinement_reaches_the_least_squares_optimum(O, S):
    pr = S.tracking_problem(seed=7, outlier_frac=0, pix_noise=0.5)
    M = pr["pts3d"].astype(np.float64)[:400]
    m = pr["pts2d"].astype(np.float64)[:400]
    out = O.solve_pnp_iterative(M, m, pr["K"])
    assert out["dlt"] == 1 and 1 <= out["lm_iters"] <= 20

    def resid(p):
        return (_project(pr["K"], Rotation.from_rotvec(p[:3]).as_matrix(), p[3:], M) - m).ravel()

    p0 = np.concatenate([out["rvec"], out["tvec"]])
    ref = least_squares(resid, p0, xtol=1e-14, ftol=1e-14, gtol=1e-14)
    # CvLevMarq stops on a FLT_EPSILON relative parameter change: cost at the optimum to ~1e-9 relative
    assert np.sum(resid(p0) ** 2) <= np.sum(ref.fun ** 2) * (1 + 1e-8)
    assert np.abs(p0 - ref.x).max() < 1e-5
    # noise-free: exact pose
    pr0 = S.tracking_problem(seed=8, outlier_frac=0, pix_noise=0)
    out0 = O.solve_pnp_iterative(pr0["pts3d"].astype(np.float64), pr0["pts2d"].astype(np.float64), pr0["K"])
    Tcw = np.linalg.inv(pr0["T_w_c"])
    assert np.abs(O.rodrigues(out0["rvec"]) - Tcw[:3, :3]).max() < 1e-6       # float32 pixels
    assert np.abs(out0["tvec"] - Tcw[:3, 3]).max() < 1e-6


@pytest.mark.parametrize("seed,outliers", [(11, 0.25), (12, 0.5), (13, 0.0)])
def test_ransac_finds_the_inlier_set_and_the_pose(O, S, seed, outliers):
    pr = S.tracking_problem(seed=seed, outlier_frac=outliers)
    res = O.solve_pnp_ransac(pr["pts3d"], pr["pts2d"], pr["K"])
    assert res["ok"] and res["dlt"] == 1
    gt = set(np.nonzero(pr["inlier_gt"])[0].tolist())
    got = set(res["inliers"].tolist())
    assert len(got - gt) <= 2 and len(gt - got) <= 0.02 * len(gt)     # a 2-px gate on 0.4-px noise
    assert np.all(np.diff(res["inliers"]) > 0)                           # ascending indices like the mask scan
    Tcw = np.linalg.inv(pr["T_w_c"])
    assert np.abs(O.rodrigues(res["rvec"]) - Tcw[:3, :3]).max() < 1e-3
    assert np.abs(res["tvec"] - Tcw[:3, 3]).max() < 2e-3
    # the sequential bookkeeping: the best iteration holds the maximum count seen before the loop stopped
    run = res["iters_run"]
    assert res["counts"][res["best_iter"]] == res["counts"][:run].max() == len(res["inliers"])
    assert run <= 100 and (res["counts"][run:] == -1).all()


def test_ransac_edge_cases(O, S):
    pr = S.tracking_problem(seed=14, outlier_frac=0.0, pix_noise=0.0)
    p3, p2 = pr["pts3d"], pr["pts2d"]
    assert not O.solve_pnp_ransac(p3[:4], p2[:4], pr["K"])["ok"]          # fewer than the 5 model points
    five = O.solve_pnp_ransac(p3[:5], p2[:5], pr["K"])                    # npoints == model_points: EPnP only
    assert five["ok"] and five["inliers"].tolist() == [0, 1, 2, 3, 4] and five["iters_run"] == 1
    rng = np.random.RandomState(0)
    junk = O.solve_pnp_ransac(p3[:60], rng.uniform(0, 480, (60, 2)).astype(np.float32), pr["K"])
    assert (not junk["ok"] and len(junk["inliers"]) == 0) or len(junk["inliers"]) < 12
    planar = S.tracking_problem(seed=15, planar=True)
    res = O.solve_pnp_ransac(planar["pts3d"], planar["pts2d"], planar["K"])
    assert res["ok"] and res["dlt"] == 0                                   # planar: refinement starts from the RANSAC model
    Tcw = np.linalg.inv(planar["T_w_c"])
    assert np.abs(res["tvec"] - Tcw[:3, 3]).max() < 5e-3
