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
