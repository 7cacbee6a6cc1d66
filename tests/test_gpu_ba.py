"""HIP bundle adjustment vs the oracle (f64): poses / landmarks within 1e-4 relative (north-star tolerance)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _args(pb, poses=None):
    return (pb["poses0"] if poses is None else poses, pb["points0"], pb["edge_pose"], pb["edge_point"],
            pb["edge_uv"], pb["focal"], pb["cx"], pb["cy"])


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _check(mvo, O, ctx, pb, tol_x=TOL, **kw):
    P, X, st = ctx.bundle_adjustment(*_args(pb), **kw)
    Po, Xo, sto = O.bundle_adjustment(*_args(pb), **kw)
    msg = "gpu %s\noracle %s" % (st, sto)
    assert np.isfinite(P).all(), msg
    assert _rel(P[:, :3, 3], Po[:, :3, 3]) < TOL and np.abs(P[:, :3, :3] - Po[:, :3, :3]).max() < TOL, msg
    assert _rel(X, Xo) < tol_x, msg
    assert abs(st["chi2_initial"] - sto["chi2_initial"]) <= 1e-9 * sto["chi2_initial"], msg
    assert abs(st["chi2_final"] - sto["chi2_final"]) <= 1e-5 * max(sto["chi2_final"], 1e-12), msg
    return st, sto


@pytest.mark.parametrize("mfma", [1, 0])
@pytest.mark.parametrize("F,L,seed", [(1, 100, 1), (3, 200, 2), (5, 2000, 7)])
def test_pose_only_ba_shipped_default(mvo, O, ctx, F, L, seed, mfma):
    """is_ba_fix_map_points: true (config.yaml:123)."""
    mvo.debug_set("ba_mfma", mfma)
    try:
        st, sto = _check(mvo, O, ctx, mvo.synth.ba_problem(F, L, seed), fix_points=True)
        assert st["iterations"] >= 1
    finally:
        mvo.debug_set("ba_mfma", 1)


def _fix(F, k):
    f = np.zeros(F, np.uint8)
    f[:k] = 1
    return f


@pytest.mark.parametrize("mfma", [1, 0])
@pytest.mark.parametrize("iters", [1, 2, 4])
@pytest.mark.parametrize("F,L,seed,nfix", [(2, 60, 3, 1), (5, 500, 4, 0), (5, 2000, 7, 2), (3, 100, 5, 0)])
def test_full_ba_first_iterations_tight(mvo, O, ctx, F, L, seed, nfix, iters, mfma):
    """A few LM iterations are pure linear algebra: the Schur path must agree with the oracle to rounding."""
    mvo.debug_set("ba_mfma", mfma)
    try:
        pb = mvo.synth.ba_problem(F, L, seed)
        kw = dict(fix_points=False, pose_fixed=_fix(F, nfix) if nfix else None, max_iterations=iters)
        P, X, st = ctx.bundle_adjustment(*_args(pb), **kw)
        Po, Xo, sto = O.bundle_adjustment(*_args(pb), **kw)
        msg = "gpu %s\noracle %s" % (st, sto)
        assert st["iterations"] == sto["iterations"] and st["trials"] == sto["trials"], msg
        assert np.abs(P - Po).max() < 1e-8 and np.abs(X - Xo).max() < 1e-8, msg
        assert abs(st["chi2_final"] - sto["chi2_final"]) < 1e-8 * sto["chi2_final"], msg
        assert abs(st["lambda_final"] - sto["lambda_final"]) < 1e-6 * sto["lambda_final"], msg
    finally:
        mvo.debug_set("ba_mfma", 1)


@pytest.mark.parametrize("mfma", [1, 0])
@pytest.mark.parametrize("F,L,seed", [(3, 200, 3), (5, 500, 4), (5, 2000, 7)])
def test_full_ba_gauge_anchored(mvo, O, ctx, F, L, seed, mfma):
    """Two fixed poses remove the 7-dof gauge (similarity) -> unique optimum -> <= 1e-4 on poses and landmarks."""
    mvo.debug_set("ba_mfma", mfma)
    try:
        pb = mvo.synth.ba_problem(F, L, seed)
        pb["poses0"][:2] = pb["poses_gt"][:2]
        _check(mvo, O, ctx, pb, fix_points=False, pose_fixed=_fix(F, 2))
    finally:
        mvo.debug_set("ba_mfma", 1)


@pytest.mark.parametrize("wgs", [1, 2, 5, 16, 64])
@pytest.mark.parametrize("fix_points", [True, False])
def test_workgroup_split_does_not_change_the_result(mvo, O, ctx, wgs, fix_points):
    """The window is split over G workgroups by landmark; any G must give the oracle's answer."""
    mvo.debug_set("ba_wgs", wgs)
    try:
        pb = mvo.synth.ba_problem(4, 700, 21)
        kw = dict(fix_points=fix_points, pose_fixed=None if fix_points else _fix(4, 1), max_iterations=3)
        P, X, st = ctx.bundle_adjustment(*_args(pb), **kw)
        Po, Xo, sto = O.bundle_adjustment(*_args(pb), **kw)
        assert st["trials"] == sto["trials"], (st, sto)
        assert np.abs(P - Po).max() < 1e-8 and np.abs(X - Xo).max() < 1e-8, (st, sto)
        pb["poses0"][:2] = pb["poses_gt"][:2]
        _check(mvo, O, ctx, pb, fix_points=fix_points, pose_fixed=_fix(4, 2))
    finally:
        mvo.debug_set("ba_wgs", 0)


def _reproj(pb, P, X):
    r = []
    for i in range(len(P)):
        Tcw = np.linalg.inv(P[i])
        sel = pb["edge_pose"] == i
        pc = X[pb["edge_point"][sel]] @ Tcw[:3, :3].T + Tcw[:3, 3]
        r.append(pb["focal"] * pc[:, :2] / pc[:, 2:] + [pb["cx"], pb["cy"]] - pb["edge_uv"][sel])
    return np.concatenate(r)


@pytest.mark.parametrize("nfix", [0, 1])
def test_full_ba_faithful_gauge_free(mvo, O, ctx, nfix):
    """The reference fixes no vertex (g2o_ba.cpp:210-211 commented out): the 7-dof gauge (6-dof scale with pose 0
    fixed) is held only by the LM damping, so 50 iterations do not pin poses/landmarks to 1e-4 -- the path taken
    depends on accept/reject decisions at rounding level.  Compared here: the gauge-invariant quantities."""
    pb = mvo.synth.ba_problem(5, 800, 9)
    kw = dict(fix_points=False, pose_fixed=_fix(5, nfix) if nfix else None)
    P, X, st = ctx.bundle_adjustment(*_args(pb), **kw)
    Po, Xo, sto = O.bundle_adjustment(*_args(pb), **kw)
    assert abs(st["chi2_final"] - sto["chi2_final"]) < 5e-3 * sto["chi2_final"], (st, sto)
    assert st["chi2_final"] < 0.1 * st["chi2_initial"]
    r, ro = _reproj(pb, P, X), _reproj(pb, Po, Xo)
    assert np.abs(r - ro).max() < 0.5, np.abs(r - ro).max()      # pixels


def gauge_aligned_landmark_error(pb, P, X, Po, Xo):
    """Similarity-aligns (P, X) onto (Po, Xo) and returns (max relative landmark difference over the landmarks observed from
    at least two poses, number of those, indices of the single-view landmarks that differ by more than 1e-4, max relative
    camera-centre difference).  Single-view landmarks have no depth: their position along the ray is only held by the LM
    damping."""
    ma, mr = X.mean(0), Xo.mean(0)
    A, Bm = X - ma, Xo - mr
    U_, S_, Vt = np.linalg.svd(Bm.T @ A)
    D = np.eye(3)
    D[2, 2] = np.sign(np.linalg.det(U_ @ Vt))
    R = U_ @ D @ Vt
    sc = (S_ * np.diag(D)).sum() / (A ** 2).sum()
    t = mr - sc * R @ ma
    Xa, Ca = sc * X @ R.T + t, sc * P[:, :3, 3] @ R.T + t
    d = np.linalg.norm(Xa - Xo, axis=1) / np.abs(Xo).max()
    views = np.zeros(len(Xo), int)
    for p_, l_ in set(zip(pb["edge_pose"].tolist(), pb["edge_point"].tolist())):
        views[l_] += 1
    multi = views >= 2
    loose_single = np.nonzero((d > 1e-4) & ~multi)[0]
    return d[multi].max(), int(multi.sum()), loose_single, np.abs(Ca - Po[:, :3, 3]).max() / max(np.abs(Po[:, :3, 3]).max(), 1e-12)


def test_bench_window_matches_the_sequential_oracle_up_to_gauge(mvo, O, ctx):
    """The benchmarked BA5 window (no fixed vertex) against the SEQUENTIAL oracle -- an independent program with another
    summation order: once the 7-dof gauge is factored out, EVERY landmark with a depth (>= 2 views) agrees within the
    north-star 1e-4 (measured ~1e-6), and so do the camera centres; the only landmarks beyond it are single-view ones."""
    pb = mvo.synth.ba_problem(5, 2000, 7)
    P, X, st = ctx.bundle_adjustment(*_args(pb), fix_points=False)
    Po, Xo, sto = O.bundle_adjustment(*_args(pb), fix_points=False)
    dmax, nmulti, loose, dc = gauge_aligned_landmark_error(pb, P, X, Po, Xo)
    assert nmulti > 1900 and dmax < 1e-4 and dc < 1e-4, (dmax, nmulti, dc)
    assert len(loose) < 20
    assert abs(st["chi2_final"] - sto["chi2_final"]) < 1e-6 * sto["chi2_final"]


def test_ba10_converged_against_the_sequential_oracle(mvo, O, ctx):
    """BA10 shape, anchored, custom information matrix, run until the LM loop stops by itself (50 iterations leave this
    low-parallax window path-dependent, see test_ba10_and_information_matrix): poses and landmarks of the device and of the
    sequential oracle then agree far below 1e-4."""
    pb = mvo.synth.ba_problem(10, 1500, 11, width=1242, height=375, K=mvo.synth.KITTI_K)
    pb["poses0"][:2] = pb["poses_gt"][:2]
    kw = dict(fix_points=False, pose_fixed=_fix(10, 2), info=(2.0, 0.3, 0.3, 1.5), huber_delta=1.5, max_iterations=1000)
    P, X, st = ctx.bundle_adjustment(*_args(pb), **kw)
    Po, Xo, sto = O.bundle_adjustment(*_args(pb), **kw)
    assert st["terminated"] and sto["terminated"], (st, sto)
    assert np.abs(P - Po).max() < 1e-8 and _rel(X, Xo) < 1e-6, (np.abs(P - Po).max(), _rel(X, Xo))


def test_ba10_and_information_matrix(mvo, O, ctx):
    pb = mvo.synth.ba_problem(10, 1500, 11, width=1242, height=375, K=mvo.synth.KITTI_K)
    pb["poses0"][:2] = pb["poses_gt"][:2]
    # poses agree to 1e-4; the landmarks of this low-parallax window are NOT converged after 50 iterations (the
    # oracle's own 50- vs 300-iteration landmarks differ by 3.5e-3), so their 50-iteration state depends on the
    # LM path at rounding level: compared at 2e-3 here, at 1e-8 for the first iterations above.
    _check(mvo, O, ctx, pb, tol_x=2e-3, fix_points=False, pose_fixed=_fix(10, 2), info=(2.0, 0.3, 0.3, 1.5),
           huber_delta=1.5)
    _check(mvo, O, ctx, pb, fix_points=True, info=(2.0, 0.3, 0.3, 1.5))


def test_noise_free_optimum_is_ground_truth(mvo, ctx):
    pb = mvo.synth.ba_problem(4, 300, seed=1, pix_noise=0, outlier_frac=0, point_noise=0, f32_storage=False)
    P, X, st = ctx.bundle_adjustment(pb["poses0"], pb["points_gt"], pb["edge_pose"], pb["edge_point"], pb["edge_uv"],
                                     pb["focal"], pb["cx"], pb["cy"], fix_points=True)
    assert np.abs(P - pb["poses_gt"]).max() < 1e-8 and st["chi2_final"] < 1e-10


def test_degenerate_windows_and_errors(mvo, O, ctx):
    pb = mvo.synth.ba_problem(3, 50, 5)
    P, X, st = ctx.bundle_adjustment(*_args(pb), fix_points=True, max_iterations=0)
    assert st["iterations"] == 0 and np.abs(P - pb["poses0"]).max() < 1e-12
    e = np.zeros(0, np.int32)
    P, X, st = ctx.bundle_adjustment(np.zeros((0, 16)), np.zeros((0, 3)), e, e, np.zeros((0, 2)), 500, 320, 240)
    assert len(P) == 0
    P, X, st = ctx.bundle_adjustment(*_args(pb), fix_points=True, pose_fixed=np.ones(3, np.uint8))
    assert st["iterations"] == 0 and np.abs(P - pb["poses0"]).max() < 1e-12
    with pytest.raises(mvo.MvoError):
        ctx.bundle_adjustment(pb["poses0"], pb["points0"], pb["edge_pose"] + 10, pb["edge_point"], pb["edge_uv"],
                              pb["focal"], pb["cx"], pb["cy"])
    # duplicate (pose, point) observations accumulate like g2o's shared Hessian block
    pb2 = mvo.synth.ba_problem(3, 80, 6)
    ep = np.concatenate([pb2["edge_pose"], pb2["edge_pose"][:40]])
    el = np.concatenate([pb2["edge_point"], pb2["edge_point"][:40]])
    uv = np.concatenate([pb2["edge_uv"], pb2["edge_uv"][:40] + 0.3])
    fixed = np.array([1, 0, 0], np.uint8)
    a = ctx.bundle_adjustment(pb2["poses0"], pb2["points0"], ep, el, uv, pb2["focal"], pb2["cx"], pb2["cy"],
                              fix_points=False, pose_fixed=fixed, max_iterations=3)
    b = O.bundle_adjustment(pb2["poses0"], pb2["points0"], ep, el, uv, pb2["focal"], pb2["cx"], pb2["cy"],
                            fix_points=False, pose_fixed=fixed, max_iterations=3)
    assert np.abs(a[0] - b[0]).max() < 1e-8 and np.abs(a[1] - b[1]).max() < 1e-8


def test_reference_style_bundle_adjustment(mvo, O):
    """optimization::bundleAdjustment signature (g2o_ba.h:23-30) with pointer-list stand-ins."""
    mvo.reset_default_context()
    pb = mvo.synth.ba_problem(3, 120, 8)
    v2d, vidx = [], []
    for i in range(3):
        sel = pb["edge_pose"] == i
        v2d.append(pb["edge_uv"][sel].astype(np.float32))
        vidx.append((pb["edge_point"][sel] + 1000).tolist())          # map-point ids need not be 0..L-1
    pts = {int(i + 1000): pb["points0"][i].astype(np.float32) for i in np.unique(pb["edge_point"])}
    poses = [pb["poses0"][i].copy() for i in range(3)]
    K = np.array([[pb["focal"], 0, pb["cx"]], [0, 516.5, pb["cy"]], [0, 0, 1]])
    before = {k: v.copy() for k, v in pts.items()}
    mvo.bundleAdjustment(v2d, vidx, K, pts, poses, np.eye(2), True, False)
    assert all(np.array_equal(before[k], pts[k]) for k in pts)          # fixed + not updated
    Po, _, _ = O.bundle_adjustment(*_args(pb), fix_points=True)
    assert np.abs(np.array(poses) - Po).max() < TOL
    mvo.reset_default_context()


def test_resident_window_can_be_resolved_repeatedly(mvo, O, ctx):
    pb = mvo.synth.ba_problem(5, 600, 12)
    pb["poses0"][:2] = pb["poses_gt"][:2]
    kw = dict(fix_points=False, pose_fixed=_fix(5, 2))
    h = ctx.ba_prepare(*_args(pb), **kw)
    res = []
    for _ in range(3):
        ctx.ba_solve_resident(h)
        res.append(ctx.ba_fetch(h))
    ctx.ba_release(h)
    Po, Xo, sto = O.bundle_adjustment(*_args(pb), **kw)
    for P, X, st in res:
        assert np.array_equal(P, res[0][0]) and np.array_equal(X, res[0][1])      # deterministic re-solve
        assert _rel(P[:, :3, 3], Po[:, :3, 3]) < TOL and _rel(X, Xo) < TOL


def test_config4_ba10_window(mvo, O, ctx):
    """BASELINE configs[3]: 10-keyframe BA, 4000 landmarks, ~40k edges (KITTI-shaped): exercises G = 128 workgroups,
    the 4-tile matrix-core Schur path and the LDS LDL^T (n = 60 > 32)."""
    pb = mvo.synth.ba_problem(10, 4000, 13, width=1242, height=375, K=mvo.synth.KITTI_K)
    assert len(pb["edge_pose"]) > 30000
    kw = dict(fix_points=False, max_iterations=3)
    P, X, st = ctx.bundle_adjustment(*_args(pb), **kw)
    Po, Xo, sto = O.bundle_adjustment(*_args(pb), **kw)
    assert st["trials"] == sto["trials"], (st, sto)
    assert np.abs(P - Po).max() < 1e-8 and np.abs(X - Xo).max() < 1e-7, (st, sto)
    pb["poses0"][:2] = pb["poses_gt"][:2]
    P, X, st = ctx.bundle_adjustment(*_args(pb), fix_points=False, pose_fixed=_fix(10, 2))
    Po, Xo, sto = O.bundle_adjustment(*_args(pb), fix_points=False, pose_fixed=_fix(10, 2))
    assert _rel(P[:, :3, 3], Po[:, :3, 3]) < TOL and np.abs(P[:, :3, :3] - Po[:, :3, :3]).max() < TOL, (st, sto)
    assert ctx.debug_ba_phases()["wgs"] >= 56


# ---------------------------------------------------------------------------------------------------------------------
# Bit-exact parity: the device declares the association of every sum of the solve; the oracle restates the
# same LM algorithm with that blocked order (oracle/ba_blocked_oracle.cpp).  Then ALL 50 iterations must agree exactly:
# every trial's damping, robust chi2, gain ratio and accept/reject decision, and the final poses and landmarks bit for bit.
def _bitwise(mvo, O, ctx, pb, trace_out=None, **kw):
    ctx.ba_trace_enable(True)
    try:
        P, X, st = ctx.bundle_adjustment(*_args(pb), **kw)
        tr = ctx.ba_trace()
        plan = ctx.ba_plan()
    finally:
        ctx.ba_trace_enable(False)
    Po, Xo, sto, tro = O.bundle_adjustment_blocked(*_args(pb), plan=plan, **kw)
    msg = "plan %s\ngpu %s\noracle %s" % ({k: (v if k != "wg_pt_start" else len(v)) for k, v in plan.items()}, st, sto)
    n = min(len(tr), len(tro))
    def bits(a):   # bit patterns, any NaN encoding counts as the same NaN (host and device quiet NaNs differ in sign)
        a = np.ascontiguousarray(a[:n]).copy()
        a[np.isnan(a)] = 1.2345678e304
        return a.view(np.uint64)
    bad = np.nonzero(((bits(tr) != bits(tro)) | (np.isnan(tr[:n]) != np.isnan(tro[:n]))).any(1))[0]
    assert len(bad) == 0, msg + "\nfirst differing trial %d:\n gpu    %r\n oracle %r" % (bad[0], tr[bad[0]], tro[bad[0]])
    assert st["trials"] == sto["trials"] and st["iterations"] == sto["iterations"] and st["terminated"] == sto["terminated"], msg
    assert st["chi2_initial"] == sto["chi2_initial"] and st["chi2_final"] == sto["chi2_final"], msg
    assert st["lambda_final"] == sto["lambda_final"], msg
    assert np.array_equal(P, Po), msg + "\nmax pose difference %g" % np.abs(P - Po).max()
    assert np.array_equal(X, Xo), msg + "\nmax landmark difference %g" % np.abs(X - Xo).max()
    if trace_out is not None:
        trace_out.append(tr)
    return st, plan


def test_bitwise_the_benchmarked_window_all_50_iterations(mvo, O, ctx):
    """The exact window bench.py solves (BA5: 5 poses / 2000 landmarks / ~9.4k edges, seed 7, NO fixed vertex, 50
    iterations): north-star '1e-4 on poses and landmarks' is met with zero difference."""
    st, plan = _bitwise(mvo, O, ctx, mvo.synth.ba_problem(5, 2000, 7), fix_points=False)
    assert st["iterations"] == 50 and st["trials"] > 60 and plan["wgs"] == 28  # (one XCD minus the 4 CUs left to other kernels)


def _stale_step_window(mvo, O, ctx):
    """A window of bench.py's pool (shard 3, window 0) whose 14-workgroup solve runs into g2o's stale-step rule: a failed
    factorisation leaves the solver's x what it was, OptimizationAlgorithmLevenberg applies and scores it with tempChi = DBL_MAX,
    and a negative computeScale() of that stale step ACCEPTS it (g2o_ba.cpp:196-197 picks LinearSolverDense, :288 runs the loop)."""
    pb = mvo.synth.ba_problem(5, 2000, 3007)
    out = []
    st, plan = _bitwise(mvo, O, ctx, pb, trace_out=out, fix_points=False)
    tr = out[0]
    failed = tr[:, 1] > 1e300
    assert plan["wgs"] == 14 and failed.sum() > 5 and (failed & (tr[:, 3] > 0)).sum() > 0, (plan["wgs"], failed.sum(), (failed & (tr[:, 3] > 0)).sum())
    assert (failed & (tr[:, 3] == 0)).sum() > 0          # and the usual outcome (positive stale scale: rejected) next to it
    return st


def test_bitwise_stale_step_after_a_failed_solve(mvo, O):
    c = mvo.Context(0)
    try:
        c.ba_set_mode("throughput")
        mvo.debug_set("ba_service", 2)
        _stale_step_window(mvo, O, c)
    finally:
        mvo.debug_set("ba_service", 1)
        c.close()


def _many_pose_windows(mvo, O, ctx):
    """Windows of more than 10 free poses (vo.h keeps up to 20 frames): the reduced system has more than 63 unknowns -- the LDS solver,
    four lanes per row in the pivot ranks, and in the pose-only form (equal x / y translation entries in every pose block) the replay of
    Eigen's swaps with two positions per lane."""
    pb = mvo.synth.ba_problem(12, 700, 41)
    _bitwise(mvo, O, ctx, pb, fix_points=False, max_iterations=4)
    _bitwise(mvo, O, ctx, pb, fix_points=True, max_iterations=6)
    pb = mvo.synth.ba_problem(14, 500, 42)
    _bitwise(mvo, O, ctx, pb, fix_points=False, pose_fixed=_fix(14, 1), max_iterations=3)
    _bitwise(mvo, O, ctx, pb, fix_points=True, max_iterations=3)
    # beyond what the LDS-resident solver holds (the reduced system of 18 free poses alone is 96 KB): a clean capacity error
    pb = mvo.synth.ba_problem(20, 300, 43)
    with pytest.raises(mvo.MvoError) as e:
        ctx.bundle_adjustment(*_args(pb), fix_points=False, pose_fixed=_fix(20, 2), max_iterations=2)
    assert e.value.code == -3 and "too large" in str(e.value)


def test_bitwise_windows_of_more_than_ten_poses(mvo, O, ctx):
    _many_pose_windows(mvo, O, ctx)


def _bench_windows(mvo):
    """The first windows of shard 0 / shard 1 of bench.py's pool (bench.window_pool: seed 7 + 1000 x shard + k)."""
    return [mvo.synth.ba_problem(5, 2000, 7), mvo.synth.ba_problem(5, 2000, 8), mvo.synth.ba_problem(5, 2000, 1007)]


@pytest.mark.parametrize("route", ["resident_grid", "launch_path"])
def test_bitwise_throughput_cut_and_service(mvo, O, route):
    """The flavour bench.py's headline number runs (g2o_ba.cpp:193-289 semantics, write-back :298-316): THROUGHPUT mode, the
    window cut into 14 workgroups (~670 observations per range: the second edge of a thread in LDS, measurements re-read from
    device memory, two chunks of U) -- once through the resident solver grid (k_ba_service<32,2>: inputs read from the pinned
    image, slots pulling from mailboxes) and once with the same cut on the launch path (k_ba_lm<false,32,2>; MVO_BA_MODE_SHARED
    is the mode that always takes it).  Held to the blocked oracle trial by trial ON THE DEVICE, not only in the emulator."""
    c = mvo.Context(0)
    try:
        if route == "resident_grid":
            c.ba_set_mode("throughput")
            mvo.debug_set("ba_service", 2)          # (the default policy brings the grid up under load only)
        else:
            c.ba_set_mode("shared")
        c.ba_launch_stats(reset=True)
        for pb in _bench_windows(mvo):
            st, plan = _bitwise(mvo, O, c, pb, fix_points=False)
            assert plan["wgs"] == 14 and (plan["nsplit"] & 0xFFFF) >= 2 and st["iterations"] == 50 and st["trials"] > 60, (plan["wgs"], plan["nsplit"], st)
        stats = c.ba_launch_stats()
        if route == "resident_grid":
            assert stats["resident_windows"] >= 3, stats
        else:
            assert stats["resident_windows"] == 0, stats
        _bitwise(mvo, O, c, _bench_windows(mvo)[0], fix_points=True)      # pose-only windows always take the launch path
    finally:
        mvo.debug_set("ba_service", 1)
        c.close()


def test_bitwise_ba10_windows_sharing_launches(mvo, O):
    """BASELINE configs[3] the way bench.py --width 1242 --height 375 --max-kp 4000 --ba-poses 10 --ba-points 4000 --streams 8 loads
    the device: several sequences submit BA10 windows at once, the launch thread packs ~4 of them (56 workgroups each, Schur
    partials summed per XCD group first) into one grid.  Every one of them still equals the blocked oracle bit for bit."""
    import threading
    pbs = [mvo.synth.ba_problem(10, 4000, 13 + k, width=1242, height=375, K=mvo.synth.KITTI_K) for k in range(4)]
    errors, plans = [], []
    gate = threading.Barrier(len(pbs))

    def work(k):
        try:
            c = mvo.Context(0)
            for _ in range(2):
                gate.wait()
                st, plan = _bitwise(mvo, O, c, pbs[k], fix_points=False)
                plans.append((plan["wgs"], plan["groups"], st["iterations"]))
            c.close()
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)[:2000]))
            gate.abort()

    th = [threading.Thread(target=work, args=(k,)) for k in range(len(pbs))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[0]
    assert len(plans) == 8 and all(g >= 56 and k >= 2 and it == 50 for g, k, it in plans), plans


def test_bitwise_config4_ba10_window(mvo, O, ctx):
    """BASELINE configs[3]: 10 keyframes / 4000 landmarks / ~36k edges (KITTI shape), no fixed vertex."""
    pb = mvo.synth.ba_problem(10, 4000, 13, width=1242, height=375, K=mvo.synth.KITTI_K)
    st, plan = _bitwise(mvo, O, ctx, pb, fix_points=False)
    assert st["iterations"] == 50 and plan["wgs"] >= 56


@pytest.mark.parametrize("mfma", [1, 0])
@pytest.mark.parametrize("case", ["pose_only", "anchored", "fixed0_info", "tiny_one_range", "ragged_chain_tails",
                                  "dups", "seven_poses"])
def test_bitwise_variants(mvo, O, ctx, case, mfma):
    mvo.debug_set("ba_mfma", mfma)
    try:
        if case == "pose_only":
            _bitwise(mvo, O, ctx, mvo.synth.ba_problem(5, 2000, 7), fix_points=True)
        elif case == "anchored":
            pb = mvo.synth.ba_problem(5, 2000, 7)
            pb["poses0"][:2] = pb["poses_gt"][:2]
            _bitwise(mvo, O, ctx, pb, fix_points=False, pose_fixed=_fix(5, 2))
        elif case == "fixed0_info":
            _bitwise(mvo, O, ctx, mvo.synth.ba_problem(4, 700, 21), fix_points=False, pose_fixed=_fix(4, 1),
                     info=(2.0, 0.3, 0.3, 1.5), huber_delta=1.5)
        elif case == "tiny_one_range":
            st, plan = _bitwise(mvo, O, ctx, mvo.synth.ba_problem(3, 40, 5), fix_points=False)
            assert plan["wgs"] == 1
        elif case == "ragged_chain_tails":
            # every remainder of the chain length modulo the unrolled group of four steps, with one and with several
            # column pieces; the first window is the committed golden one (tests/golden/ba_3x40.npz)
            _bitwise(mvo, O, ctx, mvo.synth.ba_problem(3, 40, 77), fix_points=False, max_iterations=3)
            for L in (37, 41, 42, 43, 150):
                _bitwise(mvo, O, ctx, mvo.synth.ba_problem(3, L, 100 + L), fix_points=False, max_iterations=12)
                _bitwise(mvo, O, ctx, mvo.synth.ba_problem(3, L, 200 + L), fix_points=False, max_iterations=12,
                         pose_fixed=_fix(3, 1))
        elif case == "dups":
            pb = mvo.synth.ba_problem(3, 300, 6)
            pb = dict(pb, edge_pose=np.concatenate([pb["edge_pose"], pb["edge_pose"][:90]]),
                      edge_point=np.concatenate([pb["edge_point"], pb["edge_point"][:90]]),
                      edge_uv=np.concatenate([pb["edge_uv"], pb["edge_uv"][:90] + 0.3]))
            _bitwise(mvo, O, ctx, pb, fix_points=False, pose_fixed=_fix(3, 1))
        else:
            _bitwise(mvo, O, ctx, mvo.synth.ba_problem(7, 1500, 31), fix_points=False)
    finally:
        mvo.debug_set("ba_mfma", 1)


def test_batched_windows_equal_their_single_solves_bit_for_bit(mvo, ctx):
    """mvo_ba_solve_batch: windows of different sizes in ONE call -- the launch thread packs them into grids (up to 16
    windows, never more workgroups than CUs, one solver class per grid); every window must come out exactly as when it is
    solved alone (same plan, same sums; which XCD it lands on only selects the hand-off flavour)."""
    pbs = [mvo.synth.ba_problem(5, 2000, 7), mvo.synth.ba_problem(3, 300, 11), mvo.synth.ba_problem(4, 700, 21),
           mvo.synth.ba_problem(3, 40, 77), mvo.synth.ba_problem(5, 1500, 9), mvo.synth.ba_problem(2, 150, 3)]
    pbs = pbs + [mvo.synth.ba_problem(3, 200 + 10 * k, 100 + k) for k in range(8)]          # 14 windows
    pbs.append(mvo.synth.ba_problem(7, 1500, 31))                                            # another solver class (n = 42)
    singles = [ctx.bundle_adjustment(*_args(pb), fix_points=False, max_iterations=15) for pb in pbs]
    batch = ctx.ba_solve_batch([_args(pb) for pb in pbs], fix_points=False, max_iterations=15)
    assert len(batch) == len(pbs)
    for k, ((P, X, st), (Pb, Xb, stb)) in enumerate(zip(singles, batch)):
        assert np.array_equal(P, Pb) and np.array_equal(X, Xb), (k, np.abs(P - Pb).max())
        assert st["trials"] == stb["trials"] and st["chi2_final"] == stb["chi2_final"], (k, st, stb)
    # pose-only windows go through the same path
    single = ctx.bundle_adjustment(*_args(pbs[0]), fix_points=True)
    both = ctx.ba_solve_batch([_args(pbs[0]), _args(pbs[1])], fix_points=True)
    assert np.array_equal(single[0], both[0][0])
