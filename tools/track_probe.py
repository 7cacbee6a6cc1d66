"""Dev probe: tracking rows on the GPU box -- wall time per call, kernel times (HIP events), oracle time beside it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, __graft_entry__ as g
mvo = g.load_package(); orc = g.load_oracle(); ctx = mvo.Context(0)
for name, kw in [("default n~1100, 25% outliers", {}), ("50% outliers", dict(outlier_frac=0.5)), ("n~4500", dict(n_map=12000))]:
    pr = mvo.synth.tracking_problem(seed=11, **kw)
    p3, p2, K = pr["pts3d"], pr["pts2d"], pr["K"]
    m = ctx.map_create(); ctx.map_upload(m, pr["map_pos"], pr["map_desc"])
    for _ in range(5):
        ctx.solve_pnp_ransac(p3, p2, K); ctx.map_points_in_view(m, pr["T_w_c"], K, 640, 480, cap=len(pr["map_pos"]))
    ctx.profile_enable(True); ctx.profile_reset()
    t_pnp, t_view = [], []
    for _ in range(50):
        t0 = time.perf_counter(); r = ctx.solve_pnp_ransac(p3, p2, K); t_pnp.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); ctx.map_points_in_view(m, pr["T_w_c"], K, 640, 480, cap=len(pr["map_pos"])); t_view.append(time.perf_counter() - t0)
    prof = ctx.profile_get(); ctx.profile_enable(False)
    dbg = ctx.debug_pnp()
    t0 = time.perf_counter()
    for _ in range(20): ro = orc.solve_pnp_ransac(p3, p2, K)
    t_orc = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(20): orc.map_in_view(pr["map_pos"], pr["T_w_c"], K, 640, 480)
    t_orc_v = (time.perf_counter() - t0) / 20
    print("== %s: n=%d map=%d inliers=%d iters_run=%d lm_iters=%d lm_evals=%d" % (name, len(p3), len(pr["map_pos"]), len(r["inliers"]), dbg["iters_run"], dbg["lm_iters"], dbg["lm_evals"]))
    print("   solve_pnp_ransac wall median %.3f ms (oracle, %d iterations: %.3f ms)   map_points_in_view wall %.3f ms (oracle %.3f ms)" % (
        np.median(t_pnp) * 1e3, ro["iters_run"], t_orc * 1e3, np.median(t_view) * 1e3, t_orc_v * 1e3))
    for k, (n, ms) in sorted(prof.items()):
        print("   %-20s %4d launches  avg %.1f us" % (k, n, ms / n * 1e3))
    ctx.map_release(m)
print("== keyframe row")
for name, kw in [("n=1000, 20% wrong", dict(n=1000, outlier_frac=0.2)), ("n=1000, 50% wrong", dict(n=1000, outlier_frac=0.5)), ("n=3000, 70% wrong", dict(n=3000, outlier_frac=0.7))]:
    kf = mvo.synth.keyframe_problem(seed=8, **kw)
    T = kf["T_curr_to_prev"]
    for _ in range(3):
        inl = ctx.find_essential_inliers(kf["kp_ref"], kf["kp_cur"], kf["K"]); ctx.triangulate_points(kf["kp_ref"][inl], kf["kp_cur"][inl], kf["K"], T[:3, :3], T[:3, 3])
    ctx.profile_enable(True); ctx.profile_reset()
    t_e, t_t = [], []
    for _ in range(30):
        t0 = time.perf_counter(); inl = ctx.find_essential_inliers(kf["kp_ref"], kf["kp_cur"], kf["K"]); t_e.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); ctx.triangulate_points(kf["kp_ref"][inl], kf["kp_cur"][inl], kf["K"], T[:3, :3], T[:3, 3]); t_t.append(time.perf_counter() - t0)
    prof = ctx.profile_get(); ctx.profile_enable(False); dbg = ctx.debug_essential()
    t0 = time.perf_counter()
    for _ in range(5): ro = orc.find_essential_inliers(kf["kp_ref"], kf["kp_cur"], kf["K"])
    t_o = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for _ in range(5): orc.triangulate_points(kf["kp_ref"][inl], kf["kp_cur"][inl], kf["K"], T[:3, :3], T[:3, 3])
    t_ot = (time.perf_counter() - t0) / 5
    print("   %s: inliers %d, loop ran %d iterations, evaluated %d" % (name, len(inl), dbg["iters_run"], dbg["evaluated"]))
    print("   find_essential_inliers wall median %.3f ms (oracle %.3f ms)   triangulate_points (%d) wall %.3f ms (oracle %.3f ms)" % (
        np.median(t_e) * 1e3, t_o * 1e3, len(inl), np.median(t_t) * 1e3, t_ot * 1e3))
    for k, (n_, ms) in sorted(prof.items()):
        print("   %-20s %4d launches  avg %.1f us" % (k, n_, ms / n_ * 1e3))
