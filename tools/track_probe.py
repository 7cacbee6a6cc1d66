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
