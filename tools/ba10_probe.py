"""Dev probe: per-phase cycles of the BA10 window (BASELINE configs[3]: 10 poses / 4000 landmarks / ~36k edges) on the instrumented kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, __graft_entry__ as g
mvo = g.load_package(); ctx = mvo.Context(0)
pb = mvo.synth.ba_problem(10, 4000, 13, width=1242, height=375, K=mvo.synth.KITTI_K)
a = (pb["poses0"], pb["points0"], pb["edge_pose"], pb["edge_point"], pb["edge_uv"], pb["focal"], pb["cx"], pb["cy"])
for wgs in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0"])]:
    mvo.debug_set("ba_wgs", wgs)
    ctx.ba_trace_enable(True)
    try:
        h = ctx.ba_prepare(*a, fix_points=False)
    except Exception as e:
        print("wgs", wgs, "prepare failed:", e); continue
    for prof in (0, 1):
        mvo.debug_set("ba_profile", prof)
        for _ in range(2): ctx.ba_solve_resident(h); ctx.ba_fetch(h)
        t0 = time.perf_counter(); N = 5
        for _ in range(N): ctx.ba_solve_resident(h); P, X, st = ctx.ba_fetch(h)
        dt = (time.perf_counter() - t0) / N
        ph = ctx.debug_ba_phases()
        if not prof:
            print("BA10 wgs", ph["wgs"], "ms/solve %.3f trials %d same_l2 %d" % (dt * 1e3, st["trials"], ph["x15"]))
        else:
            print("   instrumented ms/solve %.3f" % (dt * 1e3), {k: round(v / max(st["trials"], 1)) for k, v in ph.items() if k not in ("wgs", "x15")})
            raw = ctx.ba_trace(h, raw_rows=412)[400:408].ravel()
            print("   trial 6, cycles from the end of the Schur exchange (stamp 11): ranks formed %d, group sum + stores %d, barrier %d, replay %d, S assembled %d, solved %d"
                  % tuple(int(raw[i] - raw[11]) for i in (24, 25, 26, 27, 12, 13)))
    mvo.debug_set("ba_profile", 0)
    ctx.ba_release(h)
