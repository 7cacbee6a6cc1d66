"""Dev probe: BA solve time and per-phase cycles across workgroup counts (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, __graft_entry__ as g
mvo = g.load_package(); ctx = mvo.Context(0)
kinds = (("full", dict(fix_points=False)), ("pose_only", dict(fix_points=True)))
wgs_list = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
blk_list = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
for kind, kw in kinds:
 for blk in blk_list:
  mvo.debug_set("ba_edge_rows", blk - 1 if blk < 2 else 1)
  print("== edge rows knob (0 = auto, 2 = registers)", blk)
  for wgs in wgs_list:
    mvo.debug_set("ba_wgs", wgs)
    pb = mvo.synth.ba_problem(5, 2000, 7)
    a = (pb["poses0"], pb["points0"], pb["edge_pose"], pb["edge_point"], pb["edge_uv"], pb["focal"], pb["cx"], pb["cy"])
    ctx.ba_trace_enable(True)
    try:
        h = ctx.ba_prepare(*a, **kw)
    except Exception as e:
        print(kind, wgs, "prepare failed", e); continue
    for prof in (0, 1):
        mvo.debug_set("ba_profile", prof)
        for _ in range(3): ctx.ba_solve_resident(h); ctx.ba_fetch(h)
        t0=time.perf_counter(); N=10
        for _ in range(N): ctx.ba_solve_resident(h); P,X,st = ctx.ba_fetch(h)
        dt=(time.perf_counter()-t0)/N
        ph = ctx.debug_ba_phases()
        if not prof:
            print(kind, "wgs", ph["wgs"], "ms/solve %.3f (wall, production kernel) trials %d same_l2 %d | rounds with a second system %d, first one failed as guessed %d | kcycles %.0f"
                  % (dt*1e3, st["trials"], ph["x15"], ph["schur.loop"], ph["schur.wait"], ph["schur.acc"] / 1e3))
        else:
            tot = ph["total"]
            raw = ctx.ba_trace(h, raw_rows=412)[400:408].ravel()
            print("   timeline of trial 6 (cycles between stamps 0..23):", [int(raw[i + 1] - raw[i]) for i in range(23)])
            print("   pivot order inside 11 -> 12 (stamps 24..27 relative to 11; 12 relative to 11):", [int(raw[i] - raw[11]) for i in (24, 25, 26, 27, 12)])
            print("   instrumented: ms/solve %.3f" % (dt*1e3), "cyc/us %.0f" % (tot/ (dt*1e6)), {k: round(v/max(st["trials"],1)) for k,v in ph.items() if k not in ("wgs", "x15")})
    mvo.debug_set("ba_profile", 0)
    ctx.ba_release(h)
# one-shot path (window rebuilt per call) and batches
mvo.debug_set("ba_edge_rows", -1)
pb = mvo.synth.ba_problem(5, 2000, 7)
a = (pb["poses0"], pb["points0"], pb["edge_pose"], pb["edge_point"], pb["edge_uv"], pb["focal"], pb["cx"], pb["cy"])
mvo.debug_set("ba_wgs", 0)
for _ in range(3): ctx.bundle_adjustment(*a, fix_points=False)
t0=time.perf_counter(); N=20
for _ in range(N): ctx.bundle_adjustment(*a, fix_points=False)
print("mvo_bundle_adjustment (plan + upload + solve + fetch) ms/call %.3f" % ((time.perf_counter()-t0)/N*1e3))
# BASELINE configs[3]: BA10 window (10 poses / 4000 landmarks / ~36k edges)
pb = mvo.synth.ba_problem(10, 4000, 13, width=1242, height=375, K=mvo.synth.KITTI_K)
a = (pb["poses0"], pb["points0"], pb["edge_pose"], pb["edge_point"], pb["edge_uv"], pb["focal"], pb["cx"], pb["cy"])
h = ctx.ba_prepare(*a, fix_points=False)
for _ in range(2): ctx.ba_solve_resident(h); ctx.ba_fetch(h)
t0=time.perf_counter(); N=5
for _ in range(N): ctx.ba_solve_resident(h); P,X,st = ctx.ba_fetch(h)
print("BA10 resident solve ms %.3f trials %d wgs %d" % ((time.perf_counter()-t0)/N*1e3, st["trials"], ctx.debug_ba_phases()["wgs"]))
ctx.ba_release(h)
