"""Dev probe: BA solve time and per-phase cycles across workgroup counts (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, __graft_entry__ as g
mvo = g.load_package(); ctx = mvo.Context(0)
for kind, kw in (("full", dict(fix_points=False)), ("pose_only", dict(fix_points=True))):
  for wgs in (0, 8, 16, 32, 64):
    mvo.debug_set("ba_wgs", wgs)
    pb = mvo.synth.ba_problem(5, 2000, 7)
    a = (pb["poses0"], pb["points0"], pb["edge_pose"], pb["edge_point"], pb["edge_uv"], pb["focal"], pb["cx"], pb["cy"])
    try:
        h = ctx.ba_prepare(*a, **kw)
    except Exception as e:
        print(kind, wgs, "prepare failed", e); continue
    for _ in range(3): ctx.ba_solve_resident(h); ctx.ba_fetch(h)
    t0=time.perf_counter(); N=10
    for _ in range(N): ctx.ba_solve_resident(h); P,X,st = ctx.ba_fetch(h)
    dt=(time.perf_counter()-t0)/N
    ph = ctx.debug_ba_phases()
    tot = ph["total"]; 
    print(kind, "wgs", ph["wgs"], "ms/solve %.3f trials %d" % (dt*1e3, st["trials"]), "cyc/us %.0f" % (tot/ (dt*1e6)), {k: round(v/max(st["trials"],1)) for k,v in ph.items() if k!="wgs"})
    ctx.ba_release(h)
