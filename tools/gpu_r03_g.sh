#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03g2
mkdir -p $O
cat > /tmp/probe.py <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
mvo = g.load_package(); ctx = mvo.Context(0)
pb = mvo.synth.ba_problem(5, 2000, 7)
a = (pb["poses0"], pb["points0"], pb["edge_pose"], pb["edge_point"], pb["edge_uv"], pb["focal"], pb["cx"], pb["cy"])
for wgs in (28, 32, 40, 48, 56, 64):
    mvo.debug_set("ba_wgs", wgs)
    try:
        h = ctx.ba_prepare(*a, fix_points=False)
    except Exception as e:
        print(wgs, "failed", e); continue
    for prof in (0, 1):
        mvo.debug_set("ba_profile", prof)
        for _ in range(3): ctx.ba_solve_resident(h); ctx.ba_fetch(h)
        t0 = time.perf_counter(); N = 10
        for _ in range(N): ctx.ba_solve_resident(h); P, X, st = ctx.ba_fetch(h)
        dt = (time.perf_counter() - t0) / N
        ph = ctx.debug_ba_phases()
        if not prof: print("wgs", ph["wgs"], "nsplit", ctx.ba_plan(h)["nsplit"], "ms/solve %.3f trials %d same_l2 %d" % (dt * 1e3, st["trials"], ph["x15"]))
        else: print("    ", {k: round(v / max(st["trials"], 1)) for k, v in ph.items() if k not in ("wgs", "x15", "schur.loop", "schur.wait", "schur.acc")})
    mvo.debug_set("ba_profile", 0)
    ctx.ba_release(h)
PY
timeout 200 python /tmp/probe.py 2>&1 | grep -v amdgpu.ids | tee $O/probe.log
