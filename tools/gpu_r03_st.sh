#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
cat > /tmp/probe.py <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
mvo = g.load_package(); ctx = mvo.Context(0)
pb = mvo.synth.ba_problem(5, 2000, 7)
a = (pb["poses0"], pb["points0"], pb["edge_pose"], pb["edge_point"], pb["edge_uv"], pb["focal"], pb["cx"], pb["cy"])
for wgs in (13, 28):
    mvo.debug_set("ba_wgs", wgs)
    ctx.ba_trace_enable(True)
    h = ctx.ba_prepare(*a, fix_points=False)
    mvo.debug_set("ba_profile", 1)
    for _ in range(3): ctx.ba_solve_resident(h); P, X, st = ctx.ba_fetch(h)
    ph = ctx.debug_ba_phases()
    raw = ctx.ba_trace(h, raw_rows=412)[400:406].ravel()
    print("wgs", ph["wgs"], "nsplit", ctx.ba_plan(h)["nsplit"], "stamps 0..23 diffs:", [int(raw[i + 1] - raw[i]) for i in range(23)])
    print("    ", {k: round(v / max(st["trials"], 1)) for k, v in ph.items() if k not in ("wgs", "x15", "schur.loop", "schur.wait", "schur.acc")})
    mvo.debug_set("ba_profile", 0)
    ctx.ba_release(h)
PY
timeout 100 python /tmp/probe.py 2>&1 | grep -v amdgpu.ids
