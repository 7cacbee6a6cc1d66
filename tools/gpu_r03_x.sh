#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03y
mkdir -p $O
cat > /tmp/probe.py <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
mvo = g.load_package(); ctx = mvo.Context(0)
pb = mvo.synth.ba_problem(5, 2000, 7)
a = (pb["poses0"], pb["points0"], pb["edge_pose"], pb["edge_point"], pb["edge_uv"], pb["focal"], pb["cx"], pb["cy"])
for pieces in (1, 0):
    mvo.debug_set("ba_chunk_pieces", pieces)
    for wgs in (13, 14, 28):
        mvo.debug_set("ba_wgs", wgs)
        h = ctx.ba_prepare(*a, fix_points=False)
        for prof in (0, 1):
            mvo.debug_set("ba_profile", prof)
            for _ in range(3): ctx.ba_solve_resident(h); ctx.ba_fetch(h)
            t0 = time.perf_counter(); N = 10
            for _ in range(N): ctx.ba_solve_resident(h); P, X, st = ctx.ba_fetch(h)
            dt = (time.perf_counter() - t0) / N
            ph = ctx.debug_ba_phases()
            if not prof: print("pieces-knob", pieces, "wgs", ph["wgs"], "nsplit", ctx.ba_plan(h)["nsplit"], "ms/solve %.3f trials %d" % (dt * 1e3, st["trials"]))
            else: print("    ", {k: round(v / max(st["trials"], 1)) for k, v in ph.items() if k not in ("wgs", "x15", "schur.loop", "schur.wait", "schur.acc")})
        mvo.debug_set("ba_profile", 0)
        ctx.ba_release(h)
PY
timeout 200 python /tmp/probe.py 2>&1 | tee $O/probe.log
pr() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1], round(d["value"]), round(r.get("frac"), 4), r.get("launches"), round(r.get("windows_in_flight") or 0, 2), round(r.get("avg_window_ms") or 0, 3), d.get("secondary", {}).get("headline_host_us_per_frame"), d["kernel_ms_per_frame"])
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for k in 20 60 20; do timeout 300 python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_k$k.json 2> $O/bench_k$k.err; pr $O/bench_k$k.json; done
