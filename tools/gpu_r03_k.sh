#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03n
mkdir -p $O
cat > /tmp/svc_test.py <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import __graft_entry__ as g
mvo = g.load_package(); O = g.load_oracle()
from test_gpu_ba import _bitwise, _args
ctx = mvo.Context(0); ctx.ba_set_mode("throughput")
pb = mvo.synth.ba_problem(5, 2000, 7)
t0 = time.time(); st, plan = _bitwise(mvo, O, ctx, pb, fix_points=False); print("service bitwise ok", st["trials"], plan["wgs"], "%.2fs" % (time.time() - t0))
for _ in range(3): ctx.bundle_adjustment(*_args(pb), fix_points=False)
t0 = time.perf_counter(); N = 20
for _ in range(N): ctx.bundle_adjustment(*_args(pb), fix_points=False)
print("service one-shot ms/call %.3f" % ((time.perf_counter() - t0) / N * 1e3), ctx.ba_launch_stats())
c2 = mvo.Context(0)
P, X, s2 = c2.bundle_adjustment(*_args(pb), fix_points=False); print("latency-mode launch in between ok", s2["trials"])
P, X, s3 = ctx.bundle_adjustment(*_args(pb), fix_points=False); print("service again ok", s3["trials"])
PY
echo skip
echo skip
for v in "svc:X=1" "nosvc:MVO_BA_SERVICE=0"; do
  name=${v%%:*}; envs=${v#*:}
  for st in 24 32; do
  env $envs timeout 200 python bench.py --steps 60 --no-cpu-baseline --no-secondary --streams $st > $O/bench_${name}_s$st.json 2> $O/bench_${name}_s$st.err
  python - "$name" "$st" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r03n/bench_%s_s%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1], sys.argv[2], round(d["value"]), round(r["frac"], 4), r.get("avg_window_ms"), r.get("windows_in_flight"), round(r["avg_launch_ms"], 3), r["windows_per_launch"], d["secondary"].get("headline_host_us_per_frame"))
except Exception as e:
    print(sys.argv[1:], "unreadable", e)
PY
  done
done
