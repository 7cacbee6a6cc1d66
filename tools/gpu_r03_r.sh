#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03s
mkdir -p $O
for st in 20 60 100; do
  timeout 200 python bench.py --steps $st --no-cpu-baseline --no-secondary > $O/bench_k$st.json 2> $O/bench_k$st.err
  python - "$st" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r03s/bench_k%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("steps", sys.argv[1], round(d["value"]), round(r.get("avg_window_ms") or 0, 3), round(r.get("windows_in_flight") or 0, 2), d["secondary"].get("headline_shard_busy_ms"), d["secondary"].get("headline_host_us_per_frame"))
except Exception as e:
    print(sys.argv[1:], "unreadable", e)
PY
done
