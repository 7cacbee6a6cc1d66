#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03q
mkdir -p $O
for hq in 16 24 32; do for rep in 1 2; do
  GPU_MAX_HW_QUEUES=$hq timeout 200 python bench.py --steps 100 --no-cpu-baseline --no-secondary > $O/bench_q${hq}_$rep.json 2> $O/bench_q${hq}_$rep.err
  python - "$hq" "$rep" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r03q/bench_q%s_%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("hwq", sys.argv[1], "rep", sys.argv[2], round(d["value"]), round(r.get("avg_window_ms") or 0, 3), round(r.get("windows_in_flight") or 0, 2), d["secondary"].get("headline_host_us_per_frame"))
except Exception as e:
    print(sys.argv[1:], "unreadable", e)
PY
done; done
