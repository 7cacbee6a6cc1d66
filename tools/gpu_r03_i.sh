#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03j
mkdir -p $O
for st in 24 48; do
  timeout 300 python bench.py --steps 40 --no-cpu-baseline --no-secondary --streams $st > $O/bench_s${st}.json 2> $O/bench_s${st}.err
  python - "$st" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r03j/bench_s%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
    print("streams", sys.argv[1], round(d["value"]), round(d["roofline"]["avg_launch_ms"], 3), round(d["roofline"]["windows_per_launch"], 2), d["roofline"]["launch_thread_ms"], d["secondary"].get("headline_host_us_per_frame"))
except Exception as e:
    print(sys.argv[1:], "unreadable", e)
PY
done
