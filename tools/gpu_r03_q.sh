#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03q
mkdir -p $O
pr() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], round(d["value"]), round(r.get("frac"), 4), r.get("launches"), round(r.get("windows_in_flight") or 0, 2), d.get("secondary", {}).get("headline_host_us_per_frame"), d["secondary"].get("headline_shard_busy_ms"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for q in 16 24 20 32 16 24; do GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/q$q.json 2> $O/q$q.err; pr $O/q$q.json; done
