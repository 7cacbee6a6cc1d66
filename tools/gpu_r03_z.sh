#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03z2
mkdir -p $O
pr() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], round(d["value"]), round(r.get("frac"), 4), r.get("launches"), round(r.get("windows_in_flight") or 0, 2), round(r.get("avg_window_ms") or 0, 3), d.get("secondary", {}).get("headline_host_us_per_frame"), d["secondary"].get("headline_shard_busy_ms"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for s in 24 28 32 24 28 32; do timeout 200 python bench.py --streams $s --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/s$s.json 2> $O/s$s.err; pr $O/s$s.json; done
