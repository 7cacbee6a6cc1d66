#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03r2
mkdir -p $O
pr() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], round(d["value"]), round(r.get("frac"), 4), round(r.get("windows_in_flight") or 0, 2), round(r.get("avg_window_ms") or 0, 3), d.get("secondary", {}).get("headline_host_us_per_frame"), d["secondary"].get("headline_shard_busy_ms"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for v in r6:6:20 r4:4:20 r6b:6:60 r4b:4:60 r4c:4:20; do n=${v%%:*}; rest=${v#*:}; r=${rest%%:*}; k=${rest#*:}; MVO_BA_XCD_RESERVE=$r timeout 200 python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-secondary > $O/$n.json 2> $O/$n.err; pr $O/$n.json; done
