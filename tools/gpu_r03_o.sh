#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03o
mkdir -p $O
pr() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], round(d["value"]), round(r.get("frac"), 4), round(r.get("windows_in_flight") or 0, 2), d.get("secondary", {}).get("headline_host_us_per_frame"), d["secondary"].get("headline_shard_busy_ms"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for v in a0:0 b1:1 a0b:0 b1b:1; do n=${v%%:*}; o=${v#*:}; MVO_BENCH_CTX_ORDER=$o timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/$n.json 2> $O/$n.err; pr $O/$n.json; done
