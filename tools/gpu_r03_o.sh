#!/bin/bash
# A/B on one box: two full contexts (= two streams) per sequence vs a context + its sibling (mvo_create_sibling)
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
for v in separate:1 sibling:0 separate_b:1 sibling_b:0; do n=${v%%:*}; o=${v#*:}; MVO_BENCH_SEPARATE_CTX=$o timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/$n.json 2> $O/$n.err; pr $O/$n.json; done
