#!/bin/bash
# round 4, GPU call I: admission gate of the extraction launches -- capacity sweep under the 24-shard load
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r04i
mkdir -p $O
show() { python - $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"] or {}
    print(sys.argv[2], "value", round(d["value"], 1), "avg_window_ms", round(r.get("avg_window_ms", 0), 3), "in flight", round(r.get("windows_in_flight", 0), 2), "host", d["secondary"].get("headline_host_us_per_frame"), "busy", d["secondary"].get("headline_shard_busy_ms"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline"
for c in 0 4 6 8 10 12; do
  MVO_EXTRACT_CONCURRENCY=$c timeout 300 $B > $O/cap$c.json 2> $O/cap$c.err; show $O/cap$c.json cap$c
done
for c in 6 8; do
  MVO_EXTRACT_CONCURRENCY=$c timeout 300 $B --streams 32 > $O/cap${c}_s32.json 2> $O/cap${c}_s32.err; show $O/cap${c}_s32.json cap${c}_streams32
done
