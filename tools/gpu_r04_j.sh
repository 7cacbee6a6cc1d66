#!/bin/bash
# round 4, GPU call J: CUs per solver slot (13 / 12 / 11) with the extraction gate
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r04j
mkdir -p $O
show() { python - $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"] or {}
    print(sys.argv[2], "value", round(d["value"], 1), "avg_window_ms", round(r.get("avg_window_ms", 0), 3), "in flight", round(r.get("windows_in_flight", 0), 2), "host", d["secondary"].get("headline_host_us_per_frame"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline"
for cfg in "6 8 24" "6 8 32" "6 10 24" "6 10 32" "8 10 32" "6 12 32"; do
  set -- $cfg
  MVO_EXTRACT_CONCURRENCY=$1 MVO_BA_XCD_RESERVE=$2 timeout 300 $B --streams $3 > $O/c$1_r$2_s$3.json 2> $O/c$1_r$2_s$3.err; show $O/c$1_r$2_s$3.json cap$1_reserve$2_streams$3
done
