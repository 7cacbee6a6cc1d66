#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03x
mkdir -p $O
pr() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1], round(d["value"]), round(r.get("frac"), 4), r.get("launches"), round(r.get("windows_in_flight") or 0, 2), round(r.get("avg_window_ms") or 0, 3), d.get("secondary", {}).get("headline_host_us_per_frame"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for v in "ad_w10:X=1:10" "ad_w5:X=1:5" "on_w10:MVO_BA_SERVICE=2:10" "on_w5:MVO_BA_SERVICE=2:5" "ad_w5b:X=1:5" "on_w5b:MVO_BA_SERVICE=2:5"; do
  name=${v%%:*}; rest=${v#*:}; envs=${rest%%:*}; w=${rest#*:}
  env $envs timeout 200 python bench.py --steps 20 --warmup $w --no-cpu-baseline --no-secondary > $O/$name.json 2> $O/$name.err; pr $O/$name.json
done
timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-secondary > $O/k100.json 2> $O/k100.err; pr $O/k100.json
