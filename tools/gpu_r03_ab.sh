#!/bin/bash
# A/B on one box: the tree of an earlier commit (built into _ab_old/) against the current one
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03ab
mkdir -p $O
pr() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], round(d["value"]), round(r.get("frac"), 4), r.get("launches"), round(r.get("windows_in_flight") or 0, 2), round(r.get("avg_window_ms") or 0, 3), d.get("secondary", {}).get("headline_host_us_per_frame"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for i in 1 2; do
  (cd _ab_old && timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/old_$i.json 2> $O/old_$i.err); pr $O/old_$i.json
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/new_$i.json 2> $O/new_$i.err; pr $O/new_$i.json
done


timeout 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-secondary > $O/new_k60.json 2> $O/new_k60.err; pr $O/new_k60.json
