#!/bin/bash
# round 4: shard-count sweep after the resident service's threshold change (does any load between "one sequence" and "32" fall into a hole?)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04sweep
mkdir -p $O
for n in 4 8 12 16 24; do
  timeout 120 python bench.py --gpus 1 --streams $n --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/s$n.json 2> $O/s$n.err < /dev/null
  python - $O/s$n.json $n <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"] or {}
    print("streams", sys.argv[2], "frames/s", round(d["value"], 1), r.get("kernel"), "launches", r.get("launches"), "windows/launch", round(r.get("windows_per_launch", 0), 2), "avg_launch_ms", round(r.get("avg_launch_ms", 0), 2))
except Exception as e:
    print("streams", sys.argv[2], "failed", e)
PY
done
