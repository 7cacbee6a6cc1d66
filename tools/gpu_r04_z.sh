#!/bin/bash
# round 4, GPU call Z: the service's "coming" threshold below the launch path's saturated rate -- the driver's command several times in a row
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04z
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_concurrency.py -x -q 2>&1 | tail -1
for i in 1 2 3 4; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/head$i.json 2> $O/head$i.err
  python - $O/head$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("headline", round(d["value"], 1), "launches", r.get("launches"), "in flight", round(r.get("windows_in_flight", 0), 2), "window ms", round(r.get("avg_window_ms", 0), 3))
PY
done
