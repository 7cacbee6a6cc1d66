#!/bin/bash
# round 4, GPU call P: train-slice size of the matcher under load (LDS image per workgroup = occupancy on the free CUs)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r04p
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
timeout 200 python -m pytest tests/test_gpu_match.py tests/test_gpu_concurrency.py -x -q 2>&1 | tail -1
for sl in 256 128 64; do
  MVO_MATCH_SLICE=$sl timeout 300 $B > $O/s$sl.json 2> $O/s$sl.err; show $O/s$sl.json slice$sl
  MVO_MATCH_SLICE=$sl MVO_BA_XCD_RESERVE=4 timeout 300 $B > $O/s${sl}_r4.json 2> $O/s${sl}_r4.err; show $O/s${sl}_r4.json slice${sl}_2x14
done
