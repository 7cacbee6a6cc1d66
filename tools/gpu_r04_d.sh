#!/bin/bash
# round 4, GPU call D: the window body of the resident grid out of line (own register allocation) vs inlined; launch path with
# the same cut; bench twice per variant (box noise)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04d
mkdir -p $O
echo "== service inlined"; MVO_BA_SERVICE_OOL=0 timeout 120 python tools/svc_stress.py 1 60 0 2 2>&1 | grep "^N" | cut -c1-200
echo "== service out of line"; MVO_BA_SERVICE_OOL=1 timeout 120 python tools/svc_stress.py 1 60 0 2 2>&1 | grep "^N\|rror" | cut -c1-200
echo "== launch path, throughput cut"; timeout 120 python tools/svc_stress.py 1 60 0 0 2>&1 | grep "^N" | cut -c1-200
MVO_BA_SERVICE_OOL=1 timeout 300 python -m pytest tests/test_gpu_ba.py tests/test_gpu_concurrency.py -x -q 2>&1 | tail -2
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline"
for v in 0 1 0 1; do
  MVO_BA_SERVICE_OOL=$v timeout 300 $B > $O/bench_ool$v.json 2> $O/bench_ool$v.err
  python - $O/bench_ool$v.json ool$v <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"] or {}
    print(sys.argv[2], "value", round(d["value"], 1), "avg_window_ms", round(r.get("avg_window_ms", 0), 3), "kcycles", r.get("avg_window_kcycles"), "clock", r.get("shader_clock_ghz_under_load"), "in flight", round(r.get("windows_in_flight", 0), 2), "host", d["secondary"].get("headline_host_us_per_frame"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done
nproc; uptime
