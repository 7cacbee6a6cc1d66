#!/bin/bash
# round 4, GPU call C: window time in the resident grid vs slots busy, with and without extraction beside it; pose-block phase
# with tables made once + pipelined operand loads
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04c
mkdir -p $O
for cfg in "1 40 0" "2 40 0" "8 40 0" "16 40 0" "24 40 0" "16 40 1" "24 40 1"; do
  timeout 120 python tools/svc_stress.py $cfg 2>&1 | grep "^N" | cut -c1-330 | tee -a $O/svc_stress.txt
done
timeout 300 python tools/ba_probe.py 28,13 > $O/ba_probe.txt 2>&1; grep "ms/solve\|instrumented" $O/ba_probe.txt | cut -c1-600 | head -4
timeout 300 python -m pytest tests/test_gpu_ba.py -x -q 2>&1 | tail -2
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python - $O/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"] or {}
print("value", round(d["value"], 1), "avg_window_ms", round(r.get("avg_window_ms", 0), 3), "kcycles", r.get("avg_window_kcycles"), "clock", r.get("shader_clock_ghz_under_load"), "in flight", round(r.get("windows_in_flight", 0), 2), "host", d["secondary"].get("headline_host_us_per_frame"))
PY
