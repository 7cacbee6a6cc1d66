#!/bin/bash
# round 4, GPU call B: round-3 solver back + short-circuit; why is a window slower in the loaded grid than alone?
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04b
mkdir -p $O
timeout 60 tools/probes/solve_probe > $O/solve_probe.txt 2>&1; cat $O/solve_probe.txt
timeout 300 python tools/ba_probe.py 28,13 > $O/ba_probe.txt 2>&1; grep "ms/solve\|instrumented" $O/ba_probe.txt | cut -c1-600 | head -12
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline"
run() { # name, env..., extra args
  name=$1; shift
  env "$@" timeout 300 $B $EXTRA > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"] or {}
    print(sys.argv[2], "value", round(d["value"], 1), "avg_window_ms", round(r.get("avg_window_ms", 0), 3), "kcycles", r.get("avg_window_kcycles"), "clock", r.get("shader_clock_ghz_under_load"), "in flight", round(r.get("windows_in_flight", 0), 2), "host", d["secondary"].get("headline_host_us_per_frame"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
EXTRA="" run default MVO_X=0
EXTRA="" run slotmap1 MVO_BA_SLOT_MAP=1
EXTRA="--streams 2" run svc_streams2 MVO_BA_SERVICE=2
EXTRA="--streams 8" run svc_streams8 MVO_BA_SERVICE=2
EXTRA="--streams 16" run svc_streams16 MVO_BA_SERVICE=2
EXTRA="--streams 32" run streams32 MVO_X=0
