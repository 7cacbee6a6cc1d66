#!/bin/bash
# round 4, GPU call E: how the host waits (ROCr interrupt vs polling), per box
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04e
mkdir -p $O
nproc; uptime; cat /proc/cmdline | cut -c1-300
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline"
run() { name=$1; shift
  env "$@" timeout 300 $B $EXTRA > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"] or {}
    print(sys.argv[2], "value", round(d["value"], 1), "avg_window_ms", round(r.get("avg_window_ms", 0), 3), "in flight", round(r.get("windows_in_flight", 0), 2), "host", d["secondary"].get("headline_host_us_per_frame"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
EXTRA="" run default MVO_X=0
EXTRA="" run nointerrupt HSA_ENABLE_INTERRUPT=0
EXTRA="" run default2 MVO_X=0
EXTRA="" run nointerrupt2 HSA_ENABLE_INTERRUPT=0
EXTRA="--streams 1 --pipeline 0 --ba-mode none --steps 40" run one_extract_default MVO_HOST_TIMING=1
grep "mvo host" $O/bench_one_extract_default.err | tail -2
EXTRA="--streams 1 --pipeline 0 --ba-mode none --steps 40" run one_extract_nointerrupt MVO_HOST_TIMING=1 HSA_ENABLE_INTERRUPT=0
grep "mvo host" $O/bench_one_extract_nointerrupt.err | tail -2
EXTRA="--streams 32" run streams32_nointerrupt HSA_ENABLE_INTERRUPT=0
