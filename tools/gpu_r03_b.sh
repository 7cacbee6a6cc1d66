#!/bin/bash
# round 3, GPU call B: BA parity + single-window timings + bench (quick)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03b
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ba.py tests/test_gpu_concurrency.py -x -q > $O/pytest_ba.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_ba.log
timeout 600 python tools/ba_probe.py 0,28,14,16 0 > $O/ba_probe.log 2>&1; echo "probe rc $?"; grep -E "ms/solve|BA10|ms/call|timeline" $O/ba_probe.log | cut -c1-400
for v in "default:" "latency:--ba-cut latency --no-secondary" "res0:--no-secondary" "s32:--streams 32 --no-secondary" "s16:--streams 16 --no-secondary"; do
  name=${v%%:*}; args=${v#*:}
  envs="X=1"
  [ "$name" = "res0" ] && envs="MVO_BA_XCD_RESERVE=0"
  env $envs timeout 300 python bench.py --steps 60 --no-cpu-baseline $args > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$name" <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r03b/bench_%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(d["value"]), round(d["roofline"]["frac"], 4), round(d["roofline"]["avg_launch_ms"], 3), round(d["roofline"]["windows_per_launch"], 2), d["roofline"].get("launch_thread_ms"), d["secondary"].get("single_sequence_fps"), d["secondary"].get("single_sequence_host_us_per_frame"))
except Exception as e:
    print(f, "unreadable", e)
PY
done
