#!/bin/bash
# round 3, GPU call A: parity on the MI355X, single-window timings per cut / solver, bench variants
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03a
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest_gpu.log
timeout 600 python tools/ba_probe.py 32,16 0,1 > $O/ba_probe.log 2>&1; echo "probe rc $?"; grep -E "ms/solve|BA10|ms/call|block" $O/ba_probe.log
timeout 600 python bench.py --steps 60 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
for f in ("bench_default",):
    try:
        d = json.loads(open("gpurun_out/r03a/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["windows_per_launch"], d["secondary"].get("single_sequence_fps"), d["roofline"].get("launch_thread_ms"))
    except Exception as e:
        print(f, "unreadable", e)
PY
for v in "latency:--ba-cut latency" "share256:--ba-cut throughput" "blk:--ba-cut throughput" "s32:--streams 32" "s16:--streams 16"; do
  name=${v%%:*}; args=${v#*:}
  envs=""
  [ "$name" = "share256" ] && envs="MVO_BA_CU_SHARE=256"
  [ "$name" = "blk" ] && envs="MVO_BA_BLOCK_SOLVER=1"
  env $envs timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-secondary $args > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$name" <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r03a/bench_%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["windows_per_launch"])
except Exception as e:
    print(f, "unreadable", e)
PY
done
