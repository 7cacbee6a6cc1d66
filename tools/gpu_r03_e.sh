#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03e
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_ba.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest_ba.log
timeout 600 python tools/ba_probe.py 28,14 0,2 > $O/ba_probe.log 2>&1; echo "probe rc $?"; grep -E "ms/solve|BA10|ms/call|timeline" $O/ba_probe.log | grep -v pose_only | cut -c1-420 | head -12
timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-secondary > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-secondary --streams 32 > $O/bench_s32.json 2> $O/bench_s32.err
python - <<'PY'
import json
for f in ("default", "s32"):
    try:
        d = json.loads(open("gpurun_out/r03e/bench_%s.json" % f).read().strip().splitlines()[-1])
        print(f, round(d["value"]), round(d["roofline"]["frac"], 4), round(d["roofline"]["avg_launch_ms"], 3), round(d["roofline"]["windows_per_launch"], 2))
    except Exception as e:
        print(f, "unreadable", e)
PY
