#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03p
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03p/bench_driver.json").read().strip().splitlines()[-1])
print(round(d["value"]), d["roofline"]["frac"], d["cpu_baseline"], d["cpu_baseline_all_threads"]); print(d["secondary"])
PY
timeout 300 python bench.py --width 1242 --height 375 --max-kp 4000 --ba-poses 10 --ba-points 4000 --streams 4 --steps 30 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_config4.json 2> $O/bench_config4.err
timeout 300 python bench.py --width 1242 --height 375 --max-kp 4000 --ba-poses 10 --ba-points 4000 --streams 8 --steps 30 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_config4_s8.json 2> $O/bench_config4_s8.err
python - <<'PY'
import json
for f in ("bench_config4", "bench_config4_s8"):
    try:
        d = json.loads(open("gpurun_out/r03p/%s.json" % f).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f, round(d["value"]), r["frac"], r["avg_launch_ms"], r["windows_per_launch"])
    except Exception as e:
        print(f, "unreadable", e)
PY
