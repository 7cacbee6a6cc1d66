#!/bin/bash
# round 4, GPU call A: solver probe, GPU suite, per-phase cycles, driver-command bench on the code with the short-circuited
# failed solves + the scalar-forwarding reduced solve
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04a
mkdir -p $O
timeout 60 tools/probes/solve_probe > $O/solve_probe.txt 2>&1; cat $O/solve_probe.txt
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 300 python tools/ba_probe.py 28,13 > $O/ba_probe.txt 2>&1; grep -v "^$" $O/ba_probe.txt | cut -c1-700 | head -40
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 300 $O/bench_driver.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04a/bench_driver.json").read().strip().splitlines()[-1])
r = d["roofline"] or {}
print("value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), r.get("kernel"), "frac", r.get("frac"), "avg_window_ms", r.get("avg_window_ms"), "in flight", r.get("windows_in_flight"))
print({k: (round(v) if isinstance(v, float) else v) for k, v in d.get("secondary", {}).items() if k.endswith("_fps")})
print(d.get("kernels"))
print(d["secondary"].get("headline_host_us_per_frame"), d["secondary"].get("single_sequence_host_us_per_frame"))
PY
