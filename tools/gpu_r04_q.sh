#!/bin/bash
# round 4, GPU call Q: full GPU suite + the driver's command after the host trims
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04q
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 300 $O/bench_driver.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04q/bench_driver.json").read().strip().splitlines()[-1])
r = d["roofline"] or {}
print("value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), "frac", round(r.get("frac", 0), 4), "avg_window_ms", r.get("avg_window_ms"), "in flight", r.get("windows_in_flight"))
print({k: (round(v) if isinstance(v, float) else v) for k, v in d.get("secondary", {}).items() if k.endswith("_fps")})
print(d["secondary"].get("headline_host_us_per_frame"), d["secondary"].get("single_sequence_host_us_per_frame"))
PY
