#!/bin/bash
# round 4, last GPU call: descriptors from whole blurred levels (k_blur + k_brief_sample) for contexts in throughput mode
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04brief
mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_orb.py tests/test_gpu_run_vo.py tests/test_gpu_host_adapter.py -x -q < /dev/null 2>&1 | tail -2
run() { # name, env...
  name=$1; shift
  env "$@" timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/$name.json 2> $O/$name.err < /dev/null
  python - $O/$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]; k = d["kernels"]
    print(sys.argv[2], "frames/s", round(d["value"], 1), "launches", r.get("launches"), "in flight", round(r.get("windows_in_flight", 0), 2), "window ms", round(r.get("avg_window_ms", 0), 3), "k_brief us", k["k_brief"]["avg_launch_us"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run fused_2x13 MVO_BRIEF_LEVEL_BLUR=0
run level_2x13 MVO_BRIEF_LEVEL_BLUR=-1
run level_2x14 MVO_BRIEF_LEVEL_BLUR=-1 MVO_BA_XCD_RESERVE=4
