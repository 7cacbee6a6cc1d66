#!/bin/bash
# round 4, GPU call K: speculative second damping -- solve times, GPU BA suite, bench with the extraction gate
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r04k
mkdir -p $O
timeout 300 python tools/ba_probe.py 28,13 > $O/ba_probe.txt 2>&1; grep "ms/solve\|instrumented\|BA10" $O/ba_probe.txt | cut -c1-600 | head -5
timeout 300 python -m pytest tests/test_gpu_ba.py tests/test_gpu_concurrency.py -x -q 2>&1 | tail -2
show() { python - $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"] or {}
    print(sys.argv[2], "value", round(d["value"], 1), "avg_window_ms", round(r.get("avg_window_ms", 0), 3), "in flight", round(r.get("windows_in_flight", 0), 2), "host", d["secondary"].get("headline_host_us_per_frame"), {k: round(v) for k, v in d["secondary"].items() if k.endswith("_fps")})
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
MVO_EXTRACT_CONCURRENCY=8 timeout 400 $B > $O/cap8.json 2> $O/cap8.err; show $O/cap8.json cap8_with_secondary
MVO_EXTRACT_CONCURRENCY=6 timeout 300 $B --no-secondary --streams 32 > $O/cap6_s32.json 2> $O/cap6_s32.err; show $O/cap6_s32.json cap6_streams32
