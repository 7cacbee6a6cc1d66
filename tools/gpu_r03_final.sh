#!/bin/bash
# final bench lines on the final code (the rocprofv3 / PMC summaries come from tools/collect_evidence.sh)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03
mkdir -p $O
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_k20.json 2> $O/bench_driver_k20.err
timeout 200 python bench.py --track --no-cpu-baseline --no-secondary > $O/bench_track.json 2> $O/bench_track.err
MVO_HOST_TIMING=1 timeout 120 python bench.py --streams 1 --pipeline 0 --ba-mode none --steps 420 --warmup 10 --no-cpu-baseline --no-secondary > $O/bench_extract_match_alone.json 2> $O/bench_extract_match_alone.err; grep "mvo host" $O/bench_extract_match_alone.err | tail -2 > $O/host_stage_times.txt; cat $O/host_stage_times.txt
python - <<'PY'
import json
for f in ("bench_default", "bench_driver_k20", "bench_track", "bench_extract_match_alone"):
    d = json.loads(open("gpurun_out/r03/%s.json" % f).read().strip().splitlines()[-1])
    r = d["roofline"] or {}
    print(f, round(d["value"], 1), r.get("kernel"), r.get("frac"), r.get("windows_in_flight"), r.get("traffic"), r.get("traffic_source"))
    s = d.get("secondary", {})
    print("   ", {k: (round(v) if isinstance(v, float) else v) for k, v in s.items() if k.endswith("_fps")}, d.get("cpu_baseline", {}) and d["cpu_baseline"].get("value"), d.get("cpu_baseline_all_threads", {}) and d["cpu_baseline_all_threads"].get("value"))
PY
