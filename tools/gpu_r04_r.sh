#!/bin/bash
# round 4, GPU call R: tracking rows in the loop -- which path for the windows, how many shards, gate or not
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r04r
mkdir -p $O
show() { python - $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"] or {}
    print(sys.argv[2], "value", round(d["value"], 1), r.get("kernel"), "avg_window_ms", r.get("avg_window_ms"), "avg_launch_ms", round(r.get("avg_launch_ms", 0), 2), "host", d["secondary"].get("headline_host_us_per_frame"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
B="python bench.py --gpus 1 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --track"
timeout 300 $B > $O/t_default.json 2> $O/t_default.err; show $O/t_default.json track_default_32
timeout 300 $B --streams 24 > $O/t_s24.json 2> $O/t_s24.err; show $O/t_s24.json track_streams24
MVO_BA_SERVICE=2 timeout 300 $B > $O/t_svc.json 2> $O/t_svc.err; show $O/t_svc.json track_service_forced_32
MVO_BA_SERVICE=2 timeout 300 $B --streams 24 > $O/t_svc24.json 2> $O/t_svc24.err; show $O/t_svc24.json track_service_forced_24
MVO_EXTRACT_CONCURRENCY=0 timeout 300 $B > $O/t_nogate.json 2> $O/t_nogate.err; show $O/t_nogate.json track_nogate_32
MVO_BA_SERVICE=2 MVO_EXTRACT_CONCURRENCY=16 timeout 300 $B > $O/t_svc_c16.json 2> $O/t_svc_c16.err; show $O/t_svc_c16.json track_service_forced_cap16
