#!/bin/bash
# (needs _wt/callb like gpu_r04_f.sh)
# round 4, GPU call G: is the slow equilibrium of the working tree the service policy (demand estimate) or the device?
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r04g
mkdir -p $O
show() { python - $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"] or {}
    print(sys.argv[2], "value", round(d["value"], 1), "avg_window_ms", round(r.get("avg_window_ms", 0), 3), "in flight", round(r.get("windows_in_flight", 0), 2), "host", d["secondary"].get("headline_host_us_per_frame"), "busy", d["secondary"].get("headline_shard_busy_ms"), r.get("launch_thread_ms"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline"
cd $ROOT
timeout 300 $B > $O/work_default.json 2> $O/work_default.err; show $O/work_default.json work_default
MVO_BA_SERVICE=2 timeout 300 $B > $O/work_always.json 2> $O/work_always.err; show $O/work_always.json work_service_always
MVO_BA_SERVICE=2 timeout 300 $B --streams 28 > $O/work_always28.json 2> $O/work_always28.err; show $O/work_always28.json work_service_always_streams28
MVO_BA_SERVICE=2 timeout 300 $B --streams 20 > $O/work_always20.json 2> $O/work_always20.err; show $O/work_always20.json work_service_always_streams20
(cd $ROOT/_wt/callb && MVO_BA_SERVICE=2 timeout 300 $B > $O/callb_always.json 2> $O/callb_always.err); show $O/callb_always.json callb_service_always
