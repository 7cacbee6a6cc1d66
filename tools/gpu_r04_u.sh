#!/bin/bash
# round 4, GPU call U: tracking rows -- windows on the launch path with the throughput cut (13 workgroups) instead of the latency cut
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04u
mkdir -p $O
show() { python - $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"] or {}
    print(sys.argv[2], "value", round(d["value"], 1), r.get("kernel"), "avg_launch_ms", round(r.get("avg_launch_ms", 0), 2), "windows/launch", r.get("windows_per_launch"), "host", d["secondary"].get("headline_host_us_per_frame"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
B="python bench.py --gpus 1 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --track"
timeout 300 $B > $O/t_default.json 2> $O/t_default.err; show $O/t_default.json track_default
MVO_BA_SERVICE=0 timeout 300 $B > $O/t_svc0.json 2> $O/t_svc0.err; show $O/t_svc0.json track_launch_path_throughput_cut
MVO_BA_SERVICE=0 timeout 300 $B --streams 32 > $O/t_svc0_32.json 2> $O/t_svc0_32.err; show $O/t_svc0_32.json track_launch_path_throughput_cut_32
MVO_BA_SERVICE=0 MVO_BA_WGS=20 timeout 300 $B > $O/t_svc0_g20.json 2> $O/t_svc0_g20.err; show $O/t_svc0_g20.json track_launch_path_G20
MVO_BA_SERVICE=0 MVO_BA_WGS=16 timeout 300 $B > $O/t_svc0_g16.json 2> $O/t_svc0_g16.err; show $O/t_svc0_g16.json track_launch_path_G16
