#!/bin/bash
# round 4, GPU call O: with the extraction cheaper, how many CUs per solver slot?
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r04o
mkdir -p $O
show() { python - $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"] or {}
    print(sys.argv[2], "value", round(d["value"], 1), "avg_window_ms", round(r.get("avg_window_ms", 0), 3), "in flight", round(r.get("windows_in_flight", 0), 2), "host", d["secondary"].get("headline_host_us_per_frame"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline"
timeout 300 $B > $O/default.json 2> $O/default.err; show $O/default.json default_2x13
MVO_BA_XCD_RESERVE=4 timeout 300 $B > $O/r4.json 2> $O/r4.err; show $O/r4.json reserve4_2x14
MVO_BA_XCD_RESERVE=4 timeout 300 $B --streams 28 > $O/r4s28.json 2> $O/r4s28.err; show $O/r4s28.json reserve4_2x14_streams28
MVO_BA_XCD_RESERVE=2 timeout 300 $B > $O/r2.json 2> $O/r2.err; show $O/r2.json reserve2_2x15
timeout 300 $B --streams 28 > $O/s28.json 2> $O/s28.err; show $O/s28.json default_streams28
timeout 300 $B --streams 32 > $O/s32.json 2> $O/s32.err; show $O/s32.json default_streams32
