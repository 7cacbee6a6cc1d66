#!/bin/bash
# round 4, GPU call W: the reduced system of the 64-row class inside the U area (two chunks of U instead of three for BA10)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04w
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ba.py -x -q 2>&1 | tail -1
MVO_BA_ALIAS_SL=0 python tools/ba10_probe.py 0 2>&1 | tail -2
python tools/ba10_probe.py 0 2>&1 | tail -2
show() { python - $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"] or {}
    print(sys.argv[2], "value", round(d["value"], 1), r.get("kernel"), "avg_launch_ms", round(r.get("avg_launch_ms", 0), 2), "windows/launch", round(r.get("windows_per_launch", 0), 2))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
C4="python bench.py --width 1242 --height 375 --max-kp 4000 --ba-poses 10 --ba-points 4000 --streams 8 --steps 6 --warmup 1 --no-cpu-baseline --no-secondary"
MVO_BA_ALIAS_SL=0 timeout 300 $C4 > $O/c4_old.json 2> $O/c4_old.err; show $O/c4_old.json config4_own_lds
timeout 300 $C4 > $O/c4_new.json 2> $O/c4_new.err; show $O/c4_new.json config4_alias
timeout 300 $C4 --streams 12 > $O/c4_new12.json 2> $O/c4_new12.err; show $O/c4_new12.json config4_alias_streams12
