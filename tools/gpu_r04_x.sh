#!/bin/bash
# round 4, GPU call X: block solver with look-ahead (panel b + 1 beside update b)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04x
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ba.py -x -q 2>&1 | tail -1
python tools/ba10_probe.py 0 2>&1 | tail -2
show() { python - $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"] or {}
    print(sys.argv[2], "value", round(d["value"], 1), r.get("kernel"), "avg_launch_ms", round(r.get("avg_launch_ms", 0), 2), "windows/launch", round(r.get("windows_per_launch", 0), 2), "frac", r.get("frac"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
C4="python bench.py --width 1242 --height 375 --max-kp 4000 --ba-poses 10 --ba-points 4000 --streams 8 --steps 6 --warmup 1 --no-cpu-baseline --no-secondary"
timeout 300 $C4 > $O/c4.json 2> $O/c4.err; show $O/c4.json config4
timeout 300 $C4 --streams 6 > $O/c4_6.json 2> $O/c4_6.err; show $O/c4_6.json config4_streams6
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/head.json 2> $O/head.err; show $O/head.json headline
