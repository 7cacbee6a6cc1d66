#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03g
mkdir -p $O
for res in 4 6 8 10; do for st in 24 32; do
  MVO_BA_XCD_RESERVE=$res timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-secondary --streams $st > $O/bench_r${res}_s${st}.json 2> $O/bench_r${res}_s${st}.err
  python - "$res" "$st" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r03g/bench_r%s_s%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
    print("reserve", sys.argv[1], "streams", sys.argv[2], round(d["value"]), round(d["roofline"]["avg_launch_ms"], 3), round(d["roofline"]["windows_per_launch"], 2), d["kernel_ms_per_frame"])
except Exception as e:
    print(sys.argv[1:], "unreadable", e)
PY
done; done
