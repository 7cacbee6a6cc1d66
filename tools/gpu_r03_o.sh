#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03o
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ba.py tests/test_gpu_concurrency.py -x -q > $O/pytest_ba.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest_ba.log
for res in 4 6 8; do for st in 24 32; do
  MVO_BA_XCD_RESERVE=$res timeout 200 python bench.py --steps 60 --no-cpu-baseline --no-secondary --streams $st > $O/bench_r${res}_s$st.json 2> $O/bench_r${res}_s$st.err
  python - "$res" "$st" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r03o/bench_r%s_s%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("reserve", sys.argv[1], "streams", sys.argv[2], round(d["value"]), round(r["frac"], 4), round(r.get("avg_window_ms") or 0, 3), round(r.get("windows_in_flight") or 0, 2), d["secondary"].get("headline_host_us_per_frame"))
except Exception as e:
    print(sys.argv[1:], "unreadable", e)
PY
done; done
