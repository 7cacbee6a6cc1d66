#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03z
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_concurrency.py tests/test_gpu_ba.py -x -q 2>&1 | tail -3
pr() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], round(d["value"]), r.get("kernel"), round(r.get("frac"), 4), r.get("launches"), round(r.get("windows_in_flight") or 0, 2), round(r.get("avg_window_ms") or r.get("avg_launch_ms") or 0, 3), d.get("secondary", {}).get("headline_host_us_per_frame"))
    s = d.get("secondary", {})
    print("   ", {k: (round(v) if isinstance(v, float) else v) for k, v in s.items() if k.endswith("_fps")})
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/k20a.json 2> $O/k20a.err; pr $O/k20a.json
timeout 200 python bench.py --track --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/track.json 2> $O/track.err; pr $O/track.json
timeout 200 python bench.py --track --steps 60 --warmup 10 --no-cpu-baseline --no-secondary > $O/track60.json 2> $O/track60.err; pr $O/track60.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/full.json 2> $O/full.err; pr $O/full.json
