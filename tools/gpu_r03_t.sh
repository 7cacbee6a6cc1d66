#!/bin/bash
# ordered tile rows in k_fast_harris, load-driven service, (f)4 chain: parity + timing
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03u
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_orb.py tests/test_gpu_run_vo.py tests/test_gpu_concurrency.py tests/test_gpu_host_adapter.py -x -q 2>&1 | tail -25 > $O/pytest.log; cat $O/pytest.log
MVO_HOST_TIMING=1 timeout 120 python bench.py --streams 1 --pipeline 0 --ba-mode none --steps 420 --warmup 10 --no-cpu-baseline --no-secondary > $O/bench_host1_noba.json 2> $O/bench_host1_noba.err; grep "mvo host" $O/bench_host1_noba.err | tail -2
pr() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1], round(d["value"]), r.get("kernel"), r.get("frac"), r.get("windows_in_flight"), r.get("windows_per_launch"), d.get("secondary", {}).get("headline_host_us_per_frame"))
    s = d.get("secondary", {})
    print({k: (round(v) if isinstance(v, float) else v) for k, v in s.items() if k.endswith("_fps")})
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_k20.json 2> $O/bench_k20.err; pr $O/bench_k20.json
timeout 200 python bench.py --track --steps 20 --no-cpu-baseline --no-secondary > $O/track.json 2> $O/track.err; pr $O/track.json
timeout 200 python bench.py --steps 60 --no-cpu-baseline --no-secondary > $O/bench_k60.json 2> $O/bench_k60.err; pr $O/bench_k60.json
