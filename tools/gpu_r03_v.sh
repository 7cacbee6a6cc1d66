#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03w
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_orb.py tests/test_gpu_concurrency.py -x -q 2>&1 | tail -5 > $O/pytest.log; cat $O/pytest.log
pr() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1], round(d["value"]), r.get("kernel"), round(r.get("frac"), 4), r.get("launches"), r.get("windows_in_flight"), r.get("windows_per_launch"), d.get("secondary", {}).get("headline_host_us_per_frame"), d["kernel_ms_per_frame"])
    s = d.get("secondary", {})
    print({k: (round(v) if isinstance(v, float) else v) for k, v in s.items() if k.endswith("_fps")}, s.get("single_sequence_host_us_per_frame"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for k in 20 60 20; do timeout 300 python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_k$k.json 2> $O/bench_k$k.err; pr $O/bench_k$k.json; done
MVO_HOST_TIMING=1 timeout 120 python bench.py --streams 1 --pipeline 0 --ba-mode none --steps 420 --warmup 10 --no-cpu-baseline --no-secondary > $O/bench_host1_noba.json 2> $O/bench_host1_noba.err; grep "mvo host" $O/bench_host1_noba.err | tail -2
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err; pr $O/bench_full.json
