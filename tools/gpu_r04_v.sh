#!/bin/bash
# round 4, GPU call V: MVO_BA_MODE_SHARED (throughput cut, launch path only) for loops whose frames wait for the tracking rows
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04v
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ba.py tests/test_gpu_concurrency.py tests/test_gpu_track.py -x -q 2>&1 | tail -1
show() { python - $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"] or {}
    print(sys.argv[2], "value", round(d["value"], 1), r.get("kernel"), "host", d["secondary"].get("headline_host_us_per_frame"), {k: round(v) for k, v in d["secondary"].items() if k.endswith("_fps")})
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --track > $O/track.json 2> $O/track.err; show $O/track.json track_shared_mode_24
timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --track --streams 32 > $O/track32.json 2> $O/track32.err; show $O/track32.json track_shared_mode_32
timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --track --streams 16 > $O/track16.json 2> $O/track16.err; show $O/track16.json track_shared_mode_16
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/headline.json 2> $O/headline.err; show $O/headline.json headline_with_secondary
