#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ba.py tests/test_gpu_concurrency.py -x -q 2>&1 | tail -2 | tee $O/pytest_uv.log
grep -q passed $O/pytest_uv.log && ! grep -q failed $O/pytest_uv.log || exit 1
timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_k20_nosec.json 2> /dev/null
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
for f in ("bench_k20_nosec", "bench_default"):
    d = json.loads(open("gpurun_out/r03/%s.json" % f).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f, round(d["value"], 1), round(r["frac"], 4), round(r["windows_in_flight"], 2), round(r["avg_window_ms"], 3))
    s = d.get("secondary", {})
    print("   ", {k: (round(v) if isinstance(v, float) else v) for k, v in s.items() if k.endswith("_fps")})
PY
