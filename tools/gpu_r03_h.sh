#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03h
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_match.py tests/test_gpu_host_adapter.py -x -q > $O/pytest_match.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_match.log
timeout 300 python bench.py --steps 10 --warmup 3 --streams 1 --pipeline 0 --no-cpu-baseline --no-secondary > $O/bench_s1.json 2> $O/bench_s1.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03h/bench_s1.json").read().strip().splitlines()[-1]); print(d["kernels"])
PY
timeout 300 python bench.py --steps 60 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; tail -c 1500 $O/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03h/bench_default.json").read().strip().splitlines()[-1]); print(round(d["value"]), d["secondary"], d["kernels"])
PY
