#!/bin/bash
# round 4, GPU call N: occupancy of the extraction kernels on the free CUs (k_pyramid's pool sized to need, k_brief one wave per workgroup)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r04n
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_orb.py tests/test_gpu_concurrency.py tests/test_gpu_run_vo.py -x -q 2>&1 | tail -2
show() { python - $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"] or {}
    print(sys.argv[2], "value", round(d["value"], 1), "avg_window_ms", round(r.get("avg_window_ms", 0), 3), "in flight", round(r.get("windows_in_flight", 0), 2), "host", d["secondary"].get("headline_host_us_per_frame"), d.get("kernels") and {k: v["avg_launch_us"] for k, v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline"
for rep in 1 2; do
  MVO_PYR_FULL_POOL=1 MVO_BRIEF_WAVES=4 timeout 300 $B > $O/old_$rep.json 2> $O/old_$rep.err; show $O/old_$rep.json round3_form_$rep
  timeout 300 $B > $O/new_$rep.json 2> $O/new_$rep.err; show $O/new_$rep.json new_$rep
done
MVO_BRIEF_WAVES=4 timeout 300 $B > $O/pyr_only.json 2> $O/pyr_only.err; show $O/pyr_only.json pyramid_pool_only
MVO_PYR_FULL_POOL=1 timeout 300 $B > $O/brief_only.json 2> $O/brief_only.err; show $O/brief_only.json brief_waves_only
MVO_EXTRACT_CONCURRENCY=12 timeout 300 $B > $O/new_cap12.json 2> $O/new_cap12.err; show $O/new_cap12.json new_cap12
MVO_EXTRACT_CONCURRENCY=6 timeout 300 $B > $O/new_cap6.json 2> $O/new_cap6.err; show $O/new_cap6.json new_cap6
