#!/bin/bash
# (f)4 oracle-chain test, host-stage timing of one shard, tracking rows: where does --track lose its time
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03t
mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_run_vo.py -x -q -s 2>&1 | tail -15 > $O/pytest_run_vo.log; cat $O/pytest_run_vo.log
MVO_HOST_TIMING=1 timeout 120 python bench.py --streams 1 --pipeline 0 --steps 420 --warmup 10 --no-cpu-baseline --no-secondary > $O/bench_host1.json 2> $O/bench_host1.err; grep "mvo host" $O/bench_host1.err | tail -4
MVO_HOST_TIMING=1 timeout 120 python bench.py --streams 1 --pipeline 0 --ba-mode none --steps 420 --warmup 10 --no-cpu-baseline --no-secondary > $O/bench_host1_noba.json 2> $O/bench_host1_noba.err; grep "mvo host" $O/bench_host1_noba.err | tail -4; tail -c 400 $O/bench_host1_noba.err
pr() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["value"]), d["roofline"].get("windows_in_flight"), d.get("secondary", {}).get("headline_host_us_per_frame"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for v in "svc:X=1" "nosvc:MVO_BA_SERVICE=0"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs timeout 200 python bench.py --track --steps 20 --no-cpu-baseline --no-secondary > $O/track_$name.json 2> $O/track_$name.err; pr $O/track_$name.json
done
timeout 200 python bench.py --track --ba-cut latency --steps 20 --no-cpu-baseline --no-secondary > $O/track_lat.json 2> $O/track_lat.err; pr $O/track_lat.json
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_track1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --track --steps 40 --warmup 4 --streams 1 --pipeline 0 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$O/prof_track1.log 2>&1
cd $GRAFT_REPO_ROOT
cut -c1-140 $O/prof_track1/bench_kernel_stats.csv | head -16
