#!/bin/bash
# round 4, GPU call S: ONE rank as it would run on an 8-GPU node -- confined to an eighth of the host's CPUs (LOCAL_WORLD_SIZE=8
# faked on the 1-GPU box), with the runtime's wait policies
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04s
mkdir -p $O
show() { python - $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"] or {}
    print(sys.argv[2], "value", round(d["value"], 1), "in flight", round(r.get("windows_in_flight", 0), 2), "host", d["secondary"].get("headline_host_us_per_frame"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline"
timeout 300 $B > $O/all_cpus.json 2> $O/all_cpus.err; show $O/all_cpus.json all_256_cpus
LOCAL_WORLD_SIZE=8 LOCAL_RANK=0 timeout 300 $B > $O/slice_yield.json 2> $O/slice_yield.err; show $O/slice_yield.json slice_of_32_cpus_yield; grep -i "wait policy" $O/slice_yield.err
LOCAL_WORLD_SIZE=8 LOCAL_RANK=0 MVO_BENCH_WAIT_POLICY=auto timeout 300 $B > $O/slice_auto.json 2> $O/slice_auto.err; show $O/slice_auto.json slice_of_32_cpus_auto
LOCAL_WORLD_SIZE=8 LOCAL_RANK=0 MVO_BENCH_WAIT_POLICY=block timeout 300 $B > $O/slice_block.json 2> $O/slice_block.err; show $O/slice_block.json slice_of_32_cpus_block
LOCAL_WORLD_SIZE=8 LOCAL_RANK=0 MVO_BENCH_WAIT_POLICY=auto timeout 300 $B --streams 24 > $O/slice_auto24.json 2> $O/slice_auto24.err; show $O/slice_auto24.json slice_of_32_cpus_auto_streams24
