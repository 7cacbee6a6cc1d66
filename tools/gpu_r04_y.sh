#!/bin/bash
# round 4, GPU call Y: grouped Schur exchange of windows of more than one XCD (config 4: BA10 on 2 x 28 workgroups)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04y
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ba.py -x -q 2>&1 | tail -1
MVO_BA_GROUPS=0 python tools/ba10_probe.py 0 2>&1 | tail -2
python tools/ba10_probe.py 0 2>&1 | tail -2
show() { python - $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"] or {}
    print(sys.argv[2], "value", round(d["value"], 1), r.get("kernel"), "avg_launch_ms", round(r.get("avg_launch_ms", 0), 2), "windows/launch", round(r.get("windows_per_launch", 0), 2), "frac", r.get("frac"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
C4="python bench.py --width 1242 --height 375 --max-kp 4000 --ba-poses 10 --ba-points 4000 --streams 8 --steps 6 --warmup 1 --no-cpu-baseline --no-secondary"
MVO_BA_GROUPS=0 timeout 300 $C4 > $O/c4_flat.json 2> $O/c4_flat.err; show $O/c4_flat.json config4_flat
timeout 300 $C4 > $O/c4.json 2> $O/c4.err; show $O/c4.json config4_grouped
C4="python bench.py --width 1242 --height 375 --max-kp 4000 --ba-poses 10 --ba-points 4000 --streams 8 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_config4 -o bench -- $C4 > $O/pmc_fetch_config4.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_config4 -o bench -- $C4 > $O/pmc_write_config4.log 2>&1
python tools/pmc_summary.py fetch_write $(find $O/pmc_fetch_config4 -name "*counter_collection.csv" | head -1) $(find $O/pmc_write_config4 -name "*counter_collection.csv" | head -1) $O/config4_pmc_fetch_write_size_per_kernel.csv "$C4"
cat $O/config4_pmc_fetch_write_size_per_kernel.csv | cut -c1-200
rm -rf $O/pmc_fetch_config4 $O/pmc_write_config4
