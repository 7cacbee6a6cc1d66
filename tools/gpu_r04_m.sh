#!/bin/bash
# round 4, GPU call M: rocprofv3 kernel stats of the loaded default bench (which extraction kernel holds the free CUs?)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r04m
mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); echo $f; cut -c1-200 $f | head -12
tail -c 300 $O/prof.log
