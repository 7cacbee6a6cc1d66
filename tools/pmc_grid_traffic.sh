#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the resident solver grid under the headline load (two bounded --pmc passes; the launch-path form of the
# same cut as the fallback when a pass next to the never-ending kernel hangs): bash tools/pmc_grid_traffic.sh <tag>
set -u
TAG=${1:?tag}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
bounded() { local t=$1 log=$2; shift 2; setsid "$@" > $log 2>&1 & local p=$!; ( sleep $t; kill -KILL -- -$p 2>/dev/null ) & local w=$!; wait $p; local rc=$?; kill $w 2>/dev/null; wait $w 2>/dev/null; echo "rc=$rc $(echo "$*" | cut -c1-100)"; return $rc; }
NOX="--no-cpu-baseline --no-secondary --no-parity"
DEF="env MVO_BA_SERVICE=2 python bench.py --steps 10 --warmup 0 $NOX"
ok=1
bounded 80 $OUT/f.log rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_default -o bench -- $DEF || ok=0
[ $ok = 1 ] && { bounded 80 $OUT/w.log rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_default -o bench -- $DEF || ok=0; }
if [ $ok = 1 ]; then
  python tools/pmc_summary.py fetch_write $OUT/pmc_fetch_default/bench_counter_collection.csv $OUT/pmc_write_default/bench_counter_collection.csv $OUT/pmc_fetch_write_size_per_kernel.csv "$DEF" $((32 * 10 * 10))
else
  LP="env MVO_BA_SERVICE=0 python bench.py --steps 10 --warmup 0 $NOX"
  bounded 60 $OUT/lf.log rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_lp -o bench -- $LP
  bounded 60 $OUT/lw.log rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_lp -o bench -- $LP
  python tools/pmc_summary.py fetch_write $OUT/pmc_fetch_lp/bench_counter_collection.csv $OUT/pmc_write_lp/bench_counter_collection.csv $OUT/pmc_launch_path_fetch_write_size.csv "$LP"
fi
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*_counter_collection.csv" -delete; find $OUT -name "*_agent_info.csv" -delete
cat $OUT/*.csv 2>/dev/null | cut -c1-200
