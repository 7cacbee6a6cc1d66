cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2b
for m in 0; do
MVO_DBG=$m timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2b/dbg$m -o bench -- python bench.py --steps 30 --warmup 5 --streams 1 --pipeline 0 --ba-mode none --no-cpu-baseline --no-secondary > gpurun_out/r2b/dbg$m.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/r2b/dbg$m/bench_kernel_stats.csv')))
for r in rows[:6]: print($m, r['Name'][:30], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
done
