cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2b
timeout 300 python -m pytest tests/test_gpu_match.py tests/test_gpu_track.py tests/test_golden.py tests/test_properties.py -x -q -m gpu 2>&1 | tail -3
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2b/k32 -o bench -- python bench.py --steps 30 --warmup 5 --streams 1 --pipeline 0 --ba-mode none --no-cpu-baseline --no-secondary > gpurun_out/r2b/k32.log 2>&1
python - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/r2b/k32/bench_kernel_stats.csv')):
    if r['Name'].startswith('k_'): print(r['Name'][:16], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
