cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout 600 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2b/prof1 -o bench -- python bench.py --steps 40 --warmup 5 --streams 1 --pipeline 0 --no-cpu-baseline --no-secondary > gpurun_out/r2b/prof1.log 2>&1
tail -3 gpurun_out/r2b/prof1.log | cut -c 1-600
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r2b/prof1/bench_kernel_trace.csv')))
ks=sorted(rows,key=lambda r:int(r['Start_Timestamp']))
t0=int(ks[0]['Start_Timestamp'])
for r in ks[-14:]:
    print(r['Kernel_Name'][:24].ljust(24), r['Queue_Id'], "%10.1f %8.1f"%((int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3), r['Grid_Size_X'])
PY
