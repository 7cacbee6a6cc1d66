cd "${GRAFT_REPO_ROOT:-.}"
for w in 0 32 40 48 56 64; do
  echo "== MVO_BA_WGS=$w"
  MVO_BA_WGS=$w timeout 200 python bench.py --streams 1 --steps 20 --warmup 3 --frames 20 --no-cpu-baseline --no-secondary --no-parity 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('fps',round(d['value'],1),'launch_ms',round(r.get('avg_launch_ms',0),3),'wgs',r.get('workgroups_per_window'))"
done
