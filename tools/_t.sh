cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4
timeout 120 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), {k: d['roofline'][k] for k in ('avg_launch_ms','launches','windows_per_launch','launch_thread_ms','timed_region_ms')})"
