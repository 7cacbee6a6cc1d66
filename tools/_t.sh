cd $GRAFT_REPO_ROOT
timeout 200 python bench.py --steps 50 --warmup 5 --streams 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['secondary'])"
