#!/bin/bash
# round 4, GPU call F: same-box A/B of three builds (round-3 final, call-B build, working tree).  Needs the two older trees built
# next to this one first: git worktree add _wt/r03 bc83609; git worktree add _wt/callb 5d79eb4; (cd _wt/<tree> && python -c "import __graft_entry__ as g; g.build()")
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r04f
mkdir -p $O
uptime
show() { python - $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"] or {}
    print(sys.argv[2], "value", round(d["value"], 1), "avg_window_ms", round(r.get("avg_window_ms", 0), 3), "in flight", round(r.get("windows_in_flight", 0), 2), "host", d["secondary"].get("headline_host_us_per_frame"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for rep in 1 2; do
  (cd $ROOT/_wt/r03 && timeout 300 python bench.py --gpus 1 --steps 200 --warmup 50 --no-secondary --no-cpu-baseline > $O/r03_$rep.json 2> $O/r03_$rep.err); show $O/r03_$rep.json r03_$rep
  (cd $ROOT/_wt/callb && timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/callb_$rep.json 2> $O/callb_$rep.err); show $O/callb_$rep.json callb_$rep
  (cd $ROOT && timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/work_$rep.json 2> $O/work_$rep.err); show $O/work_$rep.json work_$rep
done
