#!/bin/bash
# (needs _wt/callb like gpu_r04_f.sh)
# round 4, GPU call H: host stage times of the extraction under the 24-shard load, working tree vs call-B build
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r04h
mkdir -p $O
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline"
for t in work callb; do
  d=$ROOT; [ $t = callb ] && d=$ROOT/_wt/callb
  (cd $d && MVO_HOST_TIMING=1 MVO_BA_SERVICE=2 timeout 300 $B > $O/$t.json 2> $O/$t.err)
  echo "== $t"; python - $O/$t.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", round(d["value"], 1), d["secondary"].get("headline_host_us_per_frame"))
PY
  grep "mvo host" $O/$t.err | awk '{k=$0; sub(/[0-9.]+/,"",k)} {print}' | sort | uniq -c | sort -rn | head -0
  python - $O/$t.err <<'PY'
import re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for line in open(sys.argv[1]):
    m = re.match(r"\[mvo (host|ba_stage) us/(frame|window)\] (.*)", line)
    if not m: continue
    toks = m.group(3).split()
    key = m.group(1) + ":" + toks[0]
    for i in range(0, len(toks) - 1, 2):
        acc[key][toks[i]].append(float(toks[i + 1]))
for key, d in acc.items():
    print(key, {k: round(sum(v) / len(v), 1) for k, v in d.items()}, "n", len(next(iter(d.values()))))
PY
done
