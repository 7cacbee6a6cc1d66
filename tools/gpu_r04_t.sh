#!/bin/bash
# round 4, GPU call T: k_pnp_hypotheses held to two waves per SIMD -- parity and the tracking-row bench
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04t
mkdir -p $O
MVO_PNP_OCC=2 timeout 300 python -m pytest tests/test_gpu_track.py tests/test_gpu_host_adapter.py -x -q 2>&1 | tail -2
show() { python - $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"] or {}
    print(sys.argv[2], "value", round(d["value"], 1), "host", d["secondary"].get("headline_host_us_per_frame"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
B="python bench.py --gpus 1 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --track"
for rep in 1 2; do
  MVO_PNP_OCC=1 timeout 300 $B > $O/occ1_$rep.json 2> $O/occ1_$rep.err; show $O/occ1_$rep.json track_occ1_$rep
  MVO_PNP_OCC=2 timeout 300 $B > $O/occ2_$rep.json 2> $O/occ2_$rep.err; show $O/occ2_$rep.json track_occ2_$rep
done
MVO_PNP_OCC=2 timeout 300 $B --streams 32 > $O/occ2_s32.json 2> $O/occ2_s32.err; show $O/occ2_s32.json track_occ2_streams32
python tools/track_probe.py 2>&1 | tail -6
MVO_PNP_OCC=2 python tools/track_probe.py 2>&1 | tail -6
