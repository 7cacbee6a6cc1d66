#!/bin/bash
# One parametrised A/B runner for the GPU box (replaces the one-off tools/gpu_r03_*.sh / gpu_r04_*.sh of earlier rounds, whose
# variants were all of this shape: an optional test subset, then the driver's bench command under different environments).
#   bash tools/gpu_ab.sh <tag> [-t "<pytest files / -k expr>"] [-b "<extra bench.py args>"] [-n <repeats>] label[:ENV=val[,ENV=val...]] ...
# e.g. the round-4 calls:  gpu_ab.sh r04n round3_form:MVO_PYR_FULL_POOL=1,MVO_BRIEF_WAVES=4 new cap12:MVO_EXTRACT_CONCURRENCY=12
#                          gpu_ab.sh r04p -t tests/test_gpu_match.py s256:MVO_MATCH_SLICE=256 s128_2x14:MVO_MATCH_SLICE=128,MVO_BA_XCD_RESERVE=4
# A variant may name another build of the library: LIB=<path to a libmvo_hip.so built beforehand> -- it is copied over
# csrc/libmvo_hip.so for that run (the shipped one is put back afterwards).
# Every variant's JSON line goes to gpurun_out/<tag>/<label>[_<rep>].json; one summary line per run on stdout.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${1:?tag}; shift
O=$PWD/gpurun_out/$TAG
mkdir -p "$O"
TESTS=""; EXTRA=""; REPS=1
while getopts "t:b:n:" opt; do
  case $opt in t) TESTS=$OPTARG;; b) EXTRA=$OPTARG;; n) REPS=$OPTARG;; *) exit 2;; esac
done
shift $((OPTIND - 1))
if [ -n "$TESTS" ]; then timeout 600 python -m pytest $TESTS -x -q 2>&1 | tail -3; fi
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    sec = d.get("secondary") or {}
    print(sys.argv[2], "value", round(d["value"], 1), "window_ms", round(r.get("avg_window_ms", r.get("avg_launch_ms", 0)), 3), "in_flight", round(r.get("windows_in_flight", 0), 2),
          "frac", round(r.get("frac", 0), 4), "parity", d.get("parity") and {k: d["parity"].get(k) for k in ("ba", "orb", "match")},
          "host", sec.get("headline_host_us_per_frame"), "single", sec.get("single_sequence_fps") and round(sec["single_sequence_fps"], 1),
          "kernels", d.get("kernels") and {k: v["avg_launch_us"] for k, v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline $EXTRA"
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    label=${v%%:*}; envs=""
    [ "$v" != "$label" ] && envs=$(echo "${v#*:}" | tr ',' ' ')
    f=$O/${label}_$rep
    SO=monocular-visual-odometry_amd/csrc/libmvo_hip.so; alt=""
    for kv in $envs; do case $kv in LIB=*) alt=${kv#LIB=};; esac; done
    if [ -n "$alt" ]; then cp $SO $O/.shipped.so && cp "$alt" $SO; fi
    env $envs timeout 400 $B > $f.json 2> $f.err
    if [ -n "$alt" ]; then cp $O/.shipped.so $SO && rm -f $O/.shipped.so; fi
    show $f.json "${label}_$rep"
  done
done
