#!/bin/bash
# Per-phase cycles of the window solve (tools/ba_probe.py, instrumented kernel) for the shipped library and for other builds of it:
#   bash tools/ba_phase_ab.sh <tag> <wgs list, e.g. 14,28> [name=<path to a libmvo_hip.so> ...]
# Results: gpurun_out/<tag>/<name>.txt.  (The shipped library is put back afterwards.)
cd "${GRAFT_REPO_ROOT:-.}"
TAG=${1:?tag}; WGS=${2:-14,28}; shift 2
O=gpurun_out/$TAG; mkdir -p $O
SO=monocular-visual-odometry_amd/csrc/libmvo_hip.so
timeout 300 python tools/ba_probe.py $WGS 2 > $O/shipped.txt 2>&1
cp $SO $O/.shipped.so
for v in "$@"; do
  name=${v%%=*}; lib=${v#*=}
  cp "$lib" $SO
  timeout 300 python tools/ba_probe.py $WGS 2 > $O/$name.txt 2>&1
done
cp $O/.shipped.so $SO; rm -f $O/.shipped.so
grep -h -E "^full|instrumented|timeline" $O/*.txt | cut -c1-700
