cd "${GRAFT_REPO_ROOT:-.}"
SO=monocular-visual-odometry_amd/csrc/libmvo_hip.so
echo "== round 6"; timeout 200 python tools/ba10_probe.py 0 2>&1 | tail -3
cp $SO /tmp/.shipped.so; cp monocular-visual-odometry_amd/csrc/_alt/libmvo_hip_r05.so $SO
echo "== round 5"; timeout 200 python tools/ba10_probe.py 0 2>&1 | tail -3
cp /tmp/.shipped.so $SO
