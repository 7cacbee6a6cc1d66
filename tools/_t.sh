cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_ba.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -8
timeout 200 python tools/ba_probe.py 0 2>&1 | grep -v amdgpu | head -4
