cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8
bash tools/_dbg.sh
