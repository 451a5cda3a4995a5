cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest "$@" -m gpu -x -q 2>&1 | tail -30
