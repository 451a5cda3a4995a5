cd $GRAFT_REPO_ROOT
make -s -C oracle
timeout 1200 python -m pytest $@ -m gpu -x -q 2>&1 | tail -30
