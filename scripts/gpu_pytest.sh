# usage (on the GPU box): bash scripts/gpu_pytest.sh tests/test_x.py [...]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -s -C oracle
timeout 1500 python -m pytest "$@" -m gpu -x -q 2>&1 | tail -40
