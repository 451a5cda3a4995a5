# full GPU check: build check, all gpu tests, smoke, default bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -s -C oracle
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -12
timeout 1200 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cat gpurun_out/bench_default.json; tail -3 gpurun_out/bench_default.err
