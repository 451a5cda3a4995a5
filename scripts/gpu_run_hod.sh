set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -s -C oracle
timeout 900 python -m pytest tests/test_hod_gpu.py -m gpu -x -q 2>&1 | tail -25
timeout 600 python bench.py --no-pk --steps 20 --warmup 3 > gpurun_out/bench_hod.json 2> gpurun_out/bench_hod.err
cat gpurun_out/bench_hod.json; tail -5 gpurun_out/bench_hod.err
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_hod -- python3 $GRAFT_REPO_ROOT/bench.py --no-pk --no-cpu --steps 20 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_hod.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof_hod -name "*stats*" | head; 
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_hod -name "*kernel_stats.csv" | head -1); head -12 "$f"
