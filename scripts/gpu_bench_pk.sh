cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -s -C oracle
timeout 900 python bench.py --workload pk --nmesh 1024 --steps 5 --warmup 1 --no-cpu > gpurun_out/bench_pk1024.json 2> gpurun_out/bench_pk1024.err; cat gpurun_out/bench_pk1024.json; tail -3 gpurun_out/bench_pk1024.err
timeout 900 python bench.py --workload pk --nmesh 2048 --npk 100000000 --steps 3 --warmup 1 --no-cpu > gpurun_out/bench_pk2048.json 2> gpurun_out/bench_pk2048.err; cat gpurun_out/bench_pk2048.json; tail -3 gpurun_out/bench_pk2048.err
