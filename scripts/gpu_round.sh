#!/bin/bash
# round check on the GPU box: all gpu tests, smoke, default bench, rocprofv3 kernel stats and PMC traffic passes.
# usage: bash scripts/gpu_round.sh [tests|bench|prof|pmc ...]   (default: all)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round
mkdir -p "$O"
export TMPDIR=/tmp
what=${*:-tests bench prof pmc}
for w in $what; do
case $w in
tests)
  make -s -C oracle
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=40 > "$O/gpu_tests.log" 2>&1; tail -15 "$O/gpu_tests.log"
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -12 | tee "$O/smoke.log"
  ;;
bench)
  timeout 1500 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"; cat "$O/bench_default.json"; tail -3 "$O/bench_default.err"
  ;;
prof)
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_hod" -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu --no-pk --no-hod-extra > "$O/prof_hod.log" 2>&1
  for NM in 1024 1536 2048; do
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_pk$NM" -- python3 "$R/bench.py" --workload pk --nmesh $NM --steps 3 --warmup 1 --no-cpu > "$O/prof_pk$NM.log" 2>&1
  done
  cd "$R"
  for d in prof_hod prof_pk1024 prof_pk1536 prof_pk2048; do
    f=$(find "$O/$d" -name "*kernel_stats.csv" | head -1); echo "== $d"; cut -c1-160 "$f" | head -14
    find "$O/$d" -name "*kernel_trace.csv" -delete; find "$O/$d" -name "*.db" -delete
  done
  ;;
pmc)
  cd /tmp
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$O/pmc_hod_$C" -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu --no-pk --no-hod-extra > "$O/pmc_hod_$C.log" 2>&1
    for NM in 1024 2048; do
      timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$O/pmc_pk${NM}_$C" -- python3 "$R/bench.py" --workload pk --nmesh $NM --steps 2 --warmup 1 --no-cpu > "$O/pmc_pk${NM}_$C.log" 2>&1
    done
  done
  cd "$R"
  python3 scripts/summarize_pmc.py "$O" > "$O/pmc_summary.json"; cat "$O/pmc_summary.json" | head -60
  find "$O" -path "*pmc_*" \( -name "*kernel_trace.csv" -o -name "*counter_collection.csv" -o -name "*.db" \) -delete
  ;;
esac
done
