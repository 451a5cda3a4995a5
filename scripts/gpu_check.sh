#!/bin/bash
# on the GPU box: bash scripts/gpu_check.sh [tests] [bench] [bench2] [only:<pytest -k expr>]
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/check
mkdir -p "$O"
export TMPDIR=/tmp
for w in "$@"; do
case $w in
tests)
  make -s -C oracle
  # (progress goes to the file as it happens: a run that writes nothing for seven minutes is taken to be hung)
  timeout 2700 python -u -m pytest tests -m gpu -q -p no:cacheprovider > "$O/gpu_tests.log" 2>&1; tail -40 "$O/gpu_tests.log"
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -12 | tee "$O/smoke.log"
  ;;
only:*)
  make -s -C oracle
  timeout 2400 python -u -m pytest tests -m gpu -x -q -p no:cacheprovider -k "${w#only:}" > "$O/gpu_only.log" 2>&1; tail -40 "$O/gpu_only.log"
  ;;
bench)
  timeout 1500 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"; cat "$O/bench_default.json"; tail -3 "$O/bench_default.err"
  ;;
bench2)
  # two ranks on a one-GPU box: RCCL refuses the duplicate device - the line must still be printed, with the error
  timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --hod-timeout 200 --slab-timeout 120 > "$O/bench_gpus2.json" 2> "$O/bench_gpus2.err"; echo "rc=$?"; cut -c1-1500 "$O/bench_gpus2.json"; tail -3 "$O/bench_gpus2.err"
  ;;
esac
done
