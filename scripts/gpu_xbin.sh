#!/bin/bash
# fused last pass (xbin.hip) / FFT passes: parity tests, then the 2048^3 step per form / ablation
# usage: gpu_xbin.sh [tests|alltests|notests] [mode ...]   (modes as in gpu_pk_ablate.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/xbin
mkdir -p "$O"
make -s -C oracle
what=${1:-tests}; shift
case $what in
tests) timeout 900 python -m pytest tests/test_power_gpu.py -m gpu -x -q -k "fused_last_pass or c3_full or analytic_known or full_size_2048" 2>&1 | tail -15 | tee "$O/tests.log" || exit 1 ;;
alltests) timeout 1100 python -m pytest tests/test_power_gpu.py tests/test_comm_gpu.py -m gpu -x -q 2>&1 | tail -15 | tee "$O/tests.log" || exit 1 ;;
esac
bash scripts/gpu_pk_ablate.sh ${@:-runs pairs:pk_xbin_pairs=1 gen1:pk_xbin_gen=1 runs_nobin:dbg=2 runs_noatom:dbg=4 runs_nolds:dbg=8} 2>&1 | tee "$O/ablate.log"
