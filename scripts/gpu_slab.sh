#!/bin/bash
# slab P(k) GPU tests (ranks share the one GPU; gloo staging)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_slab_power.py -x -q -m gpu 2>&1 | tail -40 | tee gpurun_out/slab_tests.log
