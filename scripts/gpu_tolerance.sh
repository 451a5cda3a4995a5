#!/bin/bash
# float32 transform noise behind the widened tolerances of tests/test_power_gpu.py (scripts/tolerance_probe.py)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/tolerance
make -s -C oracle
timeout -k 10 900 python3 scripts/tolerance_probe.py > gpurun_out/tolerance/probe.jsonl 2> gpurun_out/tolerance/probe.err
tail -8 gpurun_out/tolerance/probe.jsonl | cut -c1-600; tail -3 gpurun_out/tolerance/probe.err
