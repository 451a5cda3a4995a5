#!/bin/bash
# what the GPU box gives the host side: CPUs (affinity, cgroup quota), memory, and how the oracle's OpenMP / pocketfft legs
# scale with the team size (the bench's cpu_baseline and every oracle call of the GPU tests depend on it)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/hostinfo; mkdir -p $O
{
echo "nproc: $(nproc)  affinity: $(python3 -c 'import os; print(len(os.sched_getaffinity(0)))')"
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "cfs_quota: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) / $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null)"
echo "memory.max: $(cat /sys/fs/cgroup/memory.max 2>/dev/null) current: $(cat /sys/fs/cgroup/memory.current 2>/dev/null)"
echo "memory v1: $(cat /sys/fs/cgroup/memory/memory.limit_in_bytes 2>/dev/null)"
grep -E "MemTotal|MemAvailable" /proc/meminfo
lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node\(s\)"
echo "OMP_NUM_THREADS=$OMP_NUM_THREADS"
cat /proc/self/status | grep -i cpus_allowed_list
ulimit -a | grep -E "processes|memory"
} > $O/host.txt 2>&1
cat $O/host.txt
make -s -C oracle
python3 - > $O/threads.txt 2>&1 <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
from oracle import oracle
from abacusutils_amd import synth
import bench_pk
print('cpu_share', bench_pk.cpu_share(), 'cpu_threads', bench_pk.cpu_threads(), 'host_memory_gb', bench_pk.host_memory_gb(), 'omp max', oracle.max_threads())
L, nmesh, n = 2000.0, 512, 20_000_000
pos = np.random.default_rng(300).random((n, 3), dtype=np.float32) * np.float32(L)
kw = dict(kbins=256, mubins=4, k_max=np.pi * nmesh / L, paste='TSC', nmesh=nmesh, compensated=False, interlaced=False, poles=[0, 2, 4], accum64=True)
for t in (8, 16, 32, 64, 128, 256):
    if t > len(os.sched_getaffinity(0)): break
    os.environ['OMP_NUM_THREADS'] = str(t)
    oracle.calc_power(pos, L, nthread=t, **kw)
    t0 = time.perf_counter(); oracle.calc_power(pos, L, nthread=t, **kw); print('calc_power 512^3 2e7', t, 'threads', round(time.perf_counter() - t0, 3), 's', flush=True)
hd, pd, params = synth.synth_hod_inputs(4_000_000, 4_000_000, seed=600)
for t in (8, 16, 32, 64, 128, 256):
    if t > len(os.sched_getaffinity(0)): break
    tmin, tmean, _ = oracle.time_gen_gals(hd, pd, {'LRG': synth.LRG_PARAMS}, params, t, reps=3)
    print('gen_gals 4e6', t, 'threads', round(tmin * 1e3, 2), 'ms', flush=True)
PY
cat $O/threads.txt
