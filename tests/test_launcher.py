"""abacusutils_amd/launch.py (the one-process-per-GPU launcher behind `python bench.py --gpus N`) and bench.py's N > 1
orchestration, on the CPU: rank environment, result collection, a crashed rank, a stuck rank, the rendezvous of
abacusutils_amd/comm.py, and the bench line when no GPU exists."""
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
WORKER = [sys.executable, os.path.join(HERE, '_launch_worker.py')]


def test_launch_two_ranks_gloo():
    from abacusutils_amd.launch import launch_ranks
    res = launch_ranks(WORKER, 2, timeout=240, key='unit_test_key')
    assert res['returncodes'] == {0: 0, 1: 0} and not res['timed_out'], res
    assert res['results'][0] == {'sum': 3, 'world': 2, 'key': 'unit_test_key'} and res['results'][1] is None


def test_launch_reports_a_crashed_rank_and_ends_its_peers():
    from abacusutils_amd.launch import failure_summary, launch_ranks
    t0 = time.time()
    res = launch_ranks(WORKER + ['--fail-rank', '1'], 2, timeout=240, grace=3.0)
    assert res['returncodes'][1] == 3 and res['returncodes'][0] != 0      # rank 0 waited for its peer and was ended
    assert time.time() - t0 < 120 and not res['timed_out']
    assert 'rank 1: exit 3: boom from rank 1' in failure_summary(res)


def test_launch_ends_a_stuck_rank_at_the_timeout():
    from abacusutils_amd.launch import failure_summary, launch_ranks
    res = launch_ranks(WORKER + ['--hang-rank', '0'], 2, timeout=8.0)
    assert res['timed_out'] and all(c != 0 for c in res['returncodes'].values())
    assert 'abandoned after' in failure_summary(res)


def test_file_rendezvous(tmp_path):
    """rank 0 publishes atomically, the others poll; a stale file (older than the reader by > 10 min) is ignored"""
    from abacusutils_amd import comm
    path = str(tmp_path / 'rdzv')
    blob = bytes(range(128))
    got = {}
    th = threading.Thread(target=lambda: got.setdefault(1, comm.exchange_id(1, 2, None, path, timeout=30)))
    th.start()
    time.sleep(0.2)
    assert comm.exchange_id(0, 2, lambda: blob, path) == blob
    th.join()
    assert got[1] == blob
    os.utime(path, (time.time() - 3600, time.time() - 3600))
    try:
        comm.exchange_id(1, 2, None, path, timeout=0.3)
        raise AssertionError('a stale id was accepted')
    except TimeoutError:
        pass


def _last_json(stdout):
    return json.loads([ln for ln in stdout.splitlines() if ln.startswith('{')][-1])


def test_bench_gpus_2_fails_loudly_without_a_gpu():
    """`python bench.py --gpus 2` in a container without a GPU: the parent starts two rank processes, both report
    "no HIP device", the line is still printed (value null, error from both ranks) and the exit code is non-zero"""
    from abacusutils_amd import _lib
    if _lib.device_count() > 0:
        import pytest
        pytest.skip('a GPU is present')
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    line = _last_json(r.stdout)
    assert line['n_gpus'] == 2 and line['value'] is None
    assert 'rank 0' in line['error'] and 'rank 1' in line['error'] and line['error'].count('no HIP device') == 2
    assert 'no HIP device' in line['pk_slab']['error']


def test_bench_under_torchrun_fails_loudly_without_a_gpu():
    """the driver's N > 1 command: one orchestrator per rank, each starts its own child; rank 0's prints the line"""
    from abacusutils_amd import _lib
    if _lib.device_count() > 0:
        import pytest
        pytest.skip('a GPU is present')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', '29671', os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--no-slab']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1                      # ONE line, from rank 0's orchestrator
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and 'no HIP device' in line['error']


def test_dist_without_ranks_is_trivial():
    from abacusutils_amd.comm import Dist
    d = Dist(None)
    assert (d.rank, d.world) == (0, 1) and d.max(2.5) == 2.5 and d.sum(2.5) == 2.5
    d.barrier()
    d.finish()
    assert np.isfinite(d.max(1.0))


def test_run_with_deadline():
    """the watchdog around ncclCommInitRank: result and exceptions pass through, a call that never returns becomes TimeoutError"""
    import threading
    import time

    from abacusutils_amd.comm import _run_with_deadline
    seen = []
    _run_with_deadline(lambda: seen.append(1), 5.0, 'x')
    assert seen == [1]
    with pytest.raises(KeyError):
        _run_with_deadline(lambda: {}['missing'], 5.0, 'x')
    gate = threading.Event()
    t0 = time.time()
    with pytest.raises(TimeoutError, match='stuck'):
        _run_with_deadline(gate.wait, 0.3, 'stuck')
    assert time.time() - t0 < 3
    gate.set()


def test_join_timeout_is_not_turned_into_a_file_fallback(monkeypatch, tmp_path):
    """a ncclCommInitRank / first barrier that never returns leaves a thread of this process inside the library;
    `Dist.from_env(allow_file_fallback=True)` (the HOD leg of bench.py) must then END the rank - any other RCCL failure
    still falls back to the file barrier and reports the error"""
    from abacusutils_amd import comm
    monkeypatch.setenv('WORLD_SIZE', '2')
    monkeypatch.setenv('RANK', '0')
    monkeypatch.setenv('ABACUS_RDZV_DIR', str(tmp_path))

    def stuck(cls, **kw):
        raise comm.RcclJoinTimeout('rank 0/2: ncclCommInitRank did not return')

    monkeypatch.setattr(comm.RcclComm, 'from_env', classmethod(stuck))
    with pytest.raises(comm.RcclJoinTimeout):
        comm.Dist.from_env(allow_file_fallback=True, key='t_join')
    assert issubclass(comm.RcclJoinTimeout, TimeoutError)

    def refused(cls, **kw):
        raise RuntimeError('ncclCommInitRank: invalid usage (duplicate device)')

    monkeypatch.setattr(comm.RcclComm, 'from_env', classmethod(refused))
    d = comm.Dist.from_env(allow_file_fallback=True, key='t_join')
    assert isinstance(d.comm, comm.FileComm) and 'duplicate device' in d.rccl_error
    with pytest.raises(RuntimeError):
        comm.Dist.from_env(allow_file_fallback=False, key='t_join')


def test_cpu_share_helpers_follow_the_cgroup_quota(monkeypatch, tmp_path):
    """bench_pk.cpu_share / cpu_threads / host_memory_gb and oracle.max_threads: the team sizes of the CPU baselines and of the
    tests' oracle calls follow the cgroup's CPU quota where there is one (the GPU box of round 6: 256 logical CPUs, cpu.max = 16)"""
    import builtins
    import os

    import bench_pk
    from oracle import oracle
    files = {'/sys/fs/cgroup/cpu.max': '1600000 100000\n', '/sys/fs/cgroup/memory.max': str(64 << 30) + '\n',
             '/sys/fs/cgroup/memory.current': str(4 << 30) + '\n'}
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if str(path) in files:
            p = tmp_path / str(path).strip('/').replace('/', '_')
            p.write_text(files[str(path)])
            return real_open(p, *a, **k)
        return real_open(path, *a, **k)

    monkeypatch.setattr(builtins, 'open', fake_open)
    monkeypatch.setattr(os, 'sched_getaffinity', lambda pid: set(range(256)))
    assert bench_pk.cpu_share() == (256, 16.0)
    assert bench_pk.cpu_threads() == [8, 16, 32]
    assert bench_pk.host_memory_gb() <= 60 * 2**30 / 1e9 + 1e-6          # capped by what the cgroup has left (60 GiB)
    assert oracle.cpu_quota() == 16.0
    assert oracle.max_threads() <= 32
    files['/sys/fs/cgroup/cpu.max'] = 'max 100000\n'
    assert bench_pk.cpu_share() == (256, None) and bench_pk.cpu_threads() == [64, 128, 256]
    assert oracle.cpu_quota() is None
    assert bench_pk.cpu_pk_host_gb(2048, 100_000_000) > 100 > bench_pk.cpu_pk_host_gb(1024, 100_000_000)
