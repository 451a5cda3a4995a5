"""FileComm (abacusutils_amd/comm.py): the barrier / scalar all-reduce the collective-free HOD leg of `bench.py --gpus N`
falls back to when no RCCL communicator can be created.  Two and three rank processes on the CPU."""
import multiprocessing as mp
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def _rank(rank, world, key, tmpdir, q):
    os.environ['ABACUS_RDZV_DIR'] = tmpdir
    from abacusutils_amd.comm import Dist, FileComm
    d = Dist(FileComm(rank, world, key=key, timeout=30.0))
    d.barrier()
    mx = d.max(float(10 + rank))
    sm = d.sum(float(rank + 1))
    d.barrier()
    q.put((rank, mx, sm, d.comm.info()['transport']))
    d.finish()


@pytest.mark.parametrize('world', [2, 3])
def test_file_comm_barrier_max_sum(world, tmp_path):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, world, f'k{world}', str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=60) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for r, (rank, mx, sm, transport) in enumerate(got):
        assert rank == r and mx == 10 + world - 1 and sm == world * (world + 1) / 2
        assert 'file barrier' in transport
    assert not [f for f in os.listdir(tmp_path) if '.file.' in f]   # every rank removed its own files
