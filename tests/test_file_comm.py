"""FileComm (abacusutils_amd/comm.py): the barrier / scalar all-reduce the collective-free HOD leg of `bench.py --gpus N`
falls back to when no RCCL communicator can be created.  Two and three rank processes on the CPU."""
import multiprocessing as mp
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def _rank(rank, world, key, tmpdir, q, slow_rank=-1):
    os.environ['ABACUS_RDZV_DIR'] = tmpdir
    import time
    from abacusutils_amd.comm import Dist, FileComm

    class SlowReader(FileComm):      # a rank that is late reading its peers' files (a loaded box)
        def _read(self, path, what, t0):
            time.sleep(0.15)
            return super()._read(path, what, t0)

    d = Dist((SlowReader if rank == slow_rank else FileComm)(rank, world, key=key, timeout=30.0))
    d.barrier()
    mx = d.max(float(10 + rank))
    sm = d.sum(float(rank + 1))
    d.barrier()
    q.put((rank, mx, sm, d.comm.info()['transport']))
    d.finish()


@pytest.mark.parametrize('world,slow_rank', [(2, -1), (3, -1), (3, 1), (3, 0), (2, 1)])
def test_file_comm_barrier_max_sum(world, slow_rank, tmp_path):
    """slow_rank >= 0: that rank reads every peer file late - its peers have long finished (and called free()) when it
    reads the last round; nothing it still needs may have been removed (teardown race)"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, world, f'k{world}', str(tmp_path), q, slow_rank)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=60) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for r, (rank, mx, sm, transport) in enumerate(got):
        assert rank == r and mx == 10 + world - 1 and sm == world * (world + 1) / 2
        assert 'file barrier' in transport
    assert not [f for f in os.listdir(tmp_path) if '.file.' in f]   # rank 0 removed what was left after every rank's `done`
