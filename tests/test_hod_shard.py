"""sharded HOD (abacusutils_amd/hod/shard.py): the merged catalogue of W ranks is bit-identical to the single-process
one.  CPU: gloo world_size 2 with the oracle as the per-shard populate; GPU: the HIP path, ranks sharing the GPU."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

from abacusutils_amd import synth
from abacusutils_amd.hod import shard
from oracle import oracle

HERE = os.path.dirname(os.path.abspath(__file__))


def run_ranks(tmp_path, world, backend, port):
    out = str(tmp_path / f'hod_{backend}_{world}')
    worker = os.path.join(HERE, '_hod_shard_worker.py')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr',
           '127.0.0.1', '--master-port', str(port), worker, '--backend', backend, '--out', out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS='2'))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return [pickle.load(open(f'{out}.rank{k}.pkl', 'rb')) for k in range(world)]


def single():
    hd, pd, params = synth.synth_hod_inputs(60000, 90000, seed=77)
    tracers = {'LRG': dict(synth.LRG_PARAMS), 'ELG': dict(synth.ELG_PARAMS), 'QSO': dict(synth.QSO_PARAMS)}
    tracers['ELG'].update(conf_c=0.4, conf_a=0.3)
    return oracle.gen_gal_cat(hd, pd, tracers, params, Nthread=2, rsd=True)


def check(res, ref):
    for cat, counts in res:
        for tr in ref:
            assert cat[tr]['Ncent'] == ref[tr]['Ncent']
            assert counts[tr] == (ref[tr]['Ncent'], len(ref[tr]['x']) - ref[tr]['Ncent'])
            assert len(ref[tr]['x']) > 50
            for k in ('x', 'y', 'z', 'vx', 'vy', 'vz', 'mass', 'id'):
                np.testing.assert_array_equal(cat[tr][k], ref[tr][k], err_msg=f'{tr}.{k}')


def test_shard_catalog_partitions_rows():
    hd, pd, _ = synth.synth_hod_inputs(5000, 20000, seed=3)
    for world in (1, 2, 3, 8):
        for balance in ('particles', 'halos'):
            parts = [shard.shard_catalog(hd, pd, r, world, balance) for r in range(world)]
            assert sum(len(h['hmass']) for h, _ in parts) == 5000
            assert sum(len(p['phmass']) for _, p in parts) == 20000
            np.testing.assert_array_equal(np.concatenate([h['hid'] for h, _ in parts]), hd['hid'])
            np.testing.assert_array_equal(np.concatenate([p['prandoms'] for _, p in parts]), pd['prandoms'])
            for h, p in parts:
                if len(p['pinds']):
                    assert p['pinds'].min() >= 0 and p['pinds'].max() < len(h['hmass'])
                    np.testing.assert_array_equal(h['hid'][p['pinds']], p['phid'])
            if balance == 'particles':
                assert max(len(p['phmass']) for _, p in parts) <= 20000 // world + 1 + np.bincount(pd['pinds']).max()
    # unordered pinds: mask selection
    perm = np.random.default_rng(0).permutation(20000)
    pd2 = {k: v[perm] for k, v in pd.items()}
    parts = [shard.shard_catalog(hd, pd2, r, 2) for r in range(2)]
    assert sum(len(p['phmass']) for _, p in parts) == 20000
    for h, p in parts:
        np.testing.assert_array_equal(h['hid'][p['pinds']], p['phid'])
    # no pinds: even split of both tables
    pd3 = {k: v for k, v in pd.items() if k != 'pinds'}
    parts = [shard.shard_catalog(hd, pd3, r, 4) for r in range(4)]
    assert [len(p['phmass']) for _, p in parts] == [5000] * 4


def test_sharded_hod_gloo_cpu(tmp_path):
    check(run_ranks(tmp_path, 2, 'oracle', 29651), single())


@pytest.mark.gpu
@pytest.mark.parametrize('world', [2, 3])
def test_sharded_hod_hip(tmp_path, world):
    check(run_ranks(tmp_path, world, 'hip', 29660 + world), single())


@pytest.mark.gpu
def test_sharded_hod_eight_ranks_as_threads():
    """eight shards on the one GPU (threads of one process, tests/thread_comm.py): the merged catalogue equals the
    single-process one bit for bit, the all-reduced counts the totals"""
    from thread_comm import run_ranks as run_threads

    from abacusutils_amd import synth
    from abacusutils_amd.hod import shard
    hd, pd, params = synth.synth_hod_inputs(60000, 90000, seed=77)
    tracers = {'LRG': dict(synth.LRG_PARAMS), 'ELG': dict(synth.ELG_PARAMS), 'QSO': dict(synth.QSO_PARAMS)}
    tracers['ELG'].update(conf_c=0.4, conf_a=0.3)

    def rank_fn(tc):
        comm = shard.HodComm(tc)
        cat = shard.run_hod_sharded(hd, pd, tracers, params, comm=comm, rsd=True)
        local = shard.run_hod_sharded(hd, pd, tracers, params, comm=comm, rsd=True, gather=False)
        return cat, comm.all_reduce_counts({t: (c['Ncent'], len(c['x']) - c['Ncent']) for t, c in local.items()})

    check(run_threads(8, rank_fn), single())
