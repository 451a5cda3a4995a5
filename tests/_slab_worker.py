"""worker for test_slab_power: one rank of the slab estimator.  `--backend numpy` = CPU stand-in device side
(tests/slab_numpy_backend.py) with gloo; `--backend hip` = the HIP entry points (ranks may share one GPU) with
gloo host staging, or `--rccl`: the product transport (abacusutils_amd.comm.RcclComm, RCCL through the C ABI, no torch in
the process) with the particles routed on the device."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--backend', default='numpy')
    ap.add_argument('--rccl', action='store_true')
    ap.add_argument('--force-collectives', action='store_true')
    ap.add_argument('--nmesh', type=int, default=64)
    ap.add_argument('--n', type=int, default=20000)
    ap.add_argument('--interlaced', type=int, default=1)
    ap.add_argument('--compensated', type=int, default=1)
    ap.add_argument('--cross', type=int, default=0)
    ap.add_argument('--kbins', type=int, default=16)
    ap.add_argument('--option', action='append', default=[], help='library option name=value (abacus_set_option)')
    ap.add_argument('--out', required=True)
    a = ap.parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    from abacusutils_amd.analysis import slab_power as sp
    from abacusutils_amd.synth import synth_positions
    if a.rccl:
        from abacusutils_amd import _lib
        from abacusutils_amd.comm import RcclComm
        comm = RcclComm.from_env()
        assert 'torch' not in sys.modules, 'the RCCL transport must not pull in torch'
    else:
        import torch  # noqa: F401  (before the HIP library: one HIP runtime per process)
        import torch.distributed as dist
        if world > 1 or a.force_collectives:
            dist.init_process_group('gloo')
        from gloo_comm import GlooSlabComm
        comm = GlooSlabComm(force_collectives=a.force_collectives)
    L = 500.0
    # every rank draws the same catalogue and keeps an arbitrary half: route_particles moves them to their (folded) slabs
    pos = synth_positions(a.n, L, seed=11)
    w = np.random.default_rng(5).random(a.n, dtype=np.float32) + np.float32(0.5)
    mine = slice(comm.rank, None, comm.world)
    if a.rccl:   # device-resident particles, routed without leaving HBM
        p1, w1 = sp.route_particles(_lib.DeviceArray(np.ascontiguousarray(pos[mine])),
                                    _lib.DeviceArray(np.ascontiguousarray(w[mine])), L, comm, fold=True)
    else:
        p1, w1 = sp.route_particles(pos[mine], w[mine], L, comm, fold=True)
    kw = dict(kbins=a.kbins, mubins=4, paste='TSC', nmesh=a.nmesh, compensated=bool(a.compensated),
              interlaced=bool(a.interlaced), poles=[0, 2, 4])
    if a.backend == 'numpy':
        from slab_numpy_backend import NumpySlabBackend
        backend = NumpySlabBackend()
    else:
        backend = sp.HipSlabBackend()
        from abacusutils_amd import _lib as lib_
        for kv in a.option:
            name, _, val = kv.partition('=')
            lib_.set_option(name, int(val or 1))
    extra = {}
    if a.cross:
        pos2 = synth_positions(a.n // 2, L, seed=12)
        p2, _ = sp.route_particles(_lib.DeviceArray(np.ascontiguousarray(pos2[mine])) if a.rccl else pos2[mine], None, L, comm,
                                    fold=True)
        extra = dict(pos2=p2)
    t = sp.calc_power_slab(p1, L, comm=comm, backend=backend, w=w1, **kw, **extra)
    np.savez(f'{a.out}.rank{comm.rank}.npz', n_local=p1.shape[0], **{k: np.asarray(t[k]) for k in t.keys()})
    if a.rccl:
        info = comm.info()
        assert info['world'] == world and info['rccl_version'] > 0
        comm.barrier()
        comm.free()
    elif world > 1 or a.force_collectives:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
