"""worker for test_slab_pairs: one rank of the slab pair counter (`--backend oracle`: brute-force CPU counter as the
per-rank kernel, gloo; `--backend hip`: the HIP cell-list kernel, ranks may share one GPU)"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def catalogues(n=4000, L=100.0):
    rng = np.random.default_rng(17)
    a = (rng.random((n, 3), dtype=np.float32) - np.float32(0.5)) * np.float32(L)      # [-L/2, L/2) like the HOD output
    centres = (rng.random((40, 3), dtype=np.float32) - np.float32(0.5)) * np.float32(L)
    b = centres[rng.integers(0, 40, n // 2)] + rng.normal(0, 2.0, (n // 2, 3)).astype(np.float32)   # clustered
    b = (b + np.float32(L / 2)) % np.float32(L) - np.float32(L / 2)
    return a, b.astype(np.float32)


CASES = {
    'r_auto': dict(mode='r', bins=np.geomspace(0.5, 12.0, 9), cross=False),
    'r_auto_zero': dict(mode='r', bins=np.linspace(0.0, 10.0, 6), cross=False),
    'r_cross': dict(mode='r', bins=np.geomspace(0.5, 12.0, 9), cross=True),
    'rppi_auto': dict(mode='rppi', bins=np.geomspace(0.3, 10.0, 7), cross=False, pimax=12.0, npibins=6),
    'rppi_cross': dict(mode='rppi', bins=np.geomspace(0.3, 10.0, 7), cross=True, pimax=12.0, npibins=6),
    'smu_cross': dict(mode='smu', bins=np.geomspace(0.5, 11.0, 6), cross=True, mu_max=1.0, nmubins=5),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--backend', default='oracle')
    ap.add_argument('--out', required=True)
    a = ap.parse_args()
    import torch  # noqa: F401
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        dist.init_process_group('gloo')
    from abacusutils_amd.analysis import slab_pairs as sp
    from gloo_comm import GlooSlabComm
    comm = GlooSlabComm()
    counter = None
    if a.backend == 'oracle':
        from oracle import oracle

        def counter(m, p1, p2, boxsize, bins, **kw):
            nsub = 1 if m == 0 else (kw['npibins'] if m == 1 else kw['nmubins'])
            if len(p1) == 0 or len(p2) == 0:
                return np.zeros((len(bins) - 1) * nsub, dtype=np.uint64)
            return oracle.paircount_brute({0: 'r', 1: 'rppi', 2: 'smu'}[m], p1[:, 0], p1[:, 1], p1[:, 2], boxsize, bins,
                                          p2[:, 0], p2[:, 1], p2[:, 2], nthread=2, **kw)
    A, B = catalogues()
    mine = slice(comm.rank, None, comm.world)        # an arbitrary split: the router moves the points to their slabs
    res = {}
    for name, c in CASES.items():
        kw = {k: v for k, v in c.items() if k not in ('mode', 'bins', 'cross')}
        res[name] = sp.paircount_slab(c['mode'], A[mine], 100.0, c['bins'], comm=comm, pos2=B[mine] if c['cross'] else None,
                                      counter=counter, **kw)
    np.savez(f'{a.out}.rank{comm.rank}.npz', **res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
