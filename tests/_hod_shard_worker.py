"""worker for test_hod_shard: one rank of the sharded HOD.  `--backend oracle` uses the CPU oracle as the per-shard
populate (CPU CI, gloo); `--backend hip` the HIP gen_gal_cat (ranks may share one GPU)."""
import argparse
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--backend', default='oracle')
    ap.add_argument('--nhalo', type=int, default=60000)
    ap.add_argument('--npart', type=int, default=90000)
    ap.add_argument('--out', required=True)
    a = ap.parse_args()
    import torch  # noqa: F401
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        dist.init_process_group('gloo')
    from abacusutils_amd import synth
    from abacusutils_amd.hod import shard
    hd, pd, params = synth.synth_hod_inputs(a.nhalo, a.npart, seed=77)
    tracers = {'LRG': dict(synth.LRG_PARAMS), 'ELG': dict(synth.ELG_PARAMS), 'QSO': dict(synth.QSO_PARAMS)}
    tracers['ELG'].update(conf_c=0.4, conf_a=0.3)        # conformity: needs the host look-up through pinds
    if a.backend == 'oracle':
        from oracle import oracle
        populate = lambda h, p, t, pr, **kw: oracle.gen_gal_cat(h, p, t, pr, Nthread=2, **kw)  # noqa: E731
    else:
        populate = None
    from gloo_comm import GlooHodTransport
    comm = shard.HodComm(GlooHodTransport() if world > 1 else None)
    cat = shard.run_hod_sharded(hd, pd, tracers, params, comm=comm, populate=populate, rsd=True)
    counts = comm.all_reduce_counts({t: (c['Ncent'], len(c['x']) - c['Ncent']) for t, c in
                                     shard.run_hod_sharded(hd, pd, tracers, params, comm=comm, populate=populate,
                                                           rsd=True, gather=False).items()})
    with open(f'{a.out}.rank{comm.rank}.pkl', 'wb') as f:
        pickle.dump((cat, counts), f)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
