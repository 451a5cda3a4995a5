"""Multi-GPU AbacusHOD: shard the halo / particle subsample over ranks, populate each shard on its own GPU, merge.

The reference's unit of decomposition is the `chunk` of slab files (abacusnbody/hod/abacus_hod.py:87-94,209-226:
`AbacusHOD(..., chunk, n_chunks)` loads `numslabs / n_chunks` slabs per instance); here a chunk is a rank.  Every
decision of `gen_cent` / `gen_sats` is per halo row / per particle row (hod/GRAND_HOD.py:139-414, 825-1262) and the
only cross-table access is `keep_cent[pinds]`-style look-ups of a particle's host halo, so a shard is a contiguous
range of halos plus exactly the particles whose host lies in that range (with `pinds` rebased).  No data-path
collective: the merged catalogue (centrals of all ranks in rank order, then satellites of all ranks) is bit-identical
to the one a single process produces; only the per-tracer counts (`compute_ngal`) are all-reduced.
"""
import numpy as np

_HALO_PREFIX, _PART_PREFIX = 'h', 'p'


def shard_bounds(n, world):
    """`world + 1` row offsets splitting `n` rows into near-equal contiguous ranges"""
    return (np.arange(world + 1, dtype=np.int64) * int(n)) // int(world)


def shard_catalog(halo_data, particle_data, rank, world, balance='particles'):
    """Rank `rank`'s part of a staged catalogue (dicts in the `AbacusHOD.staging()` layout).

    balance='particles': halo ranges chosen so that every rank gets about the same number of particles
    (satellites dominate the work); 'halos': equal halo counts.  Returns (halo_shard, particle_shard) of array views;
    `pinds` is rebased to the shard's first halo."""
    nh = len(halo_data['hmass'])
    npart = len(particle_data['phmass'])
    pinds = particle_data.get('pinds')
    if pinds is None:
        # no host look-ups needed (hod/GRAND_HOD.py:1000-1010 only reads pinds for conformity): split rows evenly
        hb, pb = shard_bounds(nh, world), shard_bounds(npart, world)
        h0, h1, p0, p1 = hb[rank], hb[rank + 1], pb[rank], pb[rank + 1]
        sel = slice(p0, p1)
    else:
        pinds = np.asarray(pinds)
        ordered = npart == 0 or bool(np.all(pinds[1:] >= pinds[:-1]))
        if balance == 'particles' and ordered and npart:
            cuts = pinds[np.minimum(shard_bounds(npart, world)[1:-1], npart - 1)]   # halo that holds the cut particle
            hb = np.concatenate(([0], cuts, [nh])).astype(np.int64)
            hb = np.maximum.accumulate(hb)
        else:
            hb = shard_bounds(nh, world)
        h0, h1 = int(hb[rank]), int(hb[rank + 1])
        if ordered:
            p0, p1 = np.searchsorted(pinds, [h0, h1], side='left')
            sel = slice(int(p0), int(p1))
        else:
            sel = np.nonzero((pinds >= h0) & (pinds < h1))[0]
    halo = {k: v[h0:h1] for k, v in halo_data.items()}
    part = {k: v[sel] for k, v in particle_data.items()}
    if pinds is not None:
        part['pinds'] = part['pinds'] - h0
    return halo, part


def merge_catalogs(parts, tracers=None):
    """Per-rank `gen_gal_cat` results -> one catalogue in the single-process order: all centrals, then all satellites
    (`fast_concatenate` order, hod/GRAND_HOD.py:1265-1299,1563-1589)"""
    out = {}
    for tr in (tracers or parts[0].keys()):
        cols = [k for k in parts[0][tr].keys() if k != 'Ncent']
        ncs = [int(p[tr]['Ncent']) for p in parts]
        out[tr] = {k: np.concatenate([p[tr][k][:nc] for p, nc in zip(parts, ncs)] +
                                     [p[tr][k][nc:] for p, nc in zip(parts, ncs)]) for k in cols}
        out[tr]['Ncent'] = int(sum(ncs))
    return out


class HodComm:
    """control-plane collectives of the sharded HOD: sums of per-tracer counts and the merge of the per-rank mocks.

    transport: anything with `rank`, `world`, `all_reduce_array(int64 array)` and `all_gather_object(obj)` - an
    `abacusutils_amd.comm.RcclComm` on the GPUs (RCCL through the C ABI); None = a single process.  (The CPU tests pass a
    gloo transport of their own, tests/gloo_comm.py.)"""

    def __init__(self, transport=None):
        self.transport = transport
        self.rank = transport.rank if transport is not None else 0
        self.world = transport.world if transport is not None else 1

    def all_reduce_counts(self, counts):
        """sum {tracer: (Ncent, Nsat)}-style integer dicts over ranks"""
        keys = sorted(counts)
        vals = np.array([np.atleast_1d(counts[k]) for k in keys], dtype=np.int64)
        if self.world > 1:
            vals = np.asarray(self.transport.all_reduce_array(vals.copy())).reshape(vals.shape)
        return {k: tuple(int(x) for x in v) if np.ndim(counts[k]) else int(v[0]) for k, v in zip(keys, vals)}

    def gather_catalog(self, local, dst=None):
        """merge the per-rank catalogues; on every rank (dst=None) or only on `dst` (others get None).  Host-side:
        a mock is ~1e-3 of the particle subsample it was drawn from"""
        if self.world == 1:
            return local
        objs = self.transport.all_gather_object(local)
        return merge_catalogs(objs) if dst is None or self.rank == dst else None


def run_hod_sharded(halo_data, particle_data, tracers, params, comm=None, populate=None, gather=True, **kw):
    """Populate this rank's shard of a (replicated or memory-mapped) staged catalogue and merge.

    `populate(halo_shard, particle_shard, tracers, params, **kw)` defaults to the HIP `gen_gal_cat`."""
    if comm is None:   # launched with WORLD_SIZE > 1: the process's RCCL communicator; else a single process
        from ..comm import LocalComm, default_comm
        t = default_comm()
        comm = HodComm(None if isinstance(t, LocalComm) else t)
    if populate is None:
        from .GRAND_HOD import gen_gal_cat as populate
    h, p = shard_catalog(halo_data, particle_data, comm.rank, comm.world)
    local = populate(h, p, tracers, params, **kw)
    return comm.gather_catalog(local) if gather else local
