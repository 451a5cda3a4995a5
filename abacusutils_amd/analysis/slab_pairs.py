"""Multi-GPU pair counting: x-slab domains with a ghost layer of width r_max (BASELINE config 5).

New functionality relative to the reference (Corrfunc runs on one node's cores, analysis/tpcf_corrfunc.py:144-179).
One process per GPU.  Rank r owns the points whose wrapped x lies in [r L/W, (r+1) L/W); it receives from its two ring
neighbours the points of the second catalogue within r_max of its slab (the only exchange: r_max / (L/W) of a slab,
1.5 % of the box for r_max = 30 Mpc/h in the 2 Gpc/h box), counts the pairs (own point, own or ghost point) with the
single-GPU cell-list kernel on global coordinates (minimum image on the full box), and the few-hundred-byte histograms
are all-reduced.  Every ordered pair is counted exactly once, on the rank that owns its first point, so the result
equals `DD` / `DDrppi` / `DDsmu` on the union catalogue (integers: exactly).
Restriction: L/W >= r_max (the ghost layer reaches the direct neighbours only).
"""
import numpy as np

from ..comm import default_comm
from .slab_power import route_particles
from .tpcf_corrfunc import _paircount

MODES = {'r': 0, 'rppi': 1, 'smu': 2}


def _hip_counter(mode, p1, p2, boxsize, bins, **kw):
    if len(p1) == 0 or len(p2) == 0:
        nsub = 1 if mode == 0 else (kw['npibins'] if mode == 1 else kw['nmubins'])
        return np.zeros((len(bins) - 1) * nsub, dtype=np.uint64)
    return _paircount(mode, p1[:, 0], p1[:, 1], p1[:, 2], boxsize, bins, p2[:, 0], p2[:, 1], p2[:, 2], **kw)


def _ghosts(own, boxsize, margin, comm):
    """points of the ring neighbours within `margin` of this rank's slab (periodic in x)"""
    W, r = comm.world, comm.rank
    if W == 1:
        return np.empty((0, 3), dtype=np.float32)
    L = np.float32(boxsize)
    xw = own[:, 0] - np.floor(own[:, 0] / L) * L
    lo, hi = np.float32(r * boxsize / W), np.float32((r + 1) * boxsize / W)
    near_lo, near_hi = xw < lo + np.float32(margin), xw >= hi - np.float32(margin)
    left, right = (r - 1) % W, (r + 1) % W
    send = [np.empty((0, 3), dtype=np.float32) for _ in range(W)]
    if left == right:                       # two ranks: one peer, every point at most once
        send[left] = own[near_lo | near_hi]
    else:
        send[left], send[right] = own[near_lo], own[near_hi]
    got = comm.all_to_all_host([np.ascontiguousarray(s) for s in send])
    got = [g.reshape(-1, 3) for p, g in enumerate(got) if p != r]
    return np.concatenate(got) if got else np.empty((0, 3), dtype=np.float32)


def paircount_slab(mode, pos1, boxsize, bins, comm=None, pos2=None, pimax=0.0, npibins=0, mu_max=1.0, nmubins=0,
                   counter=None):
    """Global pair counts of catalogues distributed over ranks in any way.

    mode: 'r' | 'rppi' | 'smu' (Corrfunc's DD / DDrppi / DDsmu as the reference calls them); pos1 / pos2: this rank's
    (N, 3) part of the catalogue(s), pos2 None = autocorrelation (ordered pairs, no self pairs).  Returns the uint64
    histogram [nbins * (1 | npibins | nmubins)] on every rank."""
    comm = comm or default_comm()
    counter = counter or _hip_counter
    m = MODES[mode]
    bins = np.asarray(bins, dtype=np.float32)
    W = comm.world
    margin = float(bins[-1]) * (1 + 1e-5) + 1e-5 * float(boxsize)
    if W > 1 and boxsize / W < margin:
        raise ValueError(f'slab width {boxsize / W:g} is smaller than r_max = {float(bins[-1]):g}: use fewer ranks')
    auto = pos2 is None
    own1, _ = route_particles(np.asarray(pos1, dtype=np.float32).reshape(-1, 3), None, boxsize, comm)
    own2 = own1 if auto else route_particles(np.asarray(pos2, dtype=np.float32).reshape(-1, 3), None, boxsize, comm)[0]
    set2 = np.concatenate([own2, _ghosts(own2, boxsize, margin, comm)])
    kw = {}
    if m == 1:
        kw = dict(pimax=float(pimax), npibins=int(npibins))
    elif m == 2:
        kw = dict(mu_max=float(mu_max), nmubins=int(nmubins))
    counts = np.asarray(counter(m, own1, set2, float(boxsize), bins, **kw), dtype=np.uint64)
    if auto and bins[0] <= 0 and len(own1):     # (i, i) at zero separation was counted as a cross pair
        if m == 2:
            raise ValueError('autocorrelation in (s, mu) bins starting at s = 0 is not supported on the slab path')
        counts[0] -= np.uint64(len(own1))
    return comm.all_reduce_raw(counts.view(np.uint8), len(counts)).view(np.uint64)
