"""MI355X drop-in for `abacusnbody.analysis.tpcf_corrfunc` (reference: abacusnbody/analysis/tpcf_corrfunc.py).

The reference wraps the third-party Corrfunc pair counters; here `DD`, `DDrppi` and `DDsmu` are provided by the
HIP cell-list kernel (csrc/pairs.hip, C ABI `abacus_paircount`) with Corrfunc's calling conventions as used by the
reference, so the reference's own wrappers bind to them unchanged.  The four wrapper entry points the rest of abacusutils calls

    calc_xirppi_fast     (:97-203)      calc_wp_fast (:301-372)
    calc_multipole_fast  (:206-298)     tpcf_multipole (:17-94)

are kept as signatures over ONE shared estimator, `_natural_estimator` (float32 casts of coordinates and bins before
counting, analytic RR of the periodic box in the reference's dtype and operation order, xi = DD / RR - 1); their results
are pinned bit for bit against the reference's functions run with a brute-force Corrfunc stand-in
(tests/golden/pair_wrappers.npz, oracle/make_golden.py).

Corrfunc is not vendored in the reference and none of its tests cover these functions: parity of the pair counts is
pinned against the brute-force float32 counter of the oracle only ("parity unpinned" with respect to Corrfunc).
"""
import ctypes as C

import numpy as np

from .. import _lib
from .._lib import check, ptr


def _f4(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _paircount(mode, X1, Y1, Z1, boxsize, bins, X2=None, Y2=None, Z2=None, pimax=0.0, npibins=0, mu_max=1.0,
               nmubins=0):
    """host arrays (cast to float32 like the reference, tpcf_corrfunc.py:134-139) or - all of them - `_lib.DeviceArray`
    columns of float32 / float64 already in HBM (the HOD catalogue: `abacus_paircount_dev`, no PCIe round trip)"""
    bins = _f4(bins)
    nb = len(bins) - 1
    nsub = 1 if mode == 0 else (npibins if mode == 1 else nmubins)
    out = np.zeros(nb * nsub, dtype=np.uint64)
    cols = [c for c in (X1, Y1, Z1, X2, Y2, Z2) if c is not None]
    if any(isinstance(c, _lib.DeviceArray) for c in cols):
        if not all(isinstance(c, _lib.DeviceArray) and c.dtype == cols[0].dtype for c in cols):
            raise TypeError('device-resident coordinates: every column must be a DeviceArray of one dtype')
        dt = {np.dtype(np.float32): 0, np.dtype(np.float64): 1}[cols[0].dtype]
        dp = lambda c: None if c is None else c.ptr     # noqa: E731
        check(_lib.lib().abacus_paircount_dev(
            int(mode), dp(X1), dp(Y1), dp(Z1), C.c_int64(X1.shape[0]), dp(X2), dp(Y2), dp(Z2),
            C.c_int64(0 if X2 is None else X2.shape[0]), dt, C.c_float(boxsize), ptr(bins), int(nb), C.c_float(pimax),
            int(npibins), C.c_float(mu_max), int(nmubins), ptr(out)))
        return out
    X1, Y1, Z1, X2, Y2, Z2 = map(_f4, (X1, Y1, Z1, X2, Y2, Z2))
    check(_lib.lib().abacus_paircount(
        int(mode), ptr(X1), ptr(Y1), ptr(Z1), C.c_int64(len(X1)), ptr(X2), ptr(Y2), ptr(Z2),
        C.c_int64(0 if X2 is None else len(X2)), C.c_float(boxsize), ptr(bins), int(nb), C.c_float(pimax),
        int(npibins), C.c_float(mu_max), int(nmubins), ptr(out)))
    return out


def _result(npairs, bins, nsub, extra):
    """structured array like Corrfunc's results: one row per (r-bin, sub-bin)"""
    nb = len(bins) - 1
    dt = [('rmin', 'f8'), ('rmax', 'f8'), ('npairs', 'u8')] + [(k, 'f8') for k in extra]
    res = np.zeros(nb * nsub, dtype=dt)
    res['rmin'] = np.repeat(bins[:-1], nsub)
    res['rmax'] = np.repeat(bins[1:], nsub)
    res['npairs'] = npairs
    for k, v in extra.items():
        res[k] = np.tile(v, nb)
    return res


def DD(autocorr, nthreads, binfile, X1, Y1, Z1, X2=None, Y2=None, Z2=None, periodic=True, boxsize=None, **kw):
    """3-D pair counts in r bins (Corrfunc.theory.DD as called at scripts/emulator/generate_cfs/generate_cf.py:65-74)"""
    if not periodic or boxsize is None:
        raise NotImplementedError('only periodic boxes with an explicit boxsize are supported')
    bins = np.asarray(binfile, dtype=np.float64)
    n = _paircount(0, X1, Y1, Z1, float(boxsize), bins, None if autocorr else X2, None if autocorr else Y2,
                   None if autocorr else Z2)
    return _result(n, bins, 1, {})


def DDrppi(autocorr, nthreads, binfile=None, pimax=None, X1=None, Y1=None, Z1=None, X2=None, Y2=None, Z2=None,
           periodic=True, boxsize=None, max_cells_per_dim=None, verbose=False, **kw):
    """pair counts in (rp, pi) with 1-unit pi bins up to pimax (Corrfunc.theory.DDrppi, tpcf_corrfunc.py:144-156)"""
    if not periodic or boxsize is None:
        raise NotImplementedError('only periodic boxes with an explicit boxsize are supported')
    bins = np.asarray(binfile, dtype=np.float64)
    npi = int(pimax)
    n = _paircount(1, X1, Y1, Z1, float(boxsize), bins, None if autocorr else X2, None if autocorr else Y2,
                   None if autocorr else Z2, pimax=float(pimax), npibins=npi)
    return _result(n, bins, npi, {'pimax': np.arange(1, npi + 1, dtype='f8')})


def DDsmu(autocorr, nthreads, binfile, mu_max, nmu_bins, X1, Y1, Z1, X2=None, Y2=None, Z2=None, periodic=True,
          boxsize=None, max_cells_per_dim=None, verbose=False, **kw):
    """pair counts in (s, mu) (Corrfunc.theory.DDsmu, tpcf_corrfunc.py:240-252)"""
    if not periodic or boxsize is None:
        raise NotImplementedError('only periodic boxes with an explicit boxsize are supported')
    bins = np.asarray(binfile, dtype=np.float64)
    n = _paircount(2, X1, Y1, Z1, float(boxsize), bins, None if autocorr else X2, None if autocorr else Y2,
                   None if autocorr else Z2, mu_max=float(mu_max), nmubins=int(nmu_bins))
    return _result(n, bins, int(nmu_bins), {'mumax': (np.arange(1, nmu_bins + 1) * mu_max / nmu_bins)})


def _check_int(name, v):
    if not isinstance(v, int):
        raise ValueError(f'{name} needs to be an integer')


def _natural_estimator(counter, sample1, sample2, edges, lbox, shell_measure, nsub, **counter_kw):
    """xi = DD / RR - 1 on a periodic box, shape (len(edges) - 1, nsub).

    counter         DDrppi | DDsmu (Corrfunc calling convention), called on float32 copies of the coordinates
    sample2         None for an autocorrelation (Corrfunc then returns ordered pairs, hence the factor 2 in RR,
                    which the reference also keeps for cross counts: tpcf_corrfunc.py:190-198,284-292,363-370)
    shell_measure   volume of a bin per unit (box volume)^-1 before the N1 N2 / L^3 normalisation, already shaped
                    (nbins, 1) or (nbins, nsub); RR is formed as measure / L^3 * N1 * N2 * 2 in that order and in the
                    dtype NumPy gives the reference's expression (float32 bins and box -> float32 RR)
    """
    # device-resident columns (the HOD catalogue in HBM) are cast to float32 on the device by abacus_paircount_dev
    cast = lambda cols: [c if isinstance(c, _lib.DeviceArray) else np.asarray(c).astype(np.float32) for c in cols]  # noqa: E731
    first = cast(sample1)
    n1 = float(len(first[0]))
    if sample2 is None or sample2[0] is None:
        second, n2, auto = {}, n1, 1
    else:
        second = dict(zip(('X2', 'Y2', 'Z2'), cast(sample2)))
        n2, auto = len(second['X2']), 0
    res = counter(auto, counter_kw.pop('nthreads'), X1=first[0], Y1=first[1], Z1=first[2], boxsize=lbox, periodic=True,
                  **second, **counter_kw)
    dd = res['npairs'].reshape(len(edges) - 1, nsub)
    rr = shell_measure / lbox**3 * n1 * n2 * 2
    return dd, rr


def tpcf_multipole(s_mu_tcpf_result, mu_bins, order=0):
    """Legendre multipole of xi(s, mu) over mu in [0, 1] bins, doubled to cover [-1, 1] (tpcf_corrfunc.py:17-94):
    (2l + 1)/2 * sum_mu xi * dmu * (L_l(mu) + L_l(-mu))."""
    from scipy.special import legendre
    xi = np.atleast_1d(s_mu_tcpf_result)
    edges = np.atleast_1d(mu_bins)
    ell = int(order)
    mid = (edges[:-1] + edges[1:]) / 2.0
    poly = legendre(ell)          # poly1d, evaluated like the reference does (same rounding)
    both_signs = poly(mid) + poly(-1.0 * mid)
    return (2.0 * ell + 1.0) / 2.0 * np.sum(xi * np.diff(edges) * both_signs, axis=1)


def calc_xirppi_fast(x1, y1, z1, rpbins, pimax, pi_bin_size, lbox, Nthread, num_cells=20, x2=None, y2=None, z2=None):
    """xi(rp, pi) in pi bins of `pi_bin_size` built from unit pi bins (tpcf_corrfunc.py:97-203)"""
    _check_int('pimax', pimax)
    _check_int('pi_bin_size', pi_bin_size)
    if pimax % pi_bin_size:
        raise ValueError('pi_bin_size needs to be an integer divisor of pimax, current values are ', pi_bin_size, pimax)
    edges = rpbins.astype(np.float32)
    lbox = np.float32(lbox)
    annulus = np.pi * (edges[1:] ** 2 - edges[:-1] ** 2) * pi_bin_size
    dd, rr = _natural_estimator(DDrppi, (x1, y1, z1), (x2, y2, z2), edges, lbox, annulus, pimax, nthreads=Nthread,
                                binfile=edges, pimax=np.float32(pimax), max_cells_per_dim=num_cells, verbose=False)
    grouped = dd.reshape(len(edges) - 1, pimax // pi_bin_size, pi_bin_size).sum(axis=2)
    return grouped / rr[:, None] - 1


def calc_multipole_fast(x1, y1, z1, sbins, lbox, Nthread, nbins_mu=50, num_cells=20, x2=None, y2=None, z2=None,
                        orders=[0, 2]):
    """xi_l(s), the requested orders concatenated, from DD(s, mu) (tpcf_corrfunc.py:206-298)"""
    edges = sbins.astype(np.float32)
    lbox = np.float32(lbox)
    mu_edges = np.linspace(0, 1, nbins_mu + 1)
    wedge = 2 * np.pi / 3 * (edges[1:, None] ** 3 - edges[:-1, None] ** 3) * (mu_edges[None, 1:] - mu_edges[None, :-1])
    dd, rr = _natural_estimator(DDsmu, (x1, y1, z1), (x2, y2, z2), edges, lbox, wedge, nbins_mu, nthreads=Nthread,
                                binfile=edges, mu_max=1, nmu_bins=nbins_mu, max_cells_per_dim=num_cells)
    xi = dd / rr - 1
    return np.concatenate([tpcf_multipole(xi, mu_edges, order=ell) for ell in orders])


def calc_wp_fast(x1, y1, z1, rpbins, pimax, lbox, Nthread, num_cells=30, x2=None, y2=None, z2=None):
    """wp(rp) = 2 * sum over unit pi bins of xi(rp, pi) (tpcf_corrfunc.py:301-372)"""
    _check_int('pimax', pimax)
    edges = rpbins.astype(np.float32)
    lbox = np.float32(lbox)
    annulus = np.pi * (edges[1:] ** 2 - edges[:-1] ** 2)
    dd, rr = _natural_estimator(DDrppi, (x1, y1, z1), (x2, y2, z2), edges, lbox, annulus, pimax, nthreads=Nthread,
                                binfile=edges, pimax=np.float32(pimax), max_cells_per_dim=num_cells)
    return 2 * np.sum(dd / rr[:, None] - 1, axis=1)
