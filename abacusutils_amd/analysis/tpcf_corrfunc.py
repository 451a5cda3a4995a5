"""MI355X drop-in for `abacusnbody.analysis.tpcf_corrfunc` (reference: abacusnbody/analysis/tpcf_corrfunc.py).

The reference wraps the third-party Corrfunc pair counters; here `DD`, `DDrppi` and `DDsmu` are provided by the
HIP cell-list kernel (csrc/pairs.hip, C ABI `abacus_paircount`) with Corrfunc's calling conventions as used by the
reference, and the wrapper arithmetic (float32 casts, pi-bin regrouping, analytic RR, xi = DD/RR - 1, wp, multipoles)
follows the reference line by line:

    calc_xirppi_fast     (:97-203)      calc_wp_fast (:301-372)
    calc_multipole_fast  (:206-298)     tpcf_multipole (:17-94)

Corrfunc is not vendored in the reference and none of its tests cover these functions: parity of the pair counts is
pinned against the brute-force float32 counter of the oracle only ("parity unpinned" with respect to Corrfunc).
"""
import ctypes as C
import time

import numpy as np

from .. import _lib
from .._lib import check, ptr


def _f4(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _paircount(mode, X1, Y1, Z1, boxsize, bins, X2=None, Y2=None, Z2=None, pimax=0.0, npibins=0, mu_max=1.0,
               nmubins=0):
    X1, Y1, Z1, X2, Y2, Z2 = map(_f4, (X1, Y1, Z1, X2, Y2, Z2))
    bins = _f4(bins)
    nb = len(bins) - 1
    nsub = 1 if mode == 0 else (npibins if mode == 1 else nmubins)
    out = np.zeros(nb * nsub, dtype=np.uint64)
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


def tpcf_multipole(s_mu_tcpf_result, mu_bins, order=0):
    """Multipole of xi(s, mu) (tpcf_corrfunc.py:17-94; halotools' tpcf_multipole)."""
    from scipy.special import legendre
    s_mu_tcpf_result = np.atleast_1d(s_mu_tcpf_result)
    mu_bins = np.atleast_1d(mu_bins)
    order = int(order)
    mu_bin_centers = (mu_bins[:-1] + mu_bins[1:]) / (2.0)
    Ln = legendre(order)
    result = ((2.0 * order + 1.0) / 2.0
              * np.sum(s_mu_tcpf_result * np.diff(mu_bins) * (Ln(mu_bin_centers) + Ln(-1.0 * mu_bin_centers)), axis=1))
    return result


def calc_xirppi_fast(x1, y1, z1, rpbins, pimax, pi_bin_size, lbox, Nthread, num_cells=20, x2=None, y2=None, z2=None):
    """xi(rp, pi) (tpcf_corrfunc.py:97-203)"""
    if not isinstance(pimax, int):
        raise ValueError('pimax needs to be an integer')
    if not isinstance(pi_bin_size, int):
        raise ValueError('pi_bin_size needs to be an integer')
    if not pimax % pi_bin_size == 0:
        raise ValueError('pi_bin_size needs to be an integer divisor of pimax, current values are ', pi_bin_size, pimax)
    ND1 = float(len(x1))
    if x2 is not None:
        ND2 = len(x2)
        autocorr = 0
    else:
        autocorr = 1
        ND2 = ND1
    rpbins = rpbins.astype(np.float32)
    pimax = np.float32(pimax)
    x1, y1, z1 = (a.astype(np.float32) for a in (x1, y1, z1))
    lbox = np.float32(lbox)
    if autocorr == 1:
        results = DDrppi(autocorr, Nthread, binfile=rpbins, pimax=pimax, X1=x1, Y1=y1, Z1=z1, boxsize=lbox,
                         periodic=True, max_cells_per_dim=num_cells, verbose=False)
    else:
        x2, y2, z2 = (a.astype(np.float32) for a in (x2, y2, z2))
        results = DDrppi(autocorr, Nthread, binfile=rpbins, pimax=pimax, X1=x1, Y1=y1, Z1=z1, X2=x2, Y2=y2, Z2=z2,
                         boxsize=lbox, periodic=True, max_cells_per_dim=num_cells, verbose=False)
    DD_counts = results['npairs']
    DD_counts_new = np.array([np.sum(DD_counts[i:i + pi_bin_size]) for i in range(0, len(DD_counts), pi_bin_size)])
    DD_counts_new = DD_counts_new.reshape((len(rpbins) - 1, int(pimax / pi_bin_size)))
    RR_counts_new = (np.pi * (rpbins[1:] ** 2 - rpbins[:-1] ** 2) * pi_bin_size / lbox**3 * ND1 * ND2 * 2)
    xirppi = DD_counts_new / RR_counts_new[:, None] - 1
    return xirppi


def calc_multipole_fast(x1, y1, z1, sbins, lbox, Nthread, nbins_mu=50, num_cells=20, x2=None, y2=None, z2=None,
                        orders=[0, 2]):
    """xi_l(s) from DD(s, mu) (tpcf_corrfunc.py:206-298)"""
    ND1 = float(len(x1))
    if x2 is not None:
        ND2 = len(x2)
        autocorr = 0
    else:
        autocorr = 1
        ND2 = ND1
    sbins = sbins.astype(np.float32)
    x1, y1, z1 = (a.astype(np.float32) for a in (x1, y1, z1))
    lbox = np.float32(lbox)
    if autocorr == 1:
        results = DDsmu(autocorr, Nthread, sbins, 1, nbins_mu, x1, y1, z1, periodic=True, boxsize=lbox,
                        max_cells_per_dim=num_cells)
    else:
        x2, y2, z2 = (a.astype(np.float32) for a in (x2, y2, z2))
        results = DDsmu(autocorr, Nthread, sbins, 1, nbins_mu, x1, y1, z1, X2=x2, Y2=y2, Z2=z2, periodic=True,
                        boxsize=lbox, max_cells_per_dim=num_cells)
    DD_counts = results['npairs'].reshape((len(sbins) - 1, nbins_mu))
    mu_bins = np.linspace(0, 1, nbins_mu + 1)
    RR_counts = (2 * np.pi / 3 * (sbins[1:, None] ** 3 - sbins[:-1, None] ** 3)
                 * (mu_bins[None, 1:] - mu_bins[None, :-1]) / lbox**3 * ND1 * ND2 * 2)
    xi_s_mu = DD_counts / RR_counts - 1
    xi_array = []
    for neworder in orders:
        xi_array += [tpcf_multipole(xi_s_mu, mu_bins, order=neworder)]
    return np.concatenate(xi_array)


def calc_wp_fast(x1, y1, z1, rpbins, pimax, lbox, Nthread, num_cells=30, x2=None, y2=None, z2=None):
    """wp(rp) = 2 sum_pi xi(rp, pi) with 1 Mpc/h pi bins (tpcf_corrfunc.py:301-372)"""
    if not isinstance(pimax, int):
        raise ValueError('pimax needs to be an integer')
    ND1 = float(len(x1))
    if x2 is not None:
        ND2 = len(x2)
        autocorr = 0
    else:
        autocorr = 1
        ND2 = ND1
    rpbins = rpbins.astype(np.float32)
    pimax = np.float32(pimax)
    x1, y1, z1 = (a.astype(np.float32) for a in (x1, y1, z1))
    lbox = np.float32(lbox)
    if autocorr == 1:
        results = DDrppi(autocorr, Nthread, binfile=rpbins, pimax=pimax, X1=x1, Y1=y1, Z1=z1, boxsize=lbox,
                         periodic=True, max_cells_per_dim=num_cells)
    else:
        x2, y2, z2 = (a.astype(np.float32) for a in (x2, y2, z2))
        results = DDrppi(autocorr, Nthread, binfile=rpbins, pimax=pimax, X1=x1, Y1=y1, Z1=z1, X2=x2, Y2=y2, Z2=z2,
                         boxsize=lbox, periodic=True, max_cells_per_dim=num_cells)
    DD_counts = results['npairs'].reshape((len(rpbins) - 1, int(pimax)))
    RR_counts = np.pi * (rpbins[1:] ** 2 - rpbins[:-1] ** 2) / lbox**3 * ND1 * ND2 * 2
    xirppi = DD_counts / RR_counts[:, None] - 1
    return 2 * np.sum(xirppi, axis=1)
