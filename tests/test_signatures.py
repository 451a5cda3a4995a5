"""The drop-in surface keeps the reference's call signatures: parameter names, order and defaults of the mirror
functions against the AST of the reference's sources, read as text.  Build container only (the reference does not travel)."""
import ast
import inspect
from pathlib import Path

import pytest

REF = Path('/root/reference/abacusnbody')
pytestmark = pytest.mark.skipif(not REF.exists(), reason='the reference sources are only present in the build container')

SURFACE = {
    ('analysis/power_spectrum.py', 'abacusutils_amd.analysis.power_spectrum'): [
        'calc_power', 'calc_pk_from_deltak', 'get_k_mu_edges', 'get_field_fft', 'get_field', 'normalize_field',
        'get_W_compensated', 'bin_kmu', 'get_raw_power', 'shift_field_fft', 'get_interlaced_field_fft', 'bin_kppi',
        'project_3d_to_poles', 'pk_to_xi', 'expand_poles_to_3d', 'get_smoothing', 'get_delta_mu2'],
    ('analysis/tsc.py', 'abacusutils_amd.analysis.tsc'): ['tsc_parallel', 'partition_parallel'],
    ('analysis/cic.py', 'abacusutils_amd.analysis.cic'): ['cic_serial'],
    ('analysis/tpcf_corrfunc.py', 'abacusutils_amd.analysis.tpcf_corrfunc'): [
        'calc_xirppi_fast', 'calc_wp_fast', 'calc_multipole_fast', 'tpcf_multipole'],
    ('hod/GRAND_HOD.py', 'abacusutils_amd.hod.GRAND_HOD'): ['gen_gal_cat'],
    ('hod/menv.py', 'abacusutils_amd.hod.menv'): ['do_Menv_from_tree'],
    ('data/bitpacked.py', 'abacusutils_amd.data.bitpacked'): ['unpack_rvint', 'unpack_pids'],
    ('data/pack9.py', 'abacusutils_amd.data.pack9'): ['unpack_pack9'],
}
METHODS = ['__init__', 'run_hod', 'compute_ngal', 'compute_power', 'compute_xirppi', 'compute_wp', 'compute_multipole',
           'compute_clustering', 'apply_zcv', 'apply_zcv_xi', 'gal_reader', 'staging']


def _ref_functions(path, cls=None):
    tree = ast.parse((REF / path).read_text())
    body = tree.body
    if cls:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    return {n.name: n for n in body if isinstance(n, ast.FunctionDef)}


def _ref_params(node):
    a = node.args
    names = [x.arg for x in a.posonlyargs + a.args]
    ndef = len(a.defaults)
    defaults = {names[len(names) - ndef + q]: ast.unparse(d) for q, d in enumerate(a.defaults)}
    for x, d in zip(a.kwonlyargs, a.kw_defaults):
        names.append(x.arg)
        if d is not None:
            defaults[x.arg] = ast.unparse(d)
    return names, defaults


def _same_default(ours, ref_src):
    import numpy as np  # noqa: F401 - the reference's defaults are expressions over np
    MAX_THREADS = object()
    if 'MAX_THREADS' in ref_src or 'nthread' in ref_src:
        return True          # thread counts are accepted and ignored on the device
    try:
        ref = eval(ref_src, {'np': np, 'MAX_THREADS': MAX_THREADS})
    except Exception:
        return True
    try:
        if isinstance(ref, np.ndarray) or isinstance(ours, np.ndarray):
            return np.array_equal(np.asarray(ours), np.asarray(ref))
        return ours == ref or (ours is ref)
    except Exception:
        return False


def _check(ref_node, fn, where):
    names, defaults = _ref_params(ref_node)
    sig = inspect.signature(fn)
    ours = [p for p in sig.parameters.values() if p.kind not in (p.VAR_POSITIONAL, p.VAR_KEYWORD)]
    assert [p.name for p in ours][:len(names)] == names, f'{where}: parameters {[p.name for p in ours]} vs the reference\'s {names}'
    for p in ours[:len(names)]:
        if p.name in defaults:
            assert p.default is not inspect.Parameter.empty, f'{where}: {p.name} has a default in the reference'
            assert _same_default(p.default, defaults[p.name]), f'{where}: default of {p.name}: {p.default!r} vs {defaults[p.name]}'
        # (a parameter the reference requires may be optional here: every call the reference accepts is accepted)
    for p in ours[len(names):]:      # extensions must be optional
        assert p.default is not inspect.Parameter.empty, f'{where}: extra parameter {p.name} without a default'


@pytest.mark.parametrize('paths', list(SURFACE), ids=lambda p: p[0])
def test_module_functions(paths):
    import importlib
    ref = _ref_functions(paths[0])
    mod = importlib.import_module(paths[1])
    for name in SURFACE[paths]:
        assert name in ref, f'{name} is not a function of the reference\'s {paths[0]}'
        assert hasattr(mod, name), f'{paths[1]} does not export {name}'
        _check(ref[name], getattr(mod, name), f'{paths[1]}.{name}')


def test_abacus_hod_methods():
    from abacusutils_amd.hod.abacus_hod import AbacusHOD
    ref = _ref_functions('hod/abacus_hod.py', 'AbacusHOD')
    for name in METHODS:
        assert name in ref and hasattr(AbacusHOD, name), name
        _check(ref[name], getattr(AbacusHOD, name), f'AbacusHOD.{name}')


def test_calc_power_spectrum_is_calc_power():
    """BASELINE.json's north_star calls the estimator `calc_power_spectrum()`; the reference only has `calc_power`
    (analysis/power_spectrum.py:1131) - both names are the same callable here"""
    from abacusutils_amd.analysis import power_spectrum as ps
    assert ps.calc_power_spectrum is ps.calc_power and 'calc_power_spectrum' in ps.__all__
