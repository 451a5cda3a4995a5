import numpy as np, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden
from abacusutils_amd import synth
from abacusutils_amd.analysis import power_spectrum as ps
g = load_golden('power_f64')
Lb, N = float(g['meta.L']), int(g['meta.N'])
pos = synth.synth_positions(N, Lb, seed=300, clustered=True)
w = g['w']
for n in (24, 30):
    a = ps.get_field(pos.copy(), Lb, n, 'TSC', w, dtype=np.float64); b = g[f'n{n}.field_tsc']
    d = np.abs(a - b); i = np.unravel_index(d.argmax(), d.shape)
    print(n, 'field f32pos max', d.max() / np.abs(b).max(), 'at', i, a[i], b[i], 'n>1e-12:', int((d > 1e-12 * np.abs(b).max()).sum()))
    a = ps.get_field(pos.copy(), Lb, n, 'TSC', None, dtype=np.float64)
    print('   unweighted sum', a.sum(), 'min', a.min())
    W = ps.get_W_compensated(Lb, n, 'TSC', False)
    a = ps.get_field_fft(pos.copy(), Lb, n, 'TSC', w, W, True, False, dtype=np.float64); b = g[f'n{n}.fft_tsc_comp']
    print('   fft comp', np.abs(a - b).max() / np.abs(b).max())
