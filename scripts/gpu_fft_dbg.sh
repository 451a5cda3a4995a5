cd $GRAFT_REPO_ROOT
make -s -C oracle
python - <<'PY'
import numpy as np, os
from abacusutils_amd import synth
from abacusutils_amd.analysis.power_spectrum import get_field_fft, get_W_compensated
from oracle import oracle
box=1000.0
for nmesh in (64,128,256):
  for inter in (False, True):
    pos = synth.synth_positions(400000, box, seed=77, clustered=True)
    a = get_field_fft(pos.copy(), box, nmesh, 'TSC', None, None, False, inter)
    b = oracle.get_field_fft(pos.copy(), box, nmesh, 'TSC', None, None, False, inter, nthread=4)
    d = np.abs(a-b); i = np.unravel_index(d.argmax(), d.shape)
    os.environ['ABACUS_FFT_HIPFFT']='1'
    c = get_field_fft(pos.copy(), box, nmesh, 'TSC', None, None, False, inter)
    del os.environ['ABACUS_FFT_HIPFFT']
    d2 = np.abs(c-b)
    print(nmesh, inter, 'native: max abs diff %.3e at %s (|b|max %.3e, rms diff %.3e)' % (d.max(), i, np.abs(b).max(), np.sqrt((d**2).mean())), ' hipfft: max %.3e rms %.3e' % (d2.max(), np.sqrt((d2**2).mean())))
PY
