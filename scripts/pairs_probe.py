"""Three DD(r) calls on the pair-count bench workload (1e7 uniform points, 13 log bins to 30 Mpc/h, 2 Gpc/h box) and nothing
else: the process to put under `rocprofv3 --pmc ...` (scripts/gpu_pairs_pmc.sh)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from abacusutils_amd import _lib  # noqa: E402
from abacusutils_amd.analysis.tpcf_corrfunc import _paircount  # noqa: E402

n, L = 10_000_000, 2000.0
p = np.random.default_rng(500).random((n, 3), dtype=np.float32) * np.float32(L)
dev = [_lib.DeviceArray(np.ascontiguousarray(p[:, i])) for i in range(3)]
bins = np.geomspace(0.1, 30.0, 14).astype(np.float32)
for _ in range(3):
    c = _paircount(0, *dev, L, bins)
print(int(c.sum()))
