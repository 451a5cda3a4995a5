"""Per-rank kernel time of the slab P(k) at W ranks, measured on ONE GPU: rank R of W runs its deposit, z / y passes (y pass
writing the send buffer), and the last pass + binning from a receive buffer, with the collectives stubbed out (the ring
exchange and the all-to-all move nothing, so the spectrum is garbage - only the kernels' shapes and times are real).
What an N-GPU run adds to these numbers is the time on the links.  Usage: slab_virtual_rank.py [W] [R] [nmesh] [npart_total] [option=value ...]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from abacusutils_amd import _lib  # noqa: E402
from abacusutils_amd.analysis import slab_power as sp  # noqa: E402


class StubComm:
    collective, device = True, False

    def __init__(self, world, rank):
        self.world, self.rank = world, rank

    def ring_exchange(self, *a, **k):
        pass

    def all_to_all_piece(self, *a, **k):
        pass

    def all_to_all_piece_v(self, backend, send, recv, send_off, send_n, recv_off, recv_n, overlap=False):
        self.floats_out = getattr(self, 'floats_out', 0) + int(sum(int(send_n[p]) for p in range(self.world) if p != self.rank))

    def join(self):
        pass

    def transpose_chunks(self, npair):
        for c in (4, 2):
            if npair % c == 0 and npair // c >= 8:
                return c
        return 1

    def all_reduce_raw(self, raw, n):
        return raw

    def all_reduce_int(self, v):
        return int(v) * self.world


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    nmesh = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    ntot = int(float(sys.argv[4])) if len(sys.argv) > 4 else 100_000_000
    L = 2000.0
    _lib.set_device(0)
    for kv in sys.argv[5:]:                                  # option=value ...
        k, _, v = kv.partition('=')
        _lib.set_option(k, int(v or 1))
    n_local = ntot // W
    pos = np.random.default_rng(300 + R).random((n_local, 3), dtype=np.float32)
    slab = np.where(np.arange(n_local) < n_local // 2, R, R + W).astype(np.float32)
    pos[:, 0] = (pos[:, 0] * np.float32(0.99999) + slab) * np.float32(L / (2 * W))
    pos[:, 1:] *= np.float32(L)
    dpos = _lib.DeviceArray(pos)
    comm = StubComm(W, R)
    backend = sp.HipSlabBackend(keep_buffers=True)
    kw = dict(kbins=min(512, nmesh // 2), mubins=4, k_max=np.pi * nmesh / L + 1e-6, paste='TSC', nmesh=nmesh,
              compensated=False, interlaced=False, poles=[0, 2, 4], n_total=n_local * W)
    sp.calc_power_slab(dpos, L, comm=comm, backend=backend, **kw)
    _lib.sync()
    _lib.profile_reset()
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    steps = 4
    for _ in range(steps):
        sp.calc_power_slab(dpos, L, comm=comm, backend=backend, **kw)
    _lib.sync()
    dt = (time.perf_counter() - t0) / steps
    _lib.profile_enable(False)
    kern = {k: (round(ms / steps, 3), n // steps) for k, (ms, n) in _lib.profile_get().items() if n}
    print(json.dumps({'world': W, 'rank': R, 'nmesh': nmesh, 'particles_per_rank': n_local, 'ms_per_spectrum_kernels_only': dt * 1e3,
                      'kernels_ms_and_launches': kern,
                      'transpose_bytes_out_per_spectrum': 4 * getattr(comm, 'floats_out', 0) // (steps + 1) or None}))


if __name__ == '__main__':
    main()
