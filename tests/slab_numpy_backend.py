"""TEST INFRASTRUCTURE: a NumPy/oracle implementation of the device side of the slab estimator
(abacusutils_amd/analysis/slab_power.py), same buffer layouts as the HIP entry points, so that the host
orchestration (ghost exchange, pencil transpose, histogram all-reduce) can be exercised with gloo at world_size 2
on a machine without a GPU.  Never imported by the product."""
import numpy as np
from numpy.polynomial import legendre

from abacusutils_amd import _lib
from abacusutils_amd.analysis.slab_power import finalize_raw
from oracle import oracle


class NumpyBuf:
    def __init__(self, n):
        self.a = np.zeros(int(n), dtype=np.float32)

    def get(self, off, n):
        return self.a[off:off + n].copy()

    def set(self, off, arr):
        arr = np.asarray(arr, dtype=np.float32).ravel()
        self.a[off:off + arr.size] = arr


class NumpySlabBackend:
    name = 'numpy'

    def pitch(self, nmesh):
        return int(_lib.lib().abacus_slab_pitch(int(nmesh)))   # pure host function of the C ABI

    def new_buffer(self, n):
        return NumpyBuf(n)

    def upload_particles(self, pos, w):
        return (np.array(pos, dtype=np.float32), None if w is None else np.asarray(w, dtype=np.float32))

    def deposit(self, particles, mesh, nmesh, xoff, nx_total, Lbox, offset, norm, paste, sub=0.0, xoff2=-1):
        assert paste == 0, 'CPU stand-in: TSC only'
        pos, w = particles
        full = np.zeros((nmesh,) * 3, dtype=np.float32)
        if len(pos):
            p = pos.copy()
            oracle.wrap_inplace(p, Lbox)
            oracle.tsc_scatter(p, full, Lbox, weights=w, offset=offset)
        pitch = self.pitch(nmesh)
        nwin = 1 if xoff2 < 0 else 2
        win = np.zeros((nwin * nx_total, nmesh, pitch), dtype=np.float32)
        win[:, :, :nmesh] = -np.float32(sub)           # planes that receive no deposit: 0 * norm - sub
        taken = set()                                  # a plane both windows hold receives its deposits in the first
        for k, x0 in enumerate((xoff, xoff2)[:nwin]):
            for i in range(nx_total):                  # plane i of the window = global plane (x0 + i) mod n
                gp = (x0 + i) % nmesh
                if gp not in taken:
                    taken.add(gp)
                    win[k * nx_total + i, :, :nmesh] = full[gp] * np.float32(norm) - np.float32(sub)
        mesh.a[:win.size] = win.ravel()

    def axpy(self, dst, dst_off, src, src_off, n, add):
        d = dst.a[dst_off:dst_off + n]
        if src is not None:
            d += src.a[src_off:src_off + n]
        d += np.float32(add)

    def _real(self, buf, off, nmesh, nx):
        pitch = self.pitch(nmesh)
        return buf.a[off:off + nx * nmesh * pitch].reshape(nx, nmesh, pitch)

    def _cplx(self, buf, off, nmesh, nx):
        pitch = self.pitch(nmesh)
        return buf.a[off:off + nx * nmesh * pitch].view(np.complex64).reshape(nx, nmesh, pitch // 2)

    def fft_zy(self, mesh, off, send, nmesh, world, xsep, xg0, p0, pc):
        """plain form: z and y transforms of the planes [p0, p0 + pc) of either half, then (send given) the pack step"""
        plane = nmesh * self.pitch(nmesh)
        for s_ in (0, 1):
            o = off + (s_ * xsep + p0) * plane
            r = self._real(mesh, o, nmesh, pc)[:, :, :nmesh].astype(np.float64)
            f = np.fft.fft(np.fft.rfft(r, axis=2), axis=1)
            c = self._cplx(mesh, o, nmesh, pc)
            c[:] = 0
            c[:, :, :nmesh // 2 + 1] = f.astype(np.complex64)
        if send is not None:
            self.pack(mesh, off, send, nmesh, world, xsep, p0, pc)

    def pack(self, mesh, off, send, nmesh, world, xsep, p0, pc):
        h, nyl = nmesh // (2 * world), nmesh // world
        plane = nmesh * self.pitch(nmesh)
        pcx = self.pitch(nmesh) // 2
        s = send.a[:2 * h * nmesh * pcx * 2].view(np.complex64).reshape(world, 2 * h, nyl, pcx)
        for s_ in (0, 1):
            c = self._cplx(mesh, off + (s_ * xsep + p0) * plane, nmesh, pc)
            for q in range(world):
                s[q, s_ * h + p0:s_ * h + p0 + pc] = c[:, q * nyl:(q + 1) * nyl, :]

    def unpack(self, recv, out, off, nmesh, world):
        h, nyl = nmesh // (2 * world), nmesh // world
        o = self._cplx(out, off, nmesh, nyl)             # (y_local, x, k)
        pcx = o.shape[2]
        r = recv.a[:2 * h * nmesh * pcx * 2].view(np.complex64).reshape(world, 2 * h, nyl, pcx).copy()
        for q in range(world):
            for s_ in (0, 1):
                x0 = s_ * (nmesh // 2) + q * h
                o[:, x0:x0 + h, :] = r[q, s_ * h:(s_ + 1) * h].transpose(1, 0, 2)

    def fft_x(self, data, off, nmesh, nyl):
        c = self._cplx(data, off, nmesh, nyl)
        c[:] = np.fft.fft(c.astype(np.complex128), axis=1).astype(np.complex64)

    def raw_bytes(self, Nk, Nmu, poles):
        return Nk * Nmu * 24 + int(np.count_nonzero(poles)) * Nk * 8

    def bin_raw(self, fields, nmesh, y0, nyl, Lbox, W, interlaced, ke, me, poles):
        """raw sums of bin_kmu (reference analysis/power_spectrum.py:170-293) over the rows of a y-slab"""
        n = nmesh
        kz = n // 2 + 1
        spec = [None if b is None else self._cplx(b, o, n, nyl)[:, :, :kz].astype(np.complex128) for b, o in fields]
        fy = np.fft.fftfreq(n, 1.0 / n)[y0:y0 + nyl].astype(np.int64)
        fx = np.fft.fftfreq(n, 1.0 / n).astype(np.int64)
        fz = np.arange(kz, dtype=np.int64)
        KY, KX, KZ = np.meshgrid(fy, fx, fz, indexing='ij')
        dk = 2 * np.pi / Lbox
        d = Lbox / n
        inv = 1.0 / float(n) ** 3

        def field(a, s):
            if interlaced:
                ph = np.exp(1j * 0.5 * d * dk * (KX + KY + KZ))
                f = (a + s * ph) * (0.5 * inv)
            else:
                f = a * inv
            if W is not None:
                Wd = W.astype(np.float64)
                f = f / (Wd[np.arange(n)[None, :, None]] * Wd[(y0 + np.arange(nyl))[:, None, None]] * Wd[fz[None, None, :]])
            return f

        fa = field(spec[0], spec[1])
        fb = field(spec[2], spec[3]) if spec[2] is not None else fa
        P = (fa * np.conj(fb)).real
        kmag2 = (KX * KX + KY * KY + KZ * KZ).astype(np.float32)
        with np.errstate(divide='ignore', invalid='ignore'):
            mu2 = np.where(kmag2 > 0, (KZ * KZ).astype(np.float32) * (np.float32(1) / kmag2), np.float32(0)).astype(np.float32)
        ke2 = ((ke / dk) ** 2).astype(np.float32)
        me2 = (me ** 2).astype(np.float32)
        Nk, Nmu = len(ke) - 1, len(me) - 1
        ok = (kmag2 >= ke2[0]) & (kmag2 < ke2[-1])
        bk = np.searchsorted(ke2[1:], kmag2, side='left')
        bm = np.minimum(np.searchsorted(me2[1:], mu2, side='left'), Nmu - 1)
        wt = np.where(KZ == 0, 1, 2)
        b = (bk * Nmu + bm)[ok]
        cnt = np.bincount(b, weights=wt[ok], minlength=Nk * Nmu).astype(np.uint64)
        s = np.bincount(b, weights=(wt * P)[ok], minlength=Nk * Nmu)
        ks = np.bincount(b, weights=(wt * np.sqrt(kmag2.astype(np.float64)))[ok], minlength=Nk * Nmu)
        parts = [cnt.view(np.uint8), s.view(np.uint8), ks.view(np.uint8)]
        mu = np.sqrt(mu2.astype(np.float64))
        for ell in poles:
            if ell != 0:
                c = np.zeros(int(ell) + 1)
                c[int(ell)] = 1
                pw = (2 * ell + 1) * legendre.legval(mu, c)
                parts.append(np.bincount(bk[ok], weights=(wt * P * pw)[ok], minlength=Nk).view(np.uint8))
        return np.concatenate(parts)

    def finalize(self, raw, Lbox, Nk, Nmu, poles):
        return finalize_raw(np.ascontiguousarray(raw), Lbox, Nk, Nmu, poles)

    def sync(self):
        pass
