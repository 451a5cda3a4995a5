"""Multi-GPU P(k): slab decomposition of `calc_power` over the GPUs of one node.

New functionality relative to the reference (its TSC "can't readily scale to multiple nodes",
docs/tutorials/analysis/tsc.ipynb:19); the result is the same estimator as
abacusnbody/analysis/power_spectrum.py:1131-1319 evaluated on the union of the ranks' particles.

One process per GPU.  FOLDED SLABS: the mesh is cut into 2 W slabs of h = n / (2 W) planes and rank r owns slabs r and
r + W, i.e. the planes [r h, (r + 1) h) and the same range + n/2, together with the particles whose wrapped x lies there
(`route_particles(..., fold=True)` moves them).  Planes x and x + n/2 therefore sit on ONE rank, which is what the fused
form of the transform needs (the first radix-2 stage of the x transform pairs exactly those planes, csrc/fft.hip): every
rank runs the kernels of the single-GPU path on its pairs.  Per field:

    deposit           each of the two slabs + GHOST planes on both sides (TSC cloud +-1 cell, +1 for the half-cell
                      interlacing shift, +1 for round-half-even at the upper slab edge)      [device]
    ghost exchange    the ghost blocks go to the ring neighbours, which add them             [send/recv, 4 x 3 planes]
    z, y FFT passes   on the owned plane pairs (fused form: + first radix-2 stage of y and x)  [device]
    pencil transpose  (x_local, y, k) -> (y_local, x, k): ALL-TO-ALL of send[peer][2 h][y_local][k], which the y pass wrote
                      itself where the fused form runs                                       [RCCL all-to-all over xGMI:
                      one block per peer, every link busy at once]
    x FFT pass + binning   on the y-slab, straight from the receive buffer where the fused last pass serves the histogram
                      (no unpack, no spectrum in HBM); otherwise unpack, x pass, raw sums     [device]
    all-reduce        of the few-KB histogram, then bin_kmu's normalisation                  [RCCL all-reduce]

Collectives: `abacusutils_amd.comm.RcclComm` - RCCL through the C ABI (abacus_comm_*), enqueued on the library stream
between the kernels, no host synchronisation inside a spectrum; the pencil transpose is cut into chunks of plane pairs and
every chunk's all-to-all runs on the communicator's stream while the next chunk's z / y passes run.  A single process
needs no transport (`abacusutils_amd.comm.LocalComm`): its two slabs are the whole periodic mesh.  The host-staged stand-in
with the same methods over torch.distributed gloo lives with the tests (tests/gloo_comm.py: CPU container, or several
ranks sharing one GPU).
Restrictions: nmesh a power of two in [64, 2048] (hand-written FFT passes), nmesh % (2 W) == 0, nmesh / (2 W) >= GHOST.
"""
import ctypes as C

import numpy as np

from .. import _lib
from .power_spectrum import Table, get_k_mu_edges, get_W_compensated

GHOST = 3  # ghost planes on each side of a slab


class HipBuf:
    """float32 device buffer (DeviceArray) with the views the communicator needs"""

    def __init__(self, nfloat):
        self.n = int(nfloat)
        self.dev = _lib.DeviceArray(nbytes=self.n * 4, dtype=np.float32, shape=(self.n,))

    def ptr(self, off=0):
        return C.c_void_p(self.dev.ptr.value + 4 * int(off))

    def get(self, off, n):
        out = np.empty(int(n), dtype=np.float32)
        _lib.check(_lib.lib().abacus_memcpy_d2h(_lib.ptr(out), self.ptr(off), C.c_uint64(out.nbytes)))
        return out

    def set(self, off, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        _lib.check(_lib.lib().abacus_memcpy_h2d(self.ptr(off), _lib.ptr(arr), C.c_uint64(arr.nbytes)))

    def free(self):
        self.dev.free()


class HipSlabBackend:
    """device side of the slab estimator: thin wrappers over the abacus_slab_* entry points"""
    name = 'hip'

    def pitch(self, nmesh):
        return int(_lib.lib().abacus_slab_pitch(int(nmesh)))

    def __init__(self, keep_buffers=False):
        """keep_buffers: released work buffers are kept for the next call of the same shape (repeated spectra)"""
        self.keep = bool(keep_buffers)
        self._free = {}

    def new_buffer(self, nfloat):
        pool = self._free.get(int(nfloat))
        return pool.pop() if pool else HipBuf(nfloat)

    def release(self, buf):
        if self.keep:
            self._free.setdefault(buf.n, []).append(buf)
        else:
            buf.free()

    def drop_buffers(self):
        for pool in self._free.values():
            for b in pool:
                b.free()
        self._free = {}

    def upload_particles(self, pos, w):
        """host arrays are copied to HBM for this call; `_lib.DeviceArray`s (float32, resident) are used as they are"""
        dpos = pos if isinstance(pos, _lib.DeviceArray) else _lib.DeviceArray(np.ascontiguousarray(pos, dtype=np.float32))
        dw = w if (w is None or isinstance(w, _lib.DeviceArray)) else _lib.DeviceArray(np.ascontiguousarray(w, dtype=np.float32))
        return dpos, dw

    def deposit(self, particles, mesh, nmesh, xoff, nx_total, Lbox, offset, norm, paste, sub=0.0, xoff2=-1):
        """planes [xoff, xoff + nx_total) (mod nmesh) as rho * norm - sub - with xoff2 >= 0 a second window of nx_total planes
        from xoff2 behind the first; a plane in both windows takes its deposits in the first.  Particles outside are skipped"""
        pos, w = particles
        n = pos.shape[0]
        # a buffer padded to whole 16-plane tiles (calc_power_slab allocates it so) lets the deposit take the single-GPU path's
        # third-generation lists on the window planes
        planes = (nx_total if xoff2 < 0 else 2 * nx_total)
        have = mesh.n // (int(nmesh) * self.pitch(nmesh))
        nx_alloc = -(-planes // 16) * 16
        _lib.check(_lib.lib().abacus_slab_deposit_padded_dev(
            pos.ptr, C.c_int64(n), None if w is None else w.ptr, mesh.ptr(0), int(nmesh), int(xoff), int(xoff2), int(nx_total),
            C.c_double(Lbox), C.c_double(offset), C.c_double(norm), int(paste), C.c_double(sub), int(nx_alloc if have >= nx_alloc else 0)))

    def axpy(self, dst, dst_off, src, src_off, nfloat, add):
        _lib.check(_lib.lib().abacus_slab_axpy_dev(dst.ptr(dst_off), None if src is None else src.ptr(src_off),
                                                   C.c_int64(nfloat), C.c_float(add)))

    def fft_zy(self, mesh, off, send, nmesh, world, xsep, xg0, p0, pc, compact=None):
        """z / y passes of the plane pairs [p0, p0 + pc) of this rank's folded slab (first half at float offset `off`, global
        plane xg0; second half `xsep` planes behind it); send = None: in place, else into the send buffer of the transpose -
        compact = (Lbox, k_last): the compact one of `transpose_layout`"""
        if compact is not None:
            _lib.check(_lib.lib().abacus_slab_fft_zy_compact_dev(mesh.ptr(off), send.ptr(0), int(nmesh), int(world), C.c_int64(xsep),
                                                                 int(xg0), int(p0), int(pc), C.c_double(compact[0]), C.c_double(compact[1])))
            return
        _lib.check(_lib.lib().abacus_slab_fft_zy_dev(mesh.ptr(off), None if send is None else send.ptr(0), int(nmesh),
                                                     int(world), C.c_int64(xsep), int(xg0), int(p0), int(pc)))

    def transpose_layout(self, nmesh, world, Lbox, k_last):
        """complex elements per plane of the COMPACT transpose block for every destination rank (columns beyond the binning's
        last edge stay at home, csrc/fft.hip slab_layout), or None where the regular layout has to serve"""
        P = np.zeros(int(world), dtype=np.int64)
        rc = _lib.lib().abacus_slab_transpose_layout(int(nmesh), int(world), C.c_double(Lbox), C.c_double(k_last), _lib.ptr(P))
        if rc == 1:
            return None
        _lib.check(rc)
        return P

    def pack(self, mesh, off, send, nmesh, world, xsep, p0, pc):
        _lib.check(_lib.lib().abacus_slab_pack_dev(mesh.ptr(off), send.ptr(0), int(nmesh), int(world), C.c_int64(xsep),
                                                   int(p0), int(pc)))

    def unpack(self, recv, out, off, nmesh, world):
        _lib.check(_lib.lib().abacus_slab_unpack_dev(recv.ptr(0), out.ptr(off), int(nmesh), int(world)))

    def fft_x(self, data, off, nmesh, nyl):
        _lib.check(_lib.lib().abacus_slab_fft_x_dev(data.ptr(off), int(nmesh), int(nyl)))

    def raw_bytes(self, Nk, Nmu, poles):
        return int(_lib.lib().abacus_bin_raw_bytes(int(Nk), int(Nmu), _lib.ptr(poles), len(poles)))

    def bin_raw(self, fields, nmesh, y0, nyl, Lbox, W, interlaced, ke, me, poles):
        (a, ao), (as_, aso), (b, bo), (bs, bso) = fields
        raw = np.zeros(self.raw_bytes(len(ke) - 1, len(me) - 1, poles), dtype=np.uint8)
        p = lambda buf, off: None if buf is None else buf.ptr(off)  # noqa: E731
        _lib.check(_lib.lib().abacus_slab_bin_dev(p(a, ao), p(as_, aso), p(b, bo), p(bs, bso), int(nmesh), int(y0), int(nyl),
                                                  C.c_double(Lbox), None if W is None else _lib.ptr(W), int(interlaced),
                                                  _lib.ptr(ke), len(ke) - 1, _lib.ptr(me), len(me) - 1, _lib.ptr(poles),
                                                  len(poles), _lib.ptr(raw)))
        return raw

    xbin_pair = True       # xbin_raw takes field2: a second field (cross power) or the shifted deposit of an interlaced pair
    xbin_quad = True       # ... and field3 / field4: the interlaced pair of a second catalogue (interlaced cross power)

    def xbin_raw(self, field, nmesh, world, y0, nyl, Lbox, W, ke, me, poles, put_geom, from_transpose=False, field2=None,
                 interlaced=False, cross=False, field3=None, field4=None):
        """last x pass fused with the binning (one non-interlaced field; with field2 the cross power with a second one or,
        `interlaced`, the auto power of the interlaced pair (field, field2 = shifted); nmesh 1024 / 2048): raw sums, or None
        when the library does not serve this mesh / histogram that way (then unpack + fft_x + bin_raw).  from_transpose:
        `field` is the receive buffer of the pencil transpose, (peer, 2 h, y_local, k), not yet unpacked"""
        buf, off = field if field is not None else (None, 0)   # field None: a query (0 / None, nothing computed)
        raw = np.zeros(self.raw_bytes(len(ke) - 1, len(me) - 1, poles), dtype=np.uint8)
        fp = lambda f: None if f is None else f[0].ptr(f[1])  # noqa: E731
        rc = _lib.lib().abacus_slab_xbin_quad_dev(None if buf is None else buf.ptr(off), fp(field2), fp(field3), fp(field4),
                                                   3 if (interlaced and (cross or field3 is not None)) else 1 if interlaced
                                                   else 2 if (field2 is not None or cross) else 0,
                                                   int(nmesh), int(world), int(y0), int(nyl), C.c_double(Lbox),
                                                   None if W is None else _lib.ptr(W), _lib.ptr(ke), len(ke) - 1, _lib.ptr(me),
                                                   len(me) - 1, _lib.ptr(poles), len(poles), int(bool(put_geom)),
                                                   int(from_transpose), _lib.ptr(raw))
        if rc == 1:
            return None
        _lib.check(rc)
        return raw

    def finalize(self, raw, Lbox, Nk, Nmu, poles):
        return finalize_raw(raw, Lbox, Nk, Nmu, poles)

    def sync(self):
        _lib.sync()


def finalize_raw(raw, Lbox, Nk, Nmu, poles):
    """bin_kmu's normalisation of (all-reduced) raw sums through the C ABI (host only)"""
    power = np.zeros((Nk, Nmu), dtype=np.float32)
    N_mode = np.zeros((Nk, Nmu), dtype=np.int64)
    bp = np.zeros((len(poles), Nk), dtype=np.float32)
    Nmp = np.zeros(Nk, dtype=np.int64)
    k_avg = np.zeros((Nk, Nmu), dtype=np.float32)
    _lib.check(_lib.lib().abacus_bin_finalize(_lib.ptr(raw), C.c_double(Lbox), int(Nk), int(Nmu), _lib.ptr(poles),
                                              len(poles), _lib.ptr(power), _lib.ptr(N_mode), _lib.ptr(bp), _lib.ptr(Nmp),
                                              _lib.ptr(k_avg)))
    return power, N_mode, bp, Nmp, k_avg


def slab_owner(xw, Lbox, world, fold):
    """rank that owns wrapped float32 x: x-slabs of width L / W, or (fold) slab mod W of 2 W slabs - a rank then owns the
    slabs r and r + W, the decomposition of `calc_power_slab`"""
    nslab = 2 * world if fold else world
    k = np.clip((xw * (np.float32(nslab) / np.float32(Lbox))).astype(np.int64), 0, nslab - 1)
    return k % world


def route_particles(pos, w, Lbox, comm, fold=False):
    """send every particle to the rank that owns its slab (`slab_owner`; `fold=True` for `calc_power_slab`), for catalogs
    whose order is not slab-local (e.g. light-cone RSD moves galaxies across slabs).  With the RCCL
    communicator and `_lib.DeviceArray` inputs the particles never leave HBM (bucket sort + all-to-all-v on the device); host
    arrays and the gloo stand-in take the NumPy route below."""
    if getattr(comm, 'device', False) and isinstance(pos, _lib.DeviceArray):
        return comm.route_particles(pos, w, Lbox, fold=fold)
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    xw = pos[:, 0] - np.floor(pos[:, 0] / np.float32(Lbox)) * np.float32(Lbox)
    owner = slab_owner(xw, Lbox, comm.world, fold)
    order = np.argsort(owner, kind='stable')
    counts = np.bincount(owner, minlength=comm.world)
    starts = np.concatenate(([0], np.cumsum(counts)))
    cols = [pos[order]] + ([np.ascontiguousarray(w, dtype=np.float32)[order]] if w is not None else [])
    out = []
    for c in cols:
        parts = [np.ascontiguousarray(c[starts[p]:starts[p + 1]]) for p in range(comm.world)]
        got = comm.all_to_all_host(parts)
        out.append(np.concatenate([g.reshape(-1, 3) if c.ndim == 2 else g for g in got]))
    return out[0], (out[1] if w is not None else None)


def calc_power_slab(pos, Lbox, comm=None, backend=None, kbins=None, mubins=None, k_max=None, logk=False, paste='TSC',
                    nmesh=128, compensated=True, interlaced=True, w=None, pos2=None, w2=None, poles=None,
                    squeeze_mu_axis=True, n_total=None, n_total2=None):
    """`calc_power` (abacusnbody/analysis/power_spectrum.py:1131-1319) over folded slabs.  `pos` / `pos2` are THIS rank's
    particles (already inside its two slabs, see `route_particles(..., fold=True)`; others are ignored); every rank returns
    the full Table."""
    if comm is None:   # launched with WORLD_SIZE > 1: RCCL (one communicator per process, reused); else no transport
        from ..comm import default_comm
        comm = default_comm()
    backend = backend or HipSlabBackend()
    W, r = comm.world, comm.rank
    if nmesh % (2 * W) or nmesh // (2 * W) < GHOST:
        raise ValueError(f'nmesh={nmesh} must be divisible by twice the {W} ranks and leave at least {GHOST} planes per slab')
    if kbins is None:
        kbins = nmesh
    if k_max is None:
        k_max = np.pi * nmesh / Lbox
    return_mubins = mubins is not None
    if mubins is None:
        mubins = 1
    code = {'TSC': 0, 'CIC': 1}.get(paste.upper())
    if code is None:
        raise ValueError(f'Unknown pasting method {paste}')
    Wk = get_W_compensated(Lbox, nmesh, paste, interlaced).astype(np.float32) if compensated else None
    poles_arr = np.asarray(poles or [], dtype=np.int64)
    kbins, mubins = get_k_mu_edges(Lbox, k_max, kbins, mubins, logk)
    ke = np.ascontiguousarray(kbins, dtype=np.float64)
    me = np.ascontiguousarray(mubins, dtype=np.float64)

    h = nmesh // (2 * W)                  # plane pairs of a rank: planes [r h, (r + 1) h) and the same + nmesh / 2
    nyl = nmesh // W
    pitch = backend.pitch(nmesh)
    plane = nmesh * pitch
    G = GHOST if comm.collective else 0   # one rank without a transport: the periodic mesh has no ghost planes
    g = G * plane
    win = h + 2 * G                       # planes of one slab's window; the buffer holds the two windows back to back
    xsep = win                            # so the halves of a pair lie `win` planes apart
    d = Lbox / nmesh
    nfields = (2 if interlaced else 1) * (2 if pos2 is not None else 1)
    # (two windows back to back, padded to whole 16-plane tiles for the list build of the deposit)
    meshes = [backend.new_buffer(-(-2 * win // 16) * 16 * plane) for _ in range(nfields)]
    lazy = {}                             # send / recv buffers of the transpose, allocated when a step needs them (one rank whose

    def tbuf(name):                       # last pass bins straight from its mesh needs neither: 2 x 18 GB at 2048^3)
        if name not in lazy:              # ('send', 'recv'; 'recv2': the second field of a fused cross power keeps its own)
            if Pc is None:
                lazy[name] = backend.new_buffer(2 * h * plane)
            else:                         # compact: what goes out to all peers / what comes in from them (floats)
                lazy[name] = backend.new_buffer(2 * 2 * h * int(Pc.sum() if name == 'send' else W * Pc[r]))
        return lazy[name]

    ghost = backend.new_buffer(max(4 * g, 4))
    xa = r * h                            # first plane of the first slab; the second starts at xa + nmesh / 2
    own = g                               # float offset of the first owned plane

    # auto power of one non-interlaced field, or the cross power of two: the last x pass can bin straight from LDS (no
    # spectrum write + re-read)
    # (likewise a pair of fields through one pass: the interlaced pair of an auto power, the two fields of a cross power)
    try_xbin = hasattr(backend, 'xbin_raw') and (nfields == 1 or (nfields == 2 and getattr(backend, 'xbin_pair', False)) or
                                                 (nfields == 4 and getattr(backend, 'xbin_quad', False)))
    # ... and then nothing but that binning reads the transposed spectrum: the columns of a row beyond its last edge need not
    # cross the links (COMPACT transpose, csrc/fft.hip slab_layout: -21 % with bins up to the Nyquist frequency)
    Pc = None
    if try_xbin and nfields >= 2:       # asked before any work: will the pair / the four be served?  (else the plain three-pass form)
        try_xbin = backend.xbin_raw(None, nmesh, W, r * nyl, nyl, Lbox, Wk, ke, me, poles_arr, False, from_transpose=True,
                                    interlaced=interlaced, cross=pos2 is not None) is not None
    if try_xbin and comm.collective and hasattr(backend, 'transpose_layout') and hasattr(comm, 'all_to_all_piece_v'):
        if backend.xbin_raw(None, nmesh, W, r * nyl, nyl, Lbox, Wk, ke, me, poles_arr, False, from_transpose=True) is not None:
            Pc = backend.transpose_layout(nmesh, W, Lbox, float(ke[-1]))
    if Pc is not None:
        csoff = np.concatenate(([0], np.cumsum(2 * h * Pc)[:-1]))          # complex offset of every peer's block in the send buffer

    def spectrum(particles, ntot, offset, mesh, recv='recv'):
        norm = float(np.float32(float(nmesh) ** 3 / float(ntot)))   # dtype(field.size / tot_weight) (:856,894)
        # every cell is written as rho * norm - 1 (the overdensity's "-1" costs no pass of its own); a ghost block holds
        # contribution - 1, so its owner adds ghost + 1
        if comm.collective:
            backend.deposit(particles, mesh, nmesh, (xa - G) % nmesh, win, Lbox, offset, norm, code, sub=1.0,
                            xoff2=(xa + nmesh // 2 - G) % nmesh)
            # the 2 W slabs form one ring: slab v's neighbours are v - 1 and v + 1, i.e. the same half of the neighbouring
            # rank - except across the seam (rank W - 1 -> rank 0), where the halves swap
            for s_ in (0, 1):
                comm.ring_exchange(backend, mesh, s_ * win * plane, (s_ * win + G + h) * plane, ghost, g, recv_off=2 * s_ * g)
            for s_ in (0, 1):
                # ghost[2 s g : +g]      came from rank + 1 (the low ghosts of its half s)  = the last G owned planes of my
                #                        half s, or of my other half if rank + 1 wrapped to 0
                # ghost[(2 s + 1) g : +g] came from rank - 1 (the high ghosts of its half s) = the first G owned planes
                hi = s_ ^ 1 if r == W - 1 else s_
                lo = s_ ^ 1 if r == 0 else s_
                backend.axpy(mesh, (hi * win + h) * plane, ghost, 2 * s_ * g, g, 1.0)
                backend.axpy(mesh, (lo * win + G) * plane, ghost, (2 * s_ + 1) * g, g, 1.0)
        else:
            backend.deposit(particles, mesh, nmesh, 0, nmesh, Lbox, offset, norm, code, sub=1.0)
        # z / y passes and pencil transpose in chunks of plane pairs: chunk c is on the links (the communicator's stream)
        # while chunk c+1 is transformed; the passes write the send buffer.  Auto power of one non-interlaced field: the
        # last x pass then reads the receive buffer as it arrived and bins from LDS (no unpack pass, no x pass, no
        # spectrum in HBM); one rank without a transport needs neither buffer: its own mesh is the "received" block.
        nchunk = comm.transpose_chunks(h) if comm.collective else 1
        cp = h // nchunk
        direct = try_xbin and not comm.collective           # one rank, fused last pass: no transpose at all
        for c in range(nchunk):
            if Pc is not None:
                backend.fft_zy(mesh, own, tbuf('send'), nmesh, W, xsep, xa, c * cp, cp, compact=(Lbox, float(ke[-1])))
                for s_ in (0, 1):                             # the chunk's planes of either half: cp P[p] elements to peer p
                    comm.all_to_all_piece_v(backend, tbuf('send'), tbuf(recv),
                                            2 * (csoff + (s_ * h + c * cp) * Pc), 2 * cp * Pc,
                                            2 * (np.arange(W) * 2 * h + s_ * h + c * cp) * Pc[r], np.full(W, 2 * cp * Pc[r]),
                                            overlap=nchunk > 1)
                continue
            backend.fft_zy(mesh, own, None if direct else tbuf('send'), nmesh, W, xsep, xa, c * cp, cp)
            if comm.collective:
                for s_ in (0, 1):                             # the chunk's planes of either half within every peer block
                    comm.all_to_all_piece(backend, tbuf('send'), tbuf(recv), 2 * h * nyl * pitch,
                                          (s_ * h + c * cp) * nyl * pitch, cp * nyl * pitch, overlap=nchunk > 1)
        if comm.collective:
            comm.join()
        if direct:
            return (mesh, own)
        got = tbuf(recv) if comm.collective else tbuf('send')
        if try_xbin:                                          # (peer, 2 h, y_local, k) as delivered: for the fused last pass
            return (got, 0)
        backend.unpack(got, mesh, 0, nmesh, W)               # mesh now holds (y_local, x, k)
        backend.fft_x(mesh, 0, nmesh, nyl)
        return (mesh, 0)

    sets = [(pos, w, n_total)] + ([(pos2, w2, n_total2)] if pos2 is not None else [])
    fields = []
    mi = 0
    for p_, w_, nt in sets:
        ntot = nt if nt is not None else comm.all_reduce_int(p_.shape[0])   # tot_weight = len(pos), also with weights (:1021)
        particles = backend.upload_particles(p_, w_)
        # a second receive buffer only where the fused last pass reads both fields as they arrived; otherwise spectrum()
        # has unpacked the first one into its mesh before the second transpose starts
        # (four fields through one fused last pass: four receive buffers)
        rname = (lambda q: 'recv' if not q else f'recv{q + 1}') if try_xbin else (lambda q: 'recv')
        fields.append(spectrum(particles, ntot, 0.0, meshes[mi], rname(len(fields))))
        mi += 1
        if interlaced:
            fields.append(spectrum(particles, ntot, 0.5 * d, meshes[mi], rname(len(fields))))
            mi += 1
        else:
            fields.append((None, 0))
    if pos2 is None:
        fields += [(None, 0), (None, 0)]
    raw = None
    if try_xbin:
        xkw = (dict(field2=fields[1], field3=fields[2], field4=fields[3], interlaced=True, cross=True) if (pos2 is not None and interlaced)
               else dict(field2=fields[2]) if pos2 is not None else dict(field2=fields[1], interlaced=True) if interlaced else {})
        raw = backend.xbin_raw(fields[0], nmesh, W, r * nyl, nyl, Lbox, Wk, ke, me, poles_arr, r == 0,
                               from_transpose=2 if Pc is not None else True, **xkw)
        if raw is None and Pc is not None:
            raise RuntimeError('calc_power_slab: the fused last pass declined a compact transpose it had accepted')
        if raw is None:      # not served: unpack, x pass, binning
            for fi, mj in zip([q for q, f in enumerate(fields) if f[0] is not None], range(nfields)):   # field slot -> its mesh, in order
                src, off = fields[fi]
                if src is meshes[mj]:                         # one rank went straight from its mesh: it still has to be packed
                    backend.pack(meshes[mj], off, tbuf('send'), nmesh, W, xsep, 0, h)
                    src = tbuf('send')
                backend.unpack(src, meshes[mj], 0, nmesh, W)
                backend.fft_x(meshes[mj], 0, nmesh, nyl)
                fields[fi] = (meshes[mj], 0)
    if raw is None:
        raw = backend.bin_raw(fields, nmesh, r * nyl, nyl, Lbox, Wk, interlaced, ke, me, poles_arr)
    raw = comm.all_reduce_raw(raw, (len(ke) - 1) * (len(me) - 1))
    power, N_mode, bp, Nmp, k_avg = backend.finalize(raw, Lbox, len(ke) - 1, len(me) - 1, poles_arr)
    for b in meshes + list(lazy.values()) + [ghost]:
        if hasattr(backend, 'release'):
            backend.release(b)
        elif hasattr(b, 'free'):
            b.free()
    if squeeze_mu_axis and len(me) == 2:
        power, N_mode, k_avg = power[:, 0], N_mode[:, 0], k_avg[:, 0]
    res = dict(k_min=kbins[:-1], k_max=kbins[1:], k_mid=(kbins[1:] + kbins[:-1]) * 0.5, k_avg=k_avg, power=power,
               N_mode=N_mode)
    if len(poles_arr) > 0:
        res.update(poles=bp.T, N_mode_poles=Nmp)
    if return_mubins:
        mu_binc = (mubins[1:] + mubins[:-1]) * 0.5
        res.update(mu_min=np.broadcast_to(mubins[:-1], res['power'].shape),
                   mu_max=np.broadcast_to(mubins[1:], res['power'].shape),
                   mu_mid=np.broadcast_to(mu_binc, res['power'].shape))
    return Table(res, meta=dict(Lbox=Lbox, nmesh=nmesh, paste=paste, compensated=compensated, interlaced=interlaced,
                                n_ranks=W))
