"""P(k) leg of bench.py: TSC deposit + 3-D R2C FFT + fused (k,mu)/multipole binning on one MI355X.

Workload = BASELINE config 3 (seed 300, `rng.random((N,3), f4) * L`, L = 2000, scripts/power/bench.py:28-39
protocol): N = 1e8 particles on a 1024^3 mesh by default; `--nmesh 2048` is the north-star mesh.
A step = one full calc_power chain (non-interlaced, uncompensated like scripts/power/bench.py:31) with particles
already resident in HBM; the spectrum never leaves the device.
"""
import json
import os
import time

import numpy as np

HBM_PEAK_GBS = 8000.0
REPO = os.path.dirname(os.path.abspath(__file__))


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this same command
    (profiles/rNN/pmc_traffic.json, written by scripts/gpu_round.sh + scripts/collect_profiles.py: FETCH_SIZE and
    WRITE_SIZE in separate passes, gfx950 correction applied).  None when there is no committed pass."""
    import glob
    import json
    for f in sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*', 'pmc_traffic.json')), reverse=True):
        try:
            tab = json.load(open(f)).get(workload, {})
        except (OSError, ValueError):
            continue
        for k, e in tab.items():
            base = k.split('<')[0]
            # profiler labels vs kernel symbols: fft_cols_x <-> fft_cols, hod_filter <-> hod_filter32 (float32 shadows)
            if kernel == base or kernel.startswith(base + '_') or base.startswith(kernel + '_') or base == kernel + '32':
                return e['hbm_bytes_per_launch']
    return None


def bench_pk(args, dist, headline, nmesh=None, cpu=True, variants=True):
    import ctypes as C
    from abacusutils_amd import _lib
    from abacusutils_amd.analysis import power_spectrum as ps

    nmesh, n = nmesh or args.nmesh, args.npk
    L = 2000.0
    rng = np.random.default_rng(300 + dist.rank)
    pos = rng.random((n, 3), dtype=np.float32)
    pos *= np.float32(L)
    dpos = _lib.DeviceArray(pos)
    kbins, mubins = ps.get_k_mu_edges(L, np.pi * nmesh / L + 1e-6, min(512, nmesh // 2), 4, False)
    ke = np.ascontiguousarray(kbins, dtype=np.float64)
    me = np.ascontiguousarray(mubins, dtype=np.float64)
    poles = np.array([0, 2, 4], dtype=np.int64)
    outs = ps._alloc_outputs(len(ke) - 1, len(me) - 1, len(poles))
    lib = _lib.lib()

    def step(interlaced=0, W=None):
        _lib.check(lib.abacus_power_from_particles_dev(
            dpos.ptr, C.c_int64(n), None, None, C.c_int64(0), None, C.c_double(L), int(nmesh), 0,
            _lib.ptr(W), int(interlaced), _lib.ptr(ke), len(ke) - 1, _lib.ptr(me), len(me) - 1, _lib.ptr(poles),
            len(poles), *[_lib.ptr(o) for o in outs]))

    steps = max(1, min(args.steps, 10))
    _lib.sync()
    tf = time.perf_counter()
    step()                                   # the first call of this mesh: allocations, twiddle tables, the geometry descriptor
    _lib.sync()                              # of the (k, mu) bins, an exact (synchronous) list build
    first_call_ms = (time.perf_counter() - tf) * 1e3
    for _ in range(max(1, min(args.warmup, 2)) - 1):
        step()
    lib.abacus_power_geometry_ms.restype = C.c_double
    geometry_ms = float(lib.abacus_power_geometry_ms())   # one-off pass of the first call (cached per (nmesh, edges))
    xbin_gen = int(lib.abacus_power_xbin_generation())
    # per-kernel table from two untimed steps with every launch bracketed by HIP events; in the TIMED region only the dominant
    # kernel is bracketed (a pair of event records per launch costs a few microseconds of a 10-ms step and a barrier packet each)
    _lib.profile_reset()
    _lib.profile_enable(True)
    for _ in range(2):
        step()
    _lib.sync()
    _lib.profile_enable(False)
    kern = {k: ms / c for k, (ms, c) in _lib.profile_get().items() if c}
    dom0 = max(kern, key=kern.get)
    _lib.profile_reset()
    _lib.profile_select(dom0)
    _lib.profile_enable(True)
    dist.barrier()
    _lib.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    _lib.sync()
    t1 = time.perf_counter()
    dist.barrier()
    dt = dist.max(t1 - t0) / steps
    _lib.profile_enable(False)
    _lib.profile_select(None)
    kern.update({k: ms / c for k, (ms, c) in _lib.profile_get().items() if c})   # the dominant kernel: measured in the timed steps

    power = outs[0].copy()
    shot = L**3 / n
    M = float(nmesh) ** 3
    alg_bytes_total = 12.0 * n + 36.0 * M   # SURVEY.md 8d, non-interlaced
    REC, ENT = (1 + 2 / 128) ** 2 * (1 + 2 / 256), (1 + 2 / 16) ** 2 * (1 + 2 / 32)   # 1.04 block records, 1.345 tile entries per particle
    alg = {
        'hipfft_r2c': 24.0 * M,            # three 1-D passes x (read + write) of the 4M-byte mesh/half-spectrum
        'fft_z_r2c': 8.0 * M, 'fft_cols_y': 8.0 * M, 'fft_cols_x': 8.0 * M,   # one pass each: read 4M + write 4M
        'gfft_rows': 8.0 * M, 'gfft_cols_y': 8.0 * M, 'gfft_cols_x': 8.0 * M,  # mixed-radix passes (csrc/gfft.hip): the same
        'fft_x_bin': 4.0 * M,              # last pass fused with the binning: one read of the half-spectrum, nothing written
        'gfft_x_bin': 4.0 * M,             # the same for the mixed-radix meshes (csrc/gfft.hip)
        'tsc_tile_deposit': 4.0 * M + 8.0 * ENT * n,    # mesh written once + the 8-byte tile entries read
        'spectrum_bin': 4.0 * M,
        'tsc_bin_count': 12.0 * n,
        'tsc_bin_fill': 12.0 * n + 16.0 * 1.3 * n,
        # third-generation lists (csrc/tsc_lines3.hpp): positions read twice, 16-byte (particle, block) records written once and
        # read twice, 8-byte (particle, tile) entries written once.  REC / ENT: records and entries per particle of a uniform
        # catalogue with 16 x 16 x 32-cell tiles in blocks of 8 x 8 x 8 tiles
        'tsc_lines_count': 12.0 * n,
        'tsc_lines_coarse': 12.0 * n + 16.0 * REC * n,
        'tsc_lines_fcount': 16.0 * REC * n,
        'tsc_lines_fine': 16.0 * REC * n + 8.0 * ENT * n,
    }
    dom = max((k for k in kern if k in alg), key=lambda k: kern[k])
    ach = alg[dom] / (kern[dom] * 1e-3) / 1e9
    out = {
        'metric': f'wall-clock of {nmesh}^3 TSC+FFT P(k)',
        'value': dt * 1e3,
        'unit': 'ms',
        'n_gpus': dist.world,
        'steps': steps,
        'warmup': max(1, min(args.warmup, 2)),
        'ms_per_step': dt * 1e3,
        'higher_is_better': False,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': f'{"BASELINE config 3" if nmesh == 1024 else "C3-style"}: {n:.0e} uniform float32 particles (seed 300+rank), L=2000, nmesh={nmesh}, '
                               'TSC, non-interlaced, uncompensated, 4 mu bins, poles 0/2/4; particles resident in HBM',
                   'nmesh': nmesh, 'n_particles': n},
        'kernels_ms': {k: round(v, 4) for k, v in kern.items()},
        'geometry_ms': round(geometry_ms, 3),
        'first_call_ms': round(first_call_ms, 2),
        'geometry_note': 'N_mode, k_avg and the cell table of the (k, mu) bins depend on (nmesh, edges) alone: one pass over the '
                         f'modes at the first spectrum of a mesh / edge set, cached (fused last pass generation {xbin_gen}); outside the timed steps',
        'mean_P_over_shot_noise': float(np.mean(power[len(power) // 4:, :]) / shot),
        'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': ach / HBM_PEAK_GBS, 'algorithmic_bytes': alg[dom],
                     'per_kernel_frac': {k: round(alg[k] / (kern[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3) for k in kern if k in alg},
                     'traffic': pmc_traffic('pk2048', dom) if (nmesh, n) == (2048, 100_000_000) else None,
                     'whole_step_GBs': alg_bytes_total / dt / 1e9,
                     'whole_step_frac': alg_bytes_total / dt / 1e9 / HBM_PEAK_GBS},
    }
    # interlaced + compensated variant (the calc_power defaults): 24 N + 80 M algorithmic bytes
    if variants:
        try:
            W = ps.get_W_compensated(L, nmesh, 'TSC', True).astype(np.float32)
            step(1, W)
            step(1, W)
            _lib.sync()
            t1 = time.perf_counter()
            for _ in range(max(1, steps // 2)):
                step(1, W)
            _lib.sync()
            dti = (time.perf_counter() - t1) / max(1, steps // 2)
            _lib.profile_reset()
            _lib.profile_enable(True)
            step(1, W)
            _lib.sync()
            _lib.profile_enable(False)
            # per step: one list build, two fine passes / deposits / z / y passes, one fused last pass over both fields
            kern_i = {k: ms for k, (ms, c) in _lib.profile_get().items() if c}
            out['interlaced_compensated'] = {'ms_per_step': dti * 1e3, 'kernels_ms_per_step': kern_i,
                                             'whole_step_GBs': (24.0 * n + 80.0 * M) / dti / 1e9,
                                             'whole_step_frac': (24.0 * n + 80.0 * M) / dti / 1e9 / HBM_PEAK_GBS}
        except Exception as e:
            out['interlaced_compensated'] = {'error': repr(e)}
        # cross power of two fields (LRG x ELG of BASELINE config 5 on one GPU: a second catalogue of n / 2 particles), non-
        # interlaced: both fields stop after their y pass, the fused last pass bins Re(conj(a) b) from the pair of tiles
        try:
            n2 = n // 2
            pos2 = np.random.default_rng(900 + dist.rank).random((n2, 3), dtype=np.float32)
            pos2 *= np.float32(L)
            dpos2 = _lib.DeviceArray(pos2)
            del pos2

            def cstep():
                _lib.check(lib.abacus_power_from_particles_dev(
                    dpos.ptr, C.c_int64(n), None, dpos2.ptr, C.c_int64(n2), None, C.c_double(L), int(nmesh), 0, None, 0,
                    _lib.ptr(ke), len(ke) - 1, _lib.ptr(me), len(me) - 1, _lib.ptr(poles), len(poles), *[_lib.ptr(o) for o in outs]))
            cstep()
            cstep()
            _lib.sync()
            t1 = time.perf_counter()
            for _ in range(max(1, steps // 2)):
                cstep()
            _lib.sync()
            dtc = (time.perf_counter() - t1) / max(1, steps // 2)
            _lib.profile_reset()
            _lib.profile_enable(True)
            cstep()
            _lib.sync()
            _lib.profile_enable(False)
            out['cross'] = {'ms_per_step': dtc * 1e3, 'n_particles_2': n2,
                            'kernels_ms_per_step': {k: ms for k, (ms, c) in _lib.profile_get().items() if c},
                            # two deposits + two z / y passes + one read of both half-spectra
                            'whole_step_GBs': (12.0 * (n + n2) + 48.0 * M) / dtc / 1e9,
                            'whole_step_frac': (12.0 * (n + n2) + 48.0 * M) / dtc / 1e9 / HBM_PEAK_GBS}
            # the reference's DEFAULT estimator with a second catalogue: interlaced + compensated cross power, four fields through
            # one fused last pass (fft_x_bin2<.., QUAD>); `unfused_ms_per_step`: four complete transforms + spectrum_bin<INTER, CROSS>
            try:
                W4 = ps.get_W_compensated(L, nmesh, 'TSC', True).astype(np.float32)

                def qstep():
                    _lib.check(lib.abacus_power_from_particles_dev(
                        dpos.ptr, C.c_int64(n), None, dpos2.ptr, C.c_int64(n2), None, C.c_double(L), int(nmesh), 0, _lib.ptr(W4), 1,
                        _lib.ptr(ke), len(ke) - 1, _lib.ptr(me), len(me) - 1, _lib.ptr(poles), len(poles), *[_lib.ptr(o) for o in outs]))

                def timed(reps):
                    qstep()
                    _lib.sync()
                    t1_ = time.perf_counter()
                    for _ in range(reps):
                        qstep()
                    _lib.sync()
                    return (time.perf_counter() - t1_) / reps
                dtq = timed(max(1, steps // 3))
                _lib.profile_reset()
                _lib.profile_enable(True)
                qstep()
                _lib.sync()
                _lib.profile_enable(False)
                kq = {k: ms for k, (ms, c) in _lib.profile_get().items() if c}
                _lib.set_option('pk_noxbin_cross', 1)
                dtu = timed(max(1, steps // 3))
                _lib.set_option('pk_noxbin_cross', 0)
                out['interlaced_cross'] = {'ms_per_step': dtq * 1e3, 'unfused_ms_per_step': dtu * 1e3, 'n_particles_2': n2,
                                           'kernels_ms_per_step': kq}
            except Exception as e:
                _lib.set_option('pk_noxbin_cross', 0)
                out['interlaced_cross'] = {'error': repr(e)}
            dpos2.free()
        except Exception as e:
            out['cross'] = {'error': repr(e)}
    dpos.free()
    # the drop-in call itself: calc_power on the NumPy positions (PCIe included; never the `value`): the upload runs in batches
    # on a copy stream, each batch deposited while the next is on the link (csrc/power.hip, HostSrc)
    if nmesh == 1024 and dist.world == 1 and variants is False:
        try:
            ckw = dict(kbins=min(512, nmesh // 2), mubins=4, k_max=np.pi * nmesh / L + 1e-6, paste='TSC', nmesh=nmesh, compensated=False,
                       interlaced=False, poles=[0, 2, 4])
            th = []
            for _ in range(3):
                t1 = time.perf_counter()
                ps.calc_power(pos, L, **ckw)
                th.append((time.perf_counter() - t1) * 1e3)
            lib.abacus_power_last_batches.restype = C.c_double
            out['ms_per_call_host_arrays'] = round(min(th[1:]), 2)
            out['host_arrays_note'] = (f'calc_power(pos: NumPy float32 ({n}, 3), ...) per call, best of two after a first call of {th[0]:.1f} ms: '
                                       f'1.2 GB over PCIe in {int(lib.abacus_power_last_batches())} batches behind the deposits')
        except Exception as e:
            out['ms_per_call_host_arrays'] = repr(e)
    lib.abacus_power_release()
    if cpu and dist.rank == 0 and dist.world == 1 and not args.no_cpu:
        if nmesh == 1024 or host_memory_gb() < cpu_pk_host_gb(nmesh, n):
            out['cpu_baseline'] = cpu_baseline_pk(L, out['ms_per_step'] if nmesh == 1024 else None)      # config 3
        else:
            out['cpu_baseline'] = cpu_baseline_pk(L, out['ms_per_step'], nmesh=nmesh, n=n, reps=1)
    return out


def host_memory_gb():
    """memory this process may still take: MemAvailable, capped by what is left under the cgroup's limit"""
    avail = None
    try:
        for line in open('/proc/meminfo'):
            if line.startswith('MemAvailable'):
                avail = float(line.split()[1]) * 1024 / 1e9
    except OSError:
        pass
    for lim, cur in (('/sys/fs/cgroup/memory.max', '/sys/fs/cgroup/memory.current'),
                     ('/sys/fs/cgroup/memory/memory.limit_in_bytes', '/sys/fs/cgroup/memory/memory.usage_in_bytes')):
        try:
            m = open(lim).read().strip()
            if m != 'max' and float(m) < 1e15:
                left = (float(m) - float(open(cur).read().strip())) / 1e9
                avail = left if avail is None else min(avail, left)
            break
        except (OSError, ValueError):
            continue
    return avail if avail is not None else 0.0


def cpu_pk_host_gb(nmesh, n):
    """host memory of one oracle calc_power: float32 mesh + complex64 spectrum + float32 raw power + positions, with a
    quarter on top for the transform's own buffers"""
    M = float(nmesh) ** 3
    return 1.25 * (4.0 * M + 4.0 * M + 2.0 * M + 12.0 * n) / 1e9


def cpu_share():
    """(logical CPUs in the affinity mask, CPUs' worth of time the cgroup grants or None): an OpenMP team wider than the
    quota is throttled, not faster"""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    return aff, quota


def cpu_threads():
    """thread counts worth timing on this host: around the cgroup's CPU quota when there is one (teams beyond it only wait for
    their time slices), else the physical cores of one socket, of the box, and every logical CPU"""
    aff, quota = cpu_share()
    if quota:
        q = max(1, int(round(quota)))
        return sorted({t for t in (max(1, q // 2), q, 2 * q) if t <= aff})
    return sorted({t for t in (aff // 4, aff // 2, aff) if t >= 1})


def cpu_baseline_pk(L, gpu_ms=None, nmesh=1024, n=100_000_000, reps=2):
    """oracle (C+OpenMP stripe TSC, scipy pocketfft rfftn = the reference's FFT, OpenMP bin_kmu) on the bench's own workload -
    1e8 particles (seed 300) on a 1024^3 mesh (BASELINE config 3: one warm-up + one timed calc_power, about 10-20 s of CPU
    work on the box's cores) or on the metric's 2048^3 mesh (one timed call: 86 GB of host arrays)"""
    from oracle import oracle
    aff, quota = cpu_share()
    cores = aff if not quota else min(aff, max(1, int(round(2 * quota))))
    rng = np.random.default_rng(300)
    pos = rng.random((n, 3), dtype=np.float32)
    pos *= np.float32(L)
    kw = dict(kbins=min(512, nmesh // 2), mubins=4, k_max=np.pi * nmesh / L + 1e-6, paste='TSC', nmesh=nmesh, compensated=False,
              interlaced=False, poles=[0, 2, 4], nthread=cores, accum64=True)
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        res = oracle.calc_power(pos, L, **kw)          # wraps in place (a no-op here: the positions lie in [0, L))
        ts.append(time.perf_counter() - t)
    shot = L ** 3 / n
    what = 'BASELINE config 3' if nmesh == 1024 else "the metric's own mesh"
    calls = f'second of two calls; first {ts[0] * 1e3:.0f} ms' if reps > 1 else 'one call, no warm-up'
    return {'value': float(nmesh) ** 3 / ts[-1], 'unit': 'mesh cells/s', 'cores': cores, 'kind': 'port',
            'sample': f'{what}: nmesh {nmesh}, {n} particles, {ts[-1] * 1e3:.0f} ms per calc_power ({calls}), '
                      f'{cores} OpenMP / pocketfft threads (affinity {aff}, cgroup quota {quota})' +
                      (f'. The GPU step of this same workload, measured in this run: {gpu_ms:.2f} ms' if gpu_ms else
                       '. The GPU step of this same workload is the `pk_c3` leg of the line'),
            'mean_P_over_shot_noise': float(np.mean(np.asarray(res['power'])[len(res['power']) // 4:, :]) / shot),
            'ms': ts[-1] * 1e3}


def bench_pk_slab(args, dist):
    """strong scaling of ONE nmesh^3 P(k) over the N GPUs: deposit into the rank's two folded slabs (planes x and x + n/2 on
    one rank) with ghost planes, ring exchange, fused z / y passes writing the send buffer, all-to-all pencil transpose
    (chunked, overlapping the passes), last pass + binning from the receive buffer, all-reduce
    (abacusutils_amd/analysis/slab_power.py).  Particles (args.npk in total, uniform, generated inside each rank's slabs)
    are resident in HBM; collectives are RCCL through the C ABI (dist.comm)."""
    from abacusutils_amd import _lib
    from abacusutils_amd.analysis import slab_power as sp
    nmesh, ntot = args.nmesh, args.npk
    L = 2000.0
    W, r = dist.world, dist.rank
    from abacusutils_amd.comm import LocalComm
    comm = dist.comm if dist.comm is not None else LocalComm()
    backend = sp.HipSlabBackend(keep_buffers=True)
    n_local = ntot // W
    rng = np.random.default_rng(300 + r)
    pos = rng.random((n_local, 3), dtype=np.float32)
    # BASELINE config 4: the galaxies come out of halo-range shards (abacusutils_amd/hod/shard.py: a rank's halos are a
    # contiguous id range, not an x-slab), so every rank starts with particles from ALL OVER the box and the timed step begins
    # with `route_particles` (bucket sort by folded-slab owner + all-to-all-v over xGMI): (W - 1) / W of the catalogue moves.
    # `--slab-presorted`: the round-4 leg, particles generated inside the rank's own folded slabs (routing moves nothing)
    presorted = bool(getattr(args, 'slab_presorted', False))
    if presorted:
        slab = np.where(np.arange(n_local) < n_local // 2, r, r + W).astype(np.float32)
        pos[:, 0] = (pos[:, 0] * np.float32(0.99999) + slab) * np.float32(L / (2 * W))   # clear of the upper edge in float32
        pos[:, 1:] *= np.float32(L)
        assert np.array_equal(sp.slab_owner(pos[:, 0], L, W, True), np.full(n_local, r))
    else:
        pos *= np.float32(L)
    dpos = _lib.DeviceArray(pos)
    kw = dict(kbins=min(512, nmesh // 2), mubins=4, k_max=np.pi * nmesh / L + 1e-6, paste='TSC', nmesh=nmesh,
              compensated=False, interlaced=False, poles=[0, 2, 4], n_total=n_local * W)
    steps = max(1, min(args.steps, 5))
    can_route = getattr(comm, 'device', False) or W == 1

    def one_step():
        if presorted or not can_route or W == 1:
            return sp.calc_power_slab(dpos, L, comm=comm, backend=backend, **kw), 0
        rpos, _ = sp.route_particles(dpos, None, L, comm, fold=True)
        tab_ = sp.calc_power_slab(rpos, L, comm=comm, backend=backend, **kw)
        nrecv = rpos.shape[0]
        rpos.free()
        return tab_, nrecv

    tab, nrecv = one_step()                 # warm-up: allocations, tables, RCCL channels
    sent0 = comm.info()['bytes_sent'] if dist.comm is not None else 0
    dist.barrier()
    _lib.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        tab, nrecv = one_step()
    _lib.sync()
    t1 = time.perf_counter()
    dist.barrier()
    dt = dist.max(t1 - t0) / steps
    sent = (comm.info()['bytes_sent'] - sent0) / steps if dist.comm is not None else 0
    # the routing on its own (same particles, same destination): what it adds to the step and what it puts on the links
    routing = None
    if not presorted and can_route and W > 1:
        s0 = comm.info()['bytes_sent']
        dist.barrier()
        _lib.sync()
        tr = time.perf_counter()
        for _ in range(steps):
            rp, _ = sp.route_particles(dpos, None, L, comm, fold=True)
            rp.free()
        _lib.sync()
        routing = {'ms': dist.max(time.perf_counter() - tr) / steps * 1e3, 'bytes_sent_per_rank': (comm.info()['bytes_sent'] - s0) / steps,
                   'particles_received': int(nrecv)}
    power = np.asarray(tab['power'])
    shot = L**3 / (n_local * W)
    out = {'metric': f'wall-clock of one {nmesh}^3 TSC+FFT P(k) slab-decomposed over {W} GPUs', 'value': dt * 1e3,
           'unit': 'ms', 'n_gpus': W, 'rccl_ranks': W if dist.comm is not None else 0, 'steps': steps, 'scaling': 'strong',
           'higher_is_better': False, 'dtype': 'f32', 'data': 'synthetic',
           'config': {'workload': f'{n_local * W:.0e} uniform particles, nmesh {nmesh}, folded x-slabs, TSC, non-interlaced, '
                                  'RCCL through the C ABI: ring send/recv of ghost planes, chunked all-to-all of the '
                                  'pencil transpose (grouped ncclSend/ncclRecv), all-reduce of the histogram'},
           'bytes_sent_per_rank_per_step': sent, 'routing': routing,
           'particles': 'presorted into the folded slabs' if presorted else 'box-wide on every rank (halo-range shards): routed inside the timed step',
           'mean_P_over_shot_noise': float(np.mean(power[len(power) // 4:, :]) / shot)}
    # BASELINE config 5's spectrum over the same ranks: the cross power with a second catalogue of half the size (LRG x ELG), routed
    # like the first; both fields cross the links in the compact layout and one fused last pass bins the pair.
    # A rank that throws inside this leg would leave its peers waiting in the next collective until the orchestrator's timeout:
    # (1) the primary result is on stdout before the leg starts (launch_ranks keeps the LAST tagged line, the complete one
    # replaces it), (2) everything that can fail without a peer - the second catalogue's allocation and upload - happens first and
    # the ranks agree on it through one scalar all-reduce before any of them enters a collective of the cross step
    if r == 0 and dist.comm is not None:
        print('BENCH-LEG ' + json.dumps(dict(out, cross={'error': 'not reached: the leg ended inside the cross-power measurement'})), flush=True)
    n2 = n_local // 2
    dpos2, err2 = None, None
    try:
        pos2 = np.random.default_rng(900 + r).random((n2, 3), dtype=np.float32)
        pos2 *= np.float32(L)
        dpos2 = _lib.DeviceArray(pos2)
        del pos2
    except Exception as e:
        err2 = repr(e)
    ready = dist.sum(0.0 if err2 else 1.0)
    if ready < W:
        out['cross'] = {'error': err2 or f'skipped: {W - int(ready)} rank(s) could not stage the second catalogue'}
    else:
        try:
            ckw = dict(kw, n_total=n_local * W, n_total2=n2 * W)

            def cross_step():
                if not can_route or W == 1:
                    return sp.calc_power_slab(dpos, L, comm=comm, backend=backend, pos2=dpos2, **ckw)
                r1, _ = sp.route_particles(dpos, None, L, comm, fold=True)
                r2, _ = sp.route_particles(dpos2, None, L, comm, fold=True)
                t_ = sp.calc_power_slab(r1, L, comm=comm, backend=backend, pos2=r2, **ckw)
                r1.free()
                r2.free()
                return t_
            ctab = cross_step()
            dist.barrier()
            _lib.sync()
            tc = time.perf_counter()
            csteps = max(1, steps // 2)
            for _ in range(csteps):
                ctab = cross_step()
            _lib.sync()
            tcd = dist.max(time.perf_counter() - tc) / csteps
            out['cross'] = {'ms': tcd * 1e3, 'n_particles_2': n2 * W, 'steps': csteps,
                            'mean_abs_P_over_shot_noise': float(np.mean(np.abs(np.asarray(ctab['power'])[len(power) // 4:, :])) / shot)}
        except Exception as e:      # W == 1: a secondary measurement must not take the leg down.  W > 1: the peers are inside a
            out['cross'] = {'error': repr(e)}   # collective now; the provisional line above is what the orchestrator will keep
            if W > 1 and dist.comm is not None:
                if r == 0:
                    print('BENCH-LEG ' + json.dumps(out), flush=True)
                raise
    if dpos2 is not None:
        dpos2.free()
    # the pencil transpose on its own: every rank sends 1/W of its slab to each peer at once (one xGMI link per peer)
    if dist.comm is not None and W > 1:
        pitch = backend.pitch(nmesh)
        nfl = (nmesh // W) * nmesh * pitch
        a, b = backend.new_buffer(nfl), backend.new_buffer(nfl)
        comm.all_to_all(backend, a, b, nfl)
        dist.barrier()
        _lib.sync()
        t1 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            comm.all_to_all(backend, a, b, nfl)
        _lib.sync()
        ta = dist.max(time.perf_counter() - t1) / reps
        per_peer = 4.0 * nfl / W
        out['all_to_all'] = {'ms': ta * 1e3, 'bytes_per_peer': per_peer, 'GBs_per_link': per_peer / ta / 1e9,
                             'GBs_per_rank_out': per_peer * (W - 1) / ta / 1e9}
        a.free()
        b.free()
    backend.drop_buffers()
    dpos.free()
    return out


def bench_pairs(args, dist):
    """pair-counting leg (BASELINE config 5): DD(r) of 1e7 uniform points, 13 log bins 0.1-30 Mpc/h in the 2 Gpc/h box
    (the shape of scripts/emulator/generate_cfs/generate_cf.py:63-74).  Two timings: host arrays in / counts out like the
    Corrfunc call it replaces, and the coordinates already in HBM (abacus_paircount_dev: the HOD -> clustering step).
    Unit: candidate pair separations the kernel evaluates per second; the kernel is VALU-bound, not HBM-bound."""
    import ctypes as C
    from abacusutils_amd import _lib
    from abacusutils_amd.analysis.tpcf_corrfunc import _paircount
    n, L = 10_000_000, 2000.0
    rng = np.random.default_rng(500 + dist.rank)
    p = rng.random((n, 3), dtype=np.float32) * np.float32(L)
    x, y, z = (np.ascontiguousarray(p[:, i]) for i in range(3))
    bins = np.geomspace(0.1, 30.0, 14).astype(np.float32)
    _paircount(0, x, y, z, L, bins)
    dist.barrier()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        c = _paircount(0, x, y, z, L, bins)
    dist.barrier()
    dt_host = dist.max(time.perf_counter() - t0) / reps
    dev = [_lib.DeviceArray(a) for a in (x, y, z)]
    _paircount(0, *dev, L, bins)
    _lib.profile_reset()
    _lib.profile_enable(True)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        cd = _paircount(0, *dev, L, bins)
    dist.barrier()
    dt = dist.max(time.perf_counter() - t0) / reps
    _lib.profile_enable(False)
    kern = {k: ms / cnt for k, (ms, cnt) in _lib.profile_get().items() if cnt}
    for a in dev:
        a.free()
    assert np.array_equal(c, cd)
    cand, ncxy, ncz, R = C.c_uint64(0), C.c_int(0), C.c_int(0), C.c_int(0)
    _lib.check(_lib.lib().abacus_paircount_stats(C.byref(cand), C.byref(ncxy), C.byref(ncz), C.byref(R)))
    cand = float(cand.value)
    # what the first cell-list kernel evaluated for the same counts: every ordered pair of the 27 cells of size r_max
    ncell_r = int(np.floor(L / 30.0 * 0.9999))
    cand_27 = float(n) * (n / L**3) * 27 * (L / ncell_r) ** 3
    OPS = 20.0                  # vector instructions per candidate pair in the inner loop (ISA count of pair_count3<0, LUT>:
                                # 15 for a pair out of range, 23 for one that is counted, a quarter of them are)
    VALU_PEAK = 256 * 4 * 16 * 2.4e9   # lanes x clock: 3.9e13 lane-operations per second
    tk = kern.get('pair_count', dt * 1e3) * 1e-3
    out = {'metric': 'candidate pair separations per second, DD(r) to 30 Mpc/h', 'value': cand * dist.world / dt,
           'unit': 'pairs/s', 'n_gpus': dist.world, 'ms_per_call': dt * 1e3, 'ms_per_call_host_arrays': dt_host * 1e3,
           'n_points': n, 'pairs_counted': int(c.sum()), 'candidates_evaluated': cand,
           'candidates_of_the_27_cell_stencil': cand_27, 'cells_per_dim': int(ncxy.value), 'stencil_half_width_cells': int(R.value),
           'kernels_ms': {k: round(v, 4) for k, v in kern.items()}, 'dtype': 'f32',
           'roofline': {'bound': 'valu', 'kernel': 'pair_count', 'achieved': OPS * cand / tk / 1e12, 'peak': VALU_PEAK / 1e12,
                        'unit': 'T lane-ops/s', 'frac': OPS * cand / tk / VALU_PEAK,
                        'note': f'{OPS:.0f} vector instructions per candidate pair (inner loop of pair_count3); autocorrelation '
                                'by half stencil: every unordered pair evaluated once; two z-adjacent interior cells share one staged '
                                'stencil, which adds the ~19 % of candidates a slice point sees of the other cell\'s far plane'}}
    if dist.rank == 0 and dist.world == 1 and not args.no_cpu:
        from oracle import oracle
        cores, quota = cpu_share()
        best = None
        for t in sorted({t for t in (16, 32, 64, 128) if t <= cores} | ({cores} if not quota else set())):
            t0 = time.perf_counter()
            cc = oracle.paircount_cells('r', x, y, z, L, bins, nthread=t)
            tc = time.perf_counter() - t0
            if best is None or tc < best[0]:
                best = (tc, t)
        assert np.array_equal(cc, c)
        out['cpu_baseline'] = {'value': cand_27 / best[0], 'unit': 'pairs/s (27-cell stencil of r_max cells, all ordered pairs)',
                               'cores': best[1], 'kind': 'port', 'ms': best[0] * 1e3,
                               'sample': f'the full workload ({n} points), cell-list OpenMP counter of the oracle (Corrfunc\'s '
                                         'algorithm class without its AVX kernels), best of 16 / 32 / 64 / 128 threads (all of them where no cgroup quota applies); counts '
                                         'equal to the GPU\'s'}
    return out


def bench_catalog(args, dist):
    """catalogue-side kernels: rvint and packed-PID unpacking of 5e7 particles, device-resident
    (GB/s of algorithmic bytes: 12 B in + 24 B out, 8 B in + 31 B out), and the local mass environment of 1e7 halos in
    the 2 Gpc/h box (r_outer 5 Mpc/h, r_inner per halo; host arrays in / out like the reference call)."""
    from abacusutils_amd import _lib
    from abacusutils_amd.data import bitpacked
    from abacusutils_amd.hod.menv import do_Menv_from_tree
    n = 50_000_000
    rng = np.random.default_rng(700 + dist.rank)
    rv = rng.integers(-2**31, 2**31, size=(n, 3), dtype=np.int64).astype(np.int32)
    d_in = _lib.DeviceArray(rv)
    d_pos = _lib.DeviceArray(nbytes=n * 12, dtype=np.float32, shape=(n, 3))
    d_vel = _lib.DeviceArray(nbytes=n * 12, dtype=np.float32, shape=(n, 3))
    pk = rng.integers(0, 2**63, size=n, dtype=np.int64).astype(np.uint64)
    d_pk = _lib.DeviceArray(pk)
    outs = {'pid': _lib.DeviceArray(nbytes=n * 8, dtype=np.int64, shape=(n,)),
            'lagr_pos': _lib.DeviceArray(nbytes=n * 12, dtype=np.float32, shape=(n, 3)),
            'lagr_idx': _lib.DeviceArray(nbytes=n * 6, dtype=np.int16, shape=(n, 3)),
            'tagged': _lib.DeviceArray(nbytes=n, dtype=np.uint8, shape=(n,)),
            'density': _lib.DeviceArray(nbytes=n * 4, dtype=np.float32, shape=(n,))}
    L = _lib.lib()
    import ctypes as C

    def pids():
        _lib.check(L.abacus_unpack_pids(d_pk.ptr, C.c_int64(n), C.c_double(2000.0), C.c_int64(6912), 0, outs['pid'].ptr,
                                        outs['lagr_pos'].ptr, outs['lagr_idx'].ptr, outs['tagged'].ptr,
                                        outs['density'].ptr))
    bitpacked.unpack_rvint(d_in, 2000.0, posout=d_pos, velout=d_vel)
    pids()
    _lib.profile_reset()
    _lib.profile_enable(True)
    reps = 5
    for _ in range(reps):
        bitpacked.unpack_rvint(d_in, 2000.0, posout=d_pos, velout=d_vel)
        pids()
    _lib.profile_enable(False)
    kern = {k: ms / cnt for k, (ms, cnt) in _lib.profile_get().items() if cnt}
    out = {'n_particles': n,
           'unpack_rvint': {'ms': kern['unpack_rvint'], 'GB/s': n * 36 / kern['unpack_rvint'] / 1e6,
                            'particles/s': n / kern['unpack_rvint'] * 1e3},
           'unpack_pids': {'ms': kern['unpack_pids'], 'GB/s': n * 39 / kern['unpack_pids'] / 1e6,
                           'particles/s': n / kern['unpack_pids'] * 1e3}}
    # pack9: 5e7 records, a cell header every ~85 records like the Mini_N64_L32 slices (9 B in, 24 B out per particle)
    p9 = rng.integers(0, 255, size=(n, 9), dtype=np.int64).astype(np.uint8)     # first byte never 0xFF ...
    p9[::85, 0] = 0xFF                                                           # ... except at the headers
    d_p9 = _lib.DeviceArray(p9)
    npart = C.c_int64(0)

    def pack9():
        _lib.check(L.abacus_unpack_pack9(d_p9.ptr, C.c_int64(n), C.c_double(2000.0), C.c_double(208774.9), 0, d_pos.ptr,
                                         d_vel.ptr, C.byref(npart)))
    pack9()
    _lib.profile_reset()
    _lib.profile_enable(True)
    for _ in range(reps):
        pack9()
    _lib.profile_enable(False)
    kern = {k: ms / cnt for k, (ms, cnt) in _lib.profile_get().items() if cnt}
    t9 = kern['pack9_count'] + kern['pack9_emit']
    out['unpack_pack9'] = {'ms': t9, 'kernels_ms': {k: round(v, 4) for k, v in kern.items() if k.startswith('pack9')},
                           'GB/s': (n * 9 + npart.value * 24) / t9 / 1e6, 'particles/s': npart.value / t9 * 1e3}
    for a in (d_in, d_pos, d_vel, d_pk, d_p9, *outs.values()):
        a.free()
    nh, box = 10_000_000, 2000.0
    hpos = (rng.random((nh, 3), dtype=np.float32) - np.float32(0.5)) * np.float32(box)
    hmass = 10 ** (10.5 + rng.exponential(0.45, nh))
    rin = (0.1 + 0.4 * rng.random(nh)).astype(np.float32)
    do_Menv_from_tree(hpos, hmass, rin, 5.0, False, box, mcut=1e11)
    t0 = time.perf_counter()
    menv = do_Menv_from_tree(hpos, hmass, rin, 5.0, False, box, mcut=1e11)
    dt_host = time.perf_counter() - t0
    # the same call with the halo table resident in HBM (how prepare_sim's device path holds it): the three kernels alone
    d_pos, d_mass, d_rin = _lib.DeviceArray(hpos), _lib.DeviceArray(hmass), _lib.DeviceArray(rin)
    d_out = _lib.DeviceArray(nbytes=nh * 8, dtype=np.float64, shape=(nh,))
    do_Menv_from_tree(d_pos, d_mass, d_rin, 5.0, False, box, mcut=1e11, out=d_out)
    _lib.profile_reset()
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        do_Menv_from_tree(d_pos, d_mass, d_rin, 5.0, False, box, mcut=1e11, out=d_out)
    dt = (time.perf_counter() - t0) / reps
    _lib.profile_enable(False)
    kern = {k: ms / cnt for k, (ms, cnt) in _lib.profile_get().items() if cnt}
    assert np.allclose(d_out.get(), menv, rtol=1e-10, atol=1e-12 * float(hmass.max()))   # the order inside a cell is not fixed
    for a in (d_pos, d_mass, d_rin, d_out):
        a.free()
    out['menv'] = {'n_halos': nh, 'centres': int((hmass > 1e11).sum()), 'ms_per_call': dt * 1e3,
                   'ms_per_call_host_arrays': dt_host * 1e3,
                   'note': 'ms_per_call: pos / mass / r_inner and the result resident in HBM; ms_per_call_host_arrays: NumPy in, '
                           'NumPy out like the reference call (240 MB up, 80 MB down over PCIe)',
                   'kernels_ms': {k: round(v, 4) for k, v in kern.items() if k.startswith('menv')},
                   'halos/s': nh / dt, 'mean_Menv': float(menv.mean())}
    if dist.rank == 0 and dist.world == 1 and not args.no_cpu:
        from oracle import oracle
        m = 5_000_000
        t = time.perf_counter()
        oracle.unpack_rvint(rv[:m], 2000.0)
        tr = time.perf_counter() - t
        t = time.perf_counter()
        oracle.unpack_pids(pk[:m], box=2000.0, ppd=6912)
        tp = time.perf_counter() - t
        out['cpu_baseline'] = {'unpack_rvint particles/s': m / tr, 'unpack_pids particles/s': m / tp, 'cores': 1,
                               'kind': 'port', 'sample': f'{m} particles, NumPy restatement'}
    return out


def bench_prepare(args, dist):
    """prepare_sim's data-parallel core (reference hod/prepare_sim.py:296-1052 between loader and writer):
    one synthetic CompaSO-like slab (synth.synth_compaso_slabs: 1e6 halos, ~8e6 subsample particles), MT sample with satellite
    rank columns and assembly-bias ranks, selection drawn on the device (Philox).  The reference walks the halos in a
    Python loop; the CPU baseline is the oracle's NumPy restatement of that loop on a bounded sample of the same slab."""
    from abacusutils_amd import _lib
    from abacusutils_amd.hod import prepare_sim as prep
    from abacusutils_amd.synth import synth_compaso_slabs
    nh = 1_000_000
    slabs, header = synth_compaso_slabs(numslabs=1, n_halo=nh, seed=900 + dist.rank, lbox=2000.0, subsample_frac=0.006)
    halos, parts = slabs[0]['halos'], slabs[0]['parts']
    Mpart, h = header['ParticleMassHMsun'], header['H0'] / 100.0
    kw = dict(MT=True, want_ranks=True, want_AB=True, Lbox=header['BoxSize'])
    # warm-up: scratch allocations, and TWO sets of page-locked output columns (the caller holds one result while the next slab is
    # prepared: a pipeline over slabs reaches that state after its second slab; pinning ~400 MB costs tens of ms once)
    keep_a = prep.prepare_slab_arrays(halos, parts, Mpart, h, rng=7, **kw)
    keep_b = prep.prepare_slab_arrays(halos, parts, Mpart, h, rng=8, **kw)
    del keep_a, keep_b
    _lib.sync()
    reps = 3
    t0 = time.perf_counter()
    for r in range(reps):
        H, P, mask = prep.prepare_slab_arrays(halos, parts, Mpart, h, rng=11 + r, **kw)
    _lib.sync()
    dt = (time.perf_counter() - t0) / reps
    # the per-kernel table from one more, untimed call (with the profiler on the rank kernel stays on the library stream instead of
    # running beside the copies of the other columns)
    _lib.profile_reset()
    _lib.profile_enable(True)
    prep.prepare_slab_arrays(halos, parts, Mpart, h, rng=11, **kw)
    _lib.sync()
    _lib.profile_enable(False)
    kern = {k: round(ms, 4) for k, (ms, n) in _lib.profile_get().items() if n}
    # the same slab with its CompaSO columns already in HBM (where the reader's rvint / PID unpack kernels leave them), and through the
    # column-by-column path of rounds 3 - 5 (host gathers of the kept rows)
    extra = {}
    try:
        dh = {k: _lib.DeviceArray(v) for k, v in halos.items()}
        dp = {k: _lib.DeviceArray(v) for k, v in parts.items()}
        prep.prepare_slab_arrays(dh, dp, Mpart, h, rng=5, **kw)
        _lib.sync()
        t1 = time.perf_counter()
        for r in range(reps):
            prep.prepare_slab_arrays(dh, dp, Mpart, h, rng=21 + r, **kw)
        _lib.sync()
        extra['ms_per_slab_device_columns'] = (time.perf_counter() - t1) / reps * 1e3
        for a in list(dh.values()) + list(dp.values()):
            a.free()
        _lib.set_option('prep_columnwise', 1)
        prep.prepare_slab_arrays(halos, parts, Mpart, h, rng=5, **kw)
        t1 = time.perf_counter()
        for r in range(reps):
            prep.prepare_slab_arrays(halos, parts, Mpart, h, rng=31 + r, **kw)
        extra['ms_per_slab_column_by_column'] = (time.perf_counter() - t1) / reps * 1e3
    except Exception as e:   # noqa: BLE001
        extra['error'] = repr(e)
    finally:
        _lib.set_option('prep_columnwise', 0)
    out = {'metric': 'halos/s through prepare_slab (subsample + particle selection + rank columns)', 'value': nh / dt,
           'unit': 'halos/s', 'ms_per_slab': dt * 1e3, 'n_halos': nh, 'n_particles': int(len(parts['pos'])),
           'halos_kept': int(mask.sum()), 'particles_kept': int(len(P['pos'])), 'kernels_ms': kern,
           'kernels_ms_total': round(sum(kern.values()), 3),
           'note': 'ms_per_slab: NumPy columns in, the two tables as NumPy columns out - one pass through HBM (abacus_prepare_slab: every '
                   'input uploaded once, kept rows gathered on the device, every output column copied out once into page-locked memory); '
                   'ms_per_slab_device_columns: the CompaSO columns already in HBM; ms_per_slab_column_by_column: the path of rounds 3 - 5; '
                   'kernels_ms_total is the device share', **extra}
    if dist.rank == 0 and dist.world == 1 and not args.no_cpu:
        from oracle import prepare_oracle
        ns = 200000                                   # the loop costs ~40 us per halo
        hs = {k: v[:ns] for k, v in halos.items()}
        npart_s = int(hs['npstartA'][-1] + hs['npoutA'][-1])
        ps = {k: v[:npart_s] for k, v in parts.items()}
        np.random.seed(5)
        t1 = time.perf_counter()
        prepare_oracle.prepare_slab_core(hs, ps, Mpart, h, True, want_ranks=True, want_AB=True, Lbox=header['BoxSize'])
        tc = time.perf_counter() - t1
        out['cpu_baseline'] = {'value': ns / tc, 'unit': 'halos/s', 'cores': 1, 'kind': 'port',
                               'sample': f'the first {ns} halos ({npart_s} particles) of the same slab through the oracle\'s NumPy '
                                         f'restatement of the reference loop, {tc:.1f} s'}
    return out
