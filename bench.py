#!/usr/bin/env python3
"""bench.py - headline benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload hod|pk] [--nmesh M] [--no-cpu]

Metric (BASELINE.json): halos/sec of AbacusHOD population.  Workload at N=1 = BASELINE config 2:
10^7 synthetic halos + 10^7 subsample particles (seed 600), the LRG HOD of the reference's
tests/abacus_hod.yaml:31-47, rsd=True.  A step = one full populate of the resident catalog (decide centrals,
decide satellites, scan, ordered emission); inputs are in HBM before the timed region and the galaxy catalog
stays in HBM (device-resident rate; the PCIe-inclusive rate is quoted in DESIGN.md).  The second part of the
metric - wall-clock of the TSC + FFT + binning P(k) - is reported in the "pk" object of the same JSON line
(`--workload pk` makes it the headline instead).

N > 1: one process per GPU.  `python bench.py --gpus N` starts the N rank processes itself (abacusutils_amd/launch.py:
the parent never touches the GPU; fresh children with RANK / LOCAL_RANK / WORLD_SIZE set; a stuck or crashed rank is
ended by PID and the line is still printed, with an `error` field); under `python -m torch.distributed.run` every rank
process starts only its own child.  Collectives are RCCL through the C ABI (abacusutils_amd/comm.py) - no torch in any
of these processes.  Two legs per N: the headline, every rank populating its own 10^7-halo slab catalogue (weak scaling,
no data-path collective - halos are independent; the reference's unit is the slab chunk,
abacusnbody/hod/abacus_hod.py:301-312), and `pk_slab`: ONE nmesh^3 TSC + FFT P(k) decomposed over the N GPUs (strong
scaling: ghost exchange, all-to-all pencil transpose, histogram all-reduce).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


from bench_pk import pmc_traffic  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='hod', choices=['hod', 'pk'])
    ap.add_argument('--nhalo', type=int, default=10_000_000)
    ap.add_argument('--npart', type=int, default=10_000_000)
    ap.add_argument('--nmesh', type=int, default=2048)
    ap.add_argument('--npk', type=int, default=100_000_000, help='particles for the P(k) workload')
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--no-pk', action='store_true', help='skip the secondary P(k) / pairs / catalogue measurements')
    ap.add_argument('--no-hod-extra', action='store_true', help='skip the hod_multi and hod_large legs')
    ap.add_argument('--no-slab', action='store_true', help='N > 1: skip the slab-decomposed P(k) leg (RCCL all-to-all)')
    ap.add_argument('--slab-timeout', type=float, default=240.0, help='seconds before the slab leg is abandoned')
    ap.add_argument('--slab-presorted', action='store_true',
                    help='slab leg: particles generated inside every rank\'s own folded slabs (no routing in the timed step)')
    ap.add_argument('--hod-timeout', type=float, default=270.0, help='N > 1: seconds before the headline leg is abandoned')
    ap.add_argument('--option', action='append', default=[], metavar='NAME=VALUE',
                    help='diagnostic option of the library (abacus_set_option), e.g. hod_nocls=1: A/B timing of a comparator path')
    ap.add_argument('--leg', default=None, choices=['hod', 'pk_slab'], help='internal: run one leg as a rank process')
    return ap.parse_args()


def filter_bytes_per_object(tracers, enable_ranks, two_stage=True):
    """algorithmic bytes the rejection filter streams per halo / per particle, in the layout it actually reads
    (DESIGN.md section 4).  Key filter: ONE packed 16-bit key per object (mass bin + a 9-bit code of a lower bound of
    random / weight, built at staging and after a reseed) - 2 B read per object for ANY HOD; mixes with ELG / QSO also zero
    the 1-B keep mask in the same kernel (LRG alone: hod_exact un-keeps what the previous populate kept instead).  The
    float64 columns are read only for the table's survivors.  One-stage fallback (parameter sets the envelope table cannot
    bound): float32 shadow columns, 12 B + deltac, fenv (+ shear) per halo when weighted, 12 B + four rank columns per
    particle."""
    if two_stage:
        m = 1.0 if ('ELG' in tracers or 'QSO' in tracers) else 0.0
        return 2.0 + m, 2.0 + m
    env = any(t.get('Acent', 0) != 0 or t.get('Bcent', 0) != 0 for t in tracers.values())
    shear = 'ELG' in tracers and tracers['ELG'].get('Ccent', 0) != 0
    return 12.0 + (8.0 if env else 0.0) + (4.0 if shear else 0.0), 12.0 + (16.0 if enable_ranks else 0.0)


def measure_hod(args, dist, nh, npart, tracers, enable_ranks, with_ranks, label, extras=True):
    """one HOD measurement: stage a synthetic catalogue (seed 600 + rank), populate `steps` times with everything
    resident in HBM; returns (result dict, inputs for the CPU baseline)"""
    import numpy as np
    from abacusutils_amd import _lib, synth
    from abacusutils_amd.hod import GRAND_HOD as G

    hd, pd, params = synth.synth_hod_inputs(nh, npart, seed=600 + dist.rank, with_ranks=with_ranks)
    p = G.marshal_params(tracers, params, enable_ranks, True)
    st = G.StagedCatalog(hd, pd)  # H2D once: inputs resident in HBM from here on

    # first populate: builds keys, shadows, records, column ranges (one-off staging work outside the timed region: whole
    # durations -> stage_ms) and sizes the catalog buffers; its launches also carry one-time code-object loads
    _lib.profile_reset()
    _lib.profile_enable(True)
    st.populate(p)
    _lib.profile_enable(False)
    stage_ms = sum(ms for k, (ms, n) in _lib.profile_get().items()
                   if n and k in ('hod_shadow', 'hod_build_recs', 'hod_refresh_recs', 'hod_build_keys', 'hod_minmax', 'hod_check_pinds'))
    _lib.profile_reset()
    _lib.profile_enable(True)
    for _ in range(max(args.warmup, 1)):  # steady-state kernel durations
        st.populate(p)
    counts = st.wait_counts()
    ngal = int(np.sum(counts))
    _lib.profile_enable(False)
    wprof = _lib.profile_get()
    # one-off work that only happens from the second populate on (the mass-sorted key index of the sparse mixes) is staging too
    index_names = ('hod_index_keys', 'hod_index_sort', 'hod_index_last')
    stage_ms += sum(ms for k, (ms, n) in wprof.items() if n and k in index_names)
    STEP = ('hod_filter', 'hod_deal', 'hod_exact', 'hod_emit')
    warm = {k: ms / n for k, (ms, n) in wprof.items() if n and k not in index_names}
    launches = {k: n / max(args.warmup, 1) for k, (ms, n) in wprof.items() if n and k not in index_names}
    if 'hod_deal' in warm and 'hod_filter' in warm:   # the warm-up's first populate still streamed the keys
        del warm['hod_filter'], launches['hod_filter']
        launches['hod_deal'] = 1.0
    # the kernels of a step lie within a few microseconds of each other at C2 and the warm-up averages carry first launches
    # (the key index is built during it): the dominant kernel is picked from a few more, steady, untimed populates
    _lib.profile_reset()
    _lib.profile_enable(True)
    for _ in range(20):
        st.populate(p)
    st.wait_counts()
    _lib.profile_enable(False)
    warm.update({k: ms / n for k, (ms, n) in _lib.profile_get().items() if n and k in warm})
    cand = st.candidates()

    # timed region: HIP events only around the dominant kernel (a pair of event records per launch costs a few
    # microseconds, which is not negligible against a 0.1-ms step); the other kernels' durations are the warm-up's.
    # The dominant kernel is the one that takes the LONGEST per step (duration x launches, from the 20 steady populates above);
    # the pick by bytes among the near-equal kernels of the C2 index path (rounds 3 - 4) is reported beside it as `bytes_pick`
    bh0, bp0 = filter_bytes_per_object(tracers, enable_ranks)
    nc0 = float(cand[0] + cand[1])
    dom_bytes = {'hod_filter': bh0 * nh + bp0 * npart, 'hod_deal': 6.0 * nc0, 'hod_exact': 130.0 * nc0, 'hod_emit': 192.0 * ngal}
    per_step = {k: warm[k] * launches.get(k, 1.0) for k in warm if k in STEP}
    tmax = max(per_step.values(), default=0.0)
    # name-agnostic: the kernel that takes longest per step; kernels within 15 % of the longest are a tie at this clock (the
    # kernels of the C2 index path lie within two microseconds of each other and trade places from run to run), which goes to
    # the one that moves the most algorithmic bytes.  Every kernel of the step is reported in `roofline.alternatives`
    near = [k for k in per_step if per_step[k] >= 0.85 * tmax]
    dom_name = max(near, key=lambda k: dom_bytes.get(k, 0.0), default=None)
    bytes_pick = max((k for k in per_step if per_step[k] >= 0.5 * tmax), key=lambda k: dom_bytes.get(k, 0.0), default=None)
    _lib.profile_reset()
    _lib.profile_select(dom_name)
    _lib.profile_enable(True)
    dist.barrier()
    _lib.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st.populate_async(p)
    st.wait_counts()
    _lib.sync()
    t1 = time.perf_counter()     # this rank's K steps, device idle again; the closing barrier's own latency (an all-reduce
    dist.barrier()               # round trip, or a file poll on the fallback transport) is not part of the steps
    dt = dist.max(t1 - t0)
    _lib.profile_enable(False)
    _lib.profile_select(None)
    prof = _lib.profile_get()

    total_halos = dist.sum(float(nh)) * args.steps
    out = {
        'metric': 'halos/sec HOD populate',
        'value': total_halos / dt,
        'unit': 'halos/s',
        'n_gpus': dist.world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f64',
        'data': 'synthetic',
        'config': {'workload': label, 'n_halo': nh, 'n_part': npart, 'n_gal': ngal, 'tracers': list(tracers),
                   'enable_ranks': bool(enable_ranks)},
    }
    if extras:
        # MCMC pattern: host needs the counts every step (one sync per step)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            st.populate(p)
        out['ms_per_step_host_sync'] = (time.perf_counter() - t1) / args.steps * 1e3
        # PCIe-inclusive: catalog copied back to NumPy every step (what run_hod returns)
        st.populate(p)
        for tr in tracers:
            st.fetch(tr)   # first transfer: one-time runtime set-up of the 2-D copy path
        t2 = time.perf_counter()
        for _ in range(10):
            st.populate(p)
            for tr in tracers:
                st.fetch(tr)
        out['ms_per_step_with_d2h'] = (time.perf_counter() - t2) / 10 * 1e3
        # run_hod(reseed=s) every step (hod/abacus_hod.py:775-839): new hrandoms / hveldev / prandoms from the device generator,
        # then the keys and the random fields of the packed records are rebuilt before the populate - what `stage_ms` hides
        try:
            st.reseed(1, hsigma3d=hd['hsigma3d'])
            st.populate(p)
            _lib.profile_reset()
            _lib.profile_enable(True)
            for q in range(3):
                st.reseed(100 + q)
                st.populate(p)
            _lib.profile_enable(False)
            rk = {k: round(ms / n, 4) for k, (ms, n) in _lib.profile_get().items() if n}
            _lib.sync()
            t3 = time.perf_counter()
            for q in range(10):
                st.reseed(200 + q)
                st.populate(p)
            _lib.sync()
            out['reseed'] = {'ms_per_step': (time.perf_counter() - t3) / 10 * 1e3, 'kernels_ms': rk,
                             'note': 'st.reseed(seed) + populate per step: Philox draws for every halo and particle, keys and '
                                     'record fields rebuilt, keys streamed (the mass-sorted index belongs to unchanged keys)'}
        except Exception as e:   # a secondary measurement must not take the headline down
            out['reseed'] = {'error': repr(e)}
    # roofline of the dominant kernel: algorithmic bytes of the layout it streams / HIP-event duration
    bh, bp = filter_bytes_per_object(tracers, enable_ranks)
    step_bytes = bh * nh + bp * npart + 152.0 * ngal     # + gather 88 B and write 64 B per galaxy
    elg = 'ELG' in tracers
    survey_bytes = (40.0 + (8.0 if elg else 0.0)) * nh + (40.0 + (9.0 if elg else 0.0) + (32.0 if enable_ranks else 0.0)) * npart + 152.0 * ngal
    if 'hod_deal' in warm:
        step_bytes = 6.0 * (cand[0] + cand[1]) + 130.0 * (cand[0] + cand[1]) + 192.0 * ngal   # index path: no key stream
    kern = dict(warm)
    kern.update({k: (ms / n) for k, (ms, n) in prof.items() if n})
    out['kernels_ms'] = {k: round(v, 5) for k, v in kern.items()}   # per launch
    out['launches_per_step'] = launches
    out['filter_candidates'] = {'halos': cand[0], 'particles': cand[1]}
    out['galaxies'] = {'centrals': [int(c) for c in counts[:3]], 'satellites': [int(c) for c in counts[3:]]}
    out['stage_ms'] = round(stage_ms, 4)
    if dom_name:
        # algorithmic bytes of one launch of the dominant kernel, in the layout it reads: the filter
        # streams the keys (+ masks); hod_exact gathers one 128-B record line + a 2-B queue entry per candidate; hod_emit one
        # record line per galaxy and writes its 64 B; hod_deal reads a 4-B index and writes a 2-B queue entry per candidate
        ncand = float(cand[0] + cand[1])
        per_launch = {'hod_filter': bh * nh + bp * npart, 'hod_deal': 6.0 * ncand,
                      'hod_exact': 130.0 * ncand / max(launches.get('hod_exact', 1.0), 1.0), 'hod_emit': 192.0 * ngal}
        fbytes = per_launch[dom_name]
        ach = fbytes / (kern[dom_name] * 1e-3) / 1e9
        c2 = (nh, npart) == (10_000_000, 10_000_000) and list(tracers) == ['LRG']
        out['roofline'] = {'bound': 'hbm', 'kernel': dom_name, 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                           'frac': ach / HBM_PEAK_GBS,
                           'traffic': pmc_traffic('hod', dom_name) if c2 else None,
                           'algorithmic_bytes': fbytes,
                           # SURVEY.md 8d's yardstick for the WHOLE step - 40 B per halo and per particle streamed, 152 B per
                           # galaxy - beside the bytes this implementation really touches: a survey_frac above 1 says the step does
                           # not stream the catalogue at all (candidates are prefixes of a mass-sorted key index built at staging)
                           'survey_bytes': survey_bytes,
                           'survey_frac': survey_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                           'layout': {
                               'hod_filter': f'packed filter keys built at staging ({bh:.0f} B per halo + {bp:.0f} B per particle: a 2-B key = mass bin + a '
                                             '9-bit code of a lower bound of random / weight; + 1 B of keep mask zeroed for mixes with ELG / QSO); the float64 '
                                             'reference layout (SURVEY.md 8d: 40 B per object) is read only for the candidates',
                               'hod_deal': 'mass-sorted key index: 4-B index read + 2-B queue entry written per candidate',
                               'hod_exact': 'no key stream in the timed step (mass-sorted key index of the sparse mixes: the candidates are prefixes of '
                                            'the index, dealt to the tiles by hod_deal): the dominant kernel gathers one 128-B packed record line + a 2-B '
                                            'queue entry per candidate - a launch of a few hundred thousand dependent gathers, latency- not bandwidth-bound',
                               'hod_emit': 'one 128-B packed record line gathered and 64 B of columns written per galaxy'}[dom_name] +
                                         '; keys, index, float32 shadows and packed records are built once per catalogue (`stage_ms`, outside the timed '
                                         'region), keys and index again after a reseed',
                           'pick': 'the kernel that takes longest per step (duration x launches over 20 steady populates; among kernels within '
                                   '15 % of the longest - a tie at this clock - the one moving the most bytes); all of them in `alternatives`. '
                                   'Durations of this run: ' +
                                   ', '.join(f'{k} {per_step[k] * 1e3:.1f} us' for k in sorted(per_step, key=per_step.get, reverse=True)),
                           'alternatives': {k: {'us_per_step': round(per_step[k] * 1e3, 2), 'algorithmic_bytes': per_launch[k],
                                                'frac': per_launch[k] / (kern[k] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                                            for k in per_step if k in per_launch and kern.get(k)},
                           'bytes_pick': None if bytes_pick is None else {
                               'kernel': bytes_pick, 'frac': per_launch[bytes_pick] / (kern[bytes_pick] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               'note': 'the kernel moving the most bytes among those within 2x of the longest (rounds 3 - 4 quoted this one)'},
                           'whole_step_GBs': step_bytes / (dt / args.steps) / 1e9,
                           'whole_step_frac': step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS}
    st.free()
    return out, (hd, pd, params)


def bench_hod(args, dist):
    """headline: BASELINE config 2 (LRG, 1e7 + 1e7); at N = 1 also `hod_multi` (BASELINE config 5's HOD: LRG + ELG + QSO
    with assembly bias, ranks, conformity on the same catalogue) and `hod_large` (4e7 + 4e7: a 1-GB filter stream, four
    times the 256-MiB Infinity Cache)"""
    from abacusutils_amd import synth
    nh, npart = args.nhalo, args.npart
    out, inputs = measure_hod(args, dist, nh, npart, {'LRG': synth.LRG_PARAMS}, False, False,
                              f'C2: {nh:.0e} synthetic halos + {npart:.0e} subsample particles per GPU (seed 600+rank), '
                              'LRG HOD of tests/abacus_hod.yaml:31-47, rsd=True, catalog resident in HBM')
    single = dist.rank == 0 and dist.world == 1
    if single and not args.no_cpu:   # CPU baseline: rank 0 at N = 1 only
        out['cpu_baseline'] = cpu_baseline_hod(*inputs, {'LRG': synth.LRG_PARAMS}, nh)
    del inputs
    if single and not args.no_hod_extra:
        for key, fn in (
            ('hod_multi', lambda: measure_hod(
                args, dist, nh, npart, synth.PRODUCTION_TRACERS, True, True,
                f'C5 HOD: LRG + ELG + QSO (synth.PRODUCTION_TRACERS: tests/abacus_hod.yaml blocks with assembly bias, '
                f'rank modulation, velocity bias, ELG conformity) on the C2 catalogue ({nh:.0e} + {npart:.0e}, ranks staged)',
                extras=False)[0]),
            ('hod_large', lambda: measure_hod(
                args, dist, 4 * nh, 4 * npart, {'LRG': synth.LRG_PARAMS}, False, False,
                f'{4 * nh:.0e} halos + {4 * npart:.0e} particles, LRG: the C2 workload at four times the size '
                '(filter stream 0.96 GB >> 256 MiB Infinity Cache)', extras=False)[0]),
        ):
            try:
                out[key] = fn()
            except Exception as e:   # a secondary measurement must not take the headline down
                out[key] = {'error': repr(e)}
    return out


def bench_calls(args, dist):
    """what a CALLER of the drop-in surface sees, per call, at BASELINE config 2 (1e7 halos + 1e7 particles, LRG): the loop
    of the reference's scripts/hod/run_hod.py:40-73 - mutate a parameter, AbacusHOD.run_hod(), a clustering statistic"""
    import numpy as np
    from abacusutils_amd import synth
    from abacusutils_amd.hod.abacus_hod import AbacusHOD
    hod = dict(tracer_flags={'LRG': True, 'ELG': False, 'QSO': False}, want_ranks=False, want_AB=True, want_shear=False,
               want_rsd=True, LRG_params=synth.LRG_PARAMS, ELG_params=synth.ELG_PARAMS, QSO_params=synth.QSO_PARAMS)
    hd, pd, params = synth.synth_hod_inputs(args.nhalo, args.npart, seed=600)
    ball = AbacusHOD.from_arrays(hd, pd, params, hod)
    out = {'config': {'workload': f'AbacusHOD.run_hod() and clustering calls on the C2 catalogue ({args.nhalo:.0e} halos + '
                                  f'{args.npart:.0e} particles, LRG), one parameter changed per call'}}

    def loop(reps, body):
        body(0)
        t0 = time.perf_counter()
        for i in range(reps):
            body(i + 1)
        return (time.perf_counter() - t0) / reps * 1e3

    state = {}

    def run(i):
        ball.tracers['LRG']['logM_cut'] = 13.3 + 0.001 * (i % 5)
        state['m'] = ball.run_hod(ball.tracers, True, Nthread=16)

    for lazy, key in ((False, 'run_hod_call_ms'), (True, 'run_hod_call_lazy_ms')):
        ball.lazy_columns = lazy
        loop(5, run)
        out[key] = loop(100, run)
    out['galaxies'] = int(len(state['m']['LRG']['x']))
    out['run_hod_note'] = ('run_hod_call_ms: the eight columns returned as NumPy arrays every call (one device-to-host copy of '
                           '64 B per galaxy into recycled page-locked memory); run_hod_call_lazy_ms: AbacusHOD.lazy_columns = True, '
                           'columns left in HBM until read')
    ball.lazy_columns = True
    rp = np.geomspace(0.2, 30.0, 9)

    def run_wp(i):
        run(i)
        state['wp'] = ball.compute_wp(state['m'], rp, 30, 1)

    def run_pk(i):
        run(i)
        state['pk'] = ball.compute_power(state['m'], 32, 4, 0.5, False, poles=[0, 2], num_cells=512)

    def run_pk550(i):   # compute_power's own default mesh (hod/abacus_hod.py:1347) with the reference's default estimator of calc_power
        run(i)
        state['pk'] = ball.compute_power(state['m'], 32, 4, 0.5, False, poles=[0, 2], num_cells=550, compensated=True, interlaced=True)

    out['run_hod_plus_compute_wp_ms'] = loop(10, run_wp)
    out['run_hod_plus_compute_power_ms'] = loop(10, run_pk)
    try:
        from abacusutils_amd import _lib
        out['run_hod_plus_compute_power_550_interlaced_ms'] = loop(10, run_pk550)
        _lib.set_option('pk_noxbin_inter', 1)           # two x passes + spectrum_bin<INTER>: what round 4 ran
        out['run_hod_plus_compute_power_550_interlaced_unfused_ms'] = loop(10, run_pk550)
        _lib.set_option('pk_noxbin_inter', 0)
    except Exception as e:   # noqa: BLE001
        out['run_hod_plus_compute_power_550_interlaced_ms'] = repr(e)
    out['compute_note'] = 'lazy mock fed to compute_wp (8 log bins to 30 Mpc/h, pimax 30) / compute_power (512^3 TSC, 32 x 4 bins, poles 0, 2): galaxies never leave HBM'
    if ball._staged is not None:
        ball._staged.free()
    return out


def cpu_baseline_hod(hd, pd, params, tracers, nh, enable_ranks=False):
    """the oracle's compiled kernels (C + OpenMP restatement of the reference's two-pass chunked algorithm) on the host
    cores: arrays marshalled and outputs allocated outside the timed region (oracle.time_gen_gals), thread counts swept
    because the streaming passes stop scaling long before 256 threads; `value` is the best of the sweep"""
    from oracle import oracle
    import bench_pk
    cores, quota = bench_pk.cpu_share()
    # The GPU box of round 6 shows 256 logical CPUs in the affinity mask under a cgroup quota of 16 CPUs' worth of time per 100 ms:
    # a burst shorter than the period runs as wide as it likes until the period's budget is spent (this 10-ms pass still speeds up
    # to 128 threads: 128 x 11 ms = 1.4 of the 1.6 CPU-seconds), a team of 256 is throttled mid-pass (28 x slower).  The sweep
    # stops at half the logical CPUs when there is a quota; `value` is the best of it and the quota is stated beside it
    sweep = sorted({t for t in (8, 16, 32, 64, 128) if t <= cores} | ({cores} if not quota else set()))
    res = {}
    for t in sweep:
        os.environ['OMP_NUM_THREADS'] = str(t)
        tmin, tmean, _ = oracle.time_gen_gals(hd, pd, tracers, params, t, reps=3, enable_ranks=enable_ranks)
        res[t] = (tmin, tmean)
    best = min(res, key=lambda t: res[t][0])
    return {'value': nh / res[best][0], 'unit': 'halos/s', 'cores': best, 'kind': 'port',
            'sample': f'full workload ({nh} halos + particles), compiled kernels only (no NumPy marshalling / allocation in '
                      f'the timed region), min of 3 reps after 1 warm-up at each of {sweep} threads; best at {best}. '
                      'Published reference figure for context: 80 ms per tracer for a 2 Gpc/h box on 32 cores '
                      '(docs/hod.rst:13-15; not the same catalogue)',
            'ms_by_threads': {str(t): round(res[t][0] * 1e3, 2) for t in sweep},
            'host_cores': cores, 'cgroup_cpu_quota': quota, 'cpu_model': cpu_model()}


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def run_leg(args):
    """a rank process of an N > 1 run (started by `orchestrate`): RCCL communicator from the environment, one leg"""
    from abacusutils_amd.comm import Dist
    # binds GPU LOCAL_RANK, file rendezvous, ncclCommInitRank; raises without a HIP device.  The HOD leg has no data-path
    # collective (every rank populates its own shard): if the RCCL communicator cannot be created it still runs, with the
    # start barrier and the max over the ranks' timings through files, and says so in `rccl`
    from abacusutils_amd.comm import RcclJoinTimeout
    try:
        dist = Dist.from_env(allow_file_fallback=(args.leg == 'hod'))
    except RcclJoinTimeout as e:
        # a helper thread of this process is still inside ncclCommInitRank / the first barrier: no further library call, no
        # interpreter teardown under it - report and leave at once (the orchestrator prints the fallback line)
        print(f'BENCH-LEG-ERROR RcclJoinTimeout: {e}', file=sys.stderr, flush=True)
        os._exit(3)
    if args.leg == 'hod':
        out = bench_hod(args, dist)
    else:
        from bench_pk import bench_pk_slab
        out = bench_pk_slab(args, dist)
    if dist.comm is not None:
        info = dist.comm.info()
        # what EVERY rank saw, gathered through the communicator itself (one scalar all-reduce per rank: under torchrun the
        # orchestrator of rank 0 only ever sees its own child): ranks that took part, bytes each put on the links
        sent = [dist.sum(float(info.get('bytes_sent', 0)) if dist.rank == r else 0.0) for r in range(dist.world)]
        seen = int(round(dist.sum(1.0)))
        out['rccl'] = dict(info, ranks_seen=seen, bytes_sent_per_rank=[int(b) for b in sent])
        if dist.rccl_error:
            out['rccl']['error'] = dist.rccl_error
    if dist.rank == 0:
        print('BENCH-LEG ' + json.dumps(out), flush=True)
    dist.finish()


def orchestrate(args, under_launcher):
    """N > 1.  This process never touches the GPU: it starts rank children per leg and assembles the line."""
    from abacusutils_amd.launch import failure_summary, launch_ranks
    if under_launcher:      # one orchestrator per rank already exists (torchrun): each starts its own child
        world, ranks = int(os.environ['WORLD_SIZE']), [int(os.environ['RANK'])]
        key0 = f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}_{os.environ.get('TORCHELASTIC_RUN_ID', 'x')}"
    else:
        world, ranks = args.gpus, None
        key0 = f'{os.getpid()}_{int(time.time() * 1e3)}'
    env = {k: v for k, v in os.environ.items() if not k.startswith('TORCHELASTIC')}
    is_root = ranks is None or 0 in ranks
    common = ['--gpus', str(world), '--steps', str(args.steps), '--warmup', str(args.warmup), '--nhalo', str(args.nhalo),
              '--npart', str(args.npart), '--nmesh', str(args.nmesh), '--npk', str(args.npk), '--no-cpu']
    if args.slab_presorted:
        common.append('--slab-presorted')
    for o in args.option:
        common += ['--option', o]

    def leg(name, timeout):
        res = launch_ranks([sys.executable, os.path.abspath(__file__), '--leg', name] + common, world, ranks=ranks,
                           timeout=timeout, tag='BENCH-LEG', key=f'{key0}_{name}', env=env)
        ok = all(c == 0 for c in res['returncodes'].values())
        return (res['results'].get(0) if is_root else None), ok, (None if ok else failure_summary(res))

    class Terminated(Exception):
        pass

    def on_term(signum, frame):     # the launcher ends the surviving ranks when one fails: still print the line
        raise Terminated()

    import signal
    signal.signal(signal.SIGTERM, on_term)
    out, rc = None, 1
    fallback = {'metric': 'halos/sec HOD populate', 'value': None, 'unit': 'halos/s', 'n_gpus': world, 'steps': args.steps,
                'warmup': args.warmup, 'ms_per_step': None, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                'dtype': 'f64', 'data': 'synthetic', 'config': {'workload': 'C2 per GPU (not measured)'}}
    try:
        out, ok, err = leg('hod', args.hod_timeout)
        if out is None:
            out = dict(fallback)
        if err:
            out['error'] = err
        rc = 0 if ok else 1
        if not ok and not is_root:
            time.sleep(3.0)     # under a launcher that ends every rank at the first failure: rank 0 reports first
        if not args.no_slab and not args.no_pk:
            slab, sok, serr = leg('pk_slab', args.slab_timeout)
            slab = slab or {}
            if serr:
                slab['error'] = serr
            out['pk_slab'] = slab
    except Terminated:
        signal.signal(signal.SIGTERM, signal.SIG_IGN)
        out = out or dict(fallback)
        out.setdefault('error', 'terminated by the launcher before the leg finished (another rank failed first)')
        rc = 1
    if is_root:
        print(json.dumps(out), flush=True)
    return rc


def single(args):
    """N = 1: everything in this process, no communicator"""
    from abacusutils_amd import _lib
    from abacusutils_amd.comm import Dist
    dist = Dist(None)
    _lib.set_device(0)
    for kv in args.option:
        name, _, val = kv.partition('=')
        _lib.set_option(name, int(val or 1))
    if args.workload == 'hod':
        out = bench_hod(args, dist)
        if not args.no_pk:
            import bench_pk
            for key, fn in (('calls', lambda: bench_calls(args, dist)),
                            ('pk', lambda: bench_pk.bench_pk(args, dist, headline=False, cpu=False)),
                            # BASELINE config 3 itself (1024^3, 1e8 particles), with the CPU oracle timed on the same workload
                            ('pk_c3', lambda: bench_pk.bench_pk(args, dist, headline=False, nmesh=1024, variants=False)),
                            # a mixed-radix mesh (csrc/gfft.hip; the reference takes any size, compute_power defaults to 550)
                            ('pk_1536', lambda: bench_pk.bench_pk(args, dist, headline=False, nmesh=1536, cpu=False, variants=False)),
                            ('pairs', lambda: bench_pk.bench_pairs(args, dist)),
                            ('catalog', lambda: bench_pk.bench_catalog(args, dist)),
                            ('prepare', lambda: bench_pk.bench_prepare(args, dist))):
                try:                    # a secondary measurement must not take the headline down
                    out[key] = fn()
                except Exception as e:
                    out[key] = {'error': repr(e)}
            # the metric's own mesh (2048^3): ONE timed oracle calc_power on the same 1e8 particles when the host has the
            # memory for it (float32 mesh 34 GB + complex64 spectrum 34 GB + the raw power 17 GB); otherwise an estimate from the
            # measured config-3 call under a key of its own, so that nothing reads it as a measurement
            if not args.no_cpu and isinstance(out.get('pk'), dict) and 'error' not in out['pk']:
                try:
                    if args.nmesh != 1024 and bench_pk.host_memory_gb() >= bench_pk.cpu_pk_host_gb(args.nmesh, args.npk):
                        out['pk']['cpu_baseline'] = bench_pk.cpu_baseline_pk(2000.0, out['pk'].get('ms_per_step'), nmesh=args.nmesh,
                                                                              n=args.npk, reps=1)
                    else:
                        c3 = out['pk_c3']['cpu_baseline']
                        scale = 8.0 * 33.0 / 30.0      # mesh cells x log2(M): the transform dominates the CPU call
                        out['pk']['cpu_baseline_estimate'] = {
                            'value': c3['value'] / (33.0 / 30.0), 'unit': c3['unit'], 'cores': c3['cores'], 'kind': 'port', 'ms': c3['ms'] * scale,
                            'sample': f"NOT MEASURED at {args.nmesh}^3 (host memory {bench_pk.host_memory_gb():.0f} GB available, "
                                      f"{bench_pk.cpu_pk_host_gb(args.nmesh, args.npk):.0f} needed): the oracle's config-3 call of this run "
                                      f"({c3['ms']:.0f} ms at 1024^3, {c3['cores']} threads) scaled by cells x log2(cells) = {scale:.1f}"}
                except Exception as e:   # noqa: BLE001
                    out['pk']['cpu_baseline_error'] = repr(e)
    else:
        from bench_pk import bench_pk
        out = bench_pk(args, dist, headline=True)
    print(json.dumps(out), flush=True)
    return 0


def main():
    args = parse()
    if args.leg:
        return run_leg(args)
    under_launcher = 'RANK' in os.environ and int(os.environ.get('WORLD_SIZE', '1')) > 1
    if args.gpus > 1 or under_launcher:
        return orchestrate(args, under_launcher)
    return single(args)


if __name__ == '__main__':
    sys.exit(main() or 0)
