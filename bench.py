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

N > 1: one process per GPU (torch.distributed.run); every rank populates its own 10^7-halo slab catalog
(weak scaling, no data-path collective - halos are independent).
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
    ap.add_argument('--no-pk', action='store_true', help='skip the secondary P(k) measurement')
    ap.add_argument('--no-slab', action='store_true', help='N > 1: skip the slab-decomposed P(k) leg (RCCL all-to-all)')
    ap.add_argument('--slab-timeout', type=float, default=150.0, help='seconds before one attempt of the slab leg is abandoned')
    return ap.parse_args()


class Dist:
    """barrier / max-reduce across ranks; torch.distributed only when WORLD_SIZE > 1"""

    def __init__(self, init_method=None):
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.td = None
        if self.world > 1:
            import torch  # noqa: F401  (imported BEFORE libabacus_hip.so so both share one HIP runtime)
            import torch.distributed as td
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            td.init_process_group(backend='gloo', rank=self.rank, world_size=self.world, init_method=init_method)
            self.td = td

    def barrier(self):
        if self.td:
            self.td.barrier()

    def max(self, x):
        if not self.td:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.td.all_reduce(t, op=self.td.ReduceOp.MAX)
        return float(t[0])

    def sum(self, x):
        if not self.td:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.td.all_reduce(t, op=self.td.ReduceOp.SUM)
        return float(t[0])

    def broadcast_int(self, x):
        if not self.td:
            return int(x)
        import torch
        t = torch.tensor([int(x)], dtype=torch.int64)
        self.td.broadcast(t, src=0)
        return int(t[0])

    def finish(self):
        if self.td:
            self.td.destroy_process_group()


def bench_hod(args, dist):
    import numpy as np
    from abacusutils_amd import _lib, synth
    from abacusutils_amd.hod import GRAND_HOD as G

    nh, npart = args.nhalo, args.npart
    hd, pd, params = synth.synth_hod_inputs(nh, npart, seed=600 + dist.rank)
    tracers = {'LRG': synth.LRG_PARAMS}
    p = G.marshal_params(tracers, params, False, True)
    st = G.StagedCatalog(hd, pd)  # H2D once: inputs resident in HBM from here on

    _lib.profile_reset()
    _lib.profile_enable(True)
    for _ in range(max(args.warmup, 1)):  # also sizes the catalog buffers
        st.populate(p)
    counts = st.wait_counts()
    ngal = int(counts[0] + counts[3])
    _lib.profile_enable(False)
    warm = {k: ms / n for k, (ms, n) in _lib.profile_get().items() if n}

    # timed region: HIP events only around the dominant kernel (a pair of event records per launch costs a few
    # microseconds, which is not negligible against a 0.2-ms step); the other kernels' durations are the warm-up's
    dom_name = max((k for k in warm if k.startswith('hod_filter')), key=lambda k: warm[k], default=None)
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
    dist.barrier()
    dt = dist.max(time.perf_counter() - t0)
    _lib.profile_enable(False)
    _lib.profile_select(None)
    prof = _lib.profile_get()

    # MCMC pattern: host needs the counts every step (one sync per step)
    t1 = time.perf_counter()
    for _ in range(args.steps):
        st.populate(p)
    dt_sync = time.perf_counter() - t1

    # PCIe-inclusive: catalog copied back to NumPy every step (what run_hod returns)
    st.populate(p)
    st.fetch('LRG')   # first transfer: one-time runtime set-up of the 2-D copy path
    t2 = time.perf_counter()
    for _ in range(10):
        st.populate(p)
        st.fetch('LRG')
    dt_fetch = (time.perf_counter() - t2) / 10

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
        'config': {
            'workload': f'C2: {nh:.0e} synthetic halos + {npart:.0e} subsample particles per GPU (seed 600+rank), '
                        'LRG HOD of tests/abacus_hod.yaml:31-47, rsd=True, catalog resident in HBM',
            'n_halo': nh, 'n_part': npart, 'n_gal': ngal, 'tracers': ['LRG'],
        },
        'ms_per_step_host_sync': dt_sync / args.steps * 1e3,
        'ms_per_step_with_d2h': dt_fetch * 1e3,
    }
    # roofline of the dominant kernel: algorithmic bytes / HIP-event duration
    alg_bytes = {
        # one fused launch over the central and the satellite tiles: mass, multis, randoms per halo and hmass, weights,
        # randoms per particle, streamed from the float32 SHADOW columns the library keeps for a catalogue it owns
        # (12 B per object; the float64 originals would be 24 B, SURVEY's 40 B count deltac / fenv, which the LRG HOD of
        # the test yaml multiplies by Acent = Bcent = 0 and the filter does not read) -- the smallest, honest numerator
        'hod_filter': 12.0 * nh + 12.0 * npart,
        'hod_emit': 1.0 * (nh + npart) + 152.0 * ngal,   # mask + gather 88 B + write 64 B per galaxy
    }
    kern = dict(warm)
    kern.update({k: (ms / n) for k, (ms, n) in prof.items() if n})
    out['kernels_ms'] = {k: round(v, 5) for k, v in kern.items()}
    dom = max((k for k in kern if k in alg_bytes), key=lambda k: kern[k], default=None)
    if dom:
        ach = alg_bytes[dom] / (kern[dom] * 1e-3) / 1e9
        out['roofline'] = {'bound': 'hbm', 'kernel': dom, 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                           'frac': ach / HBM_PEAK_GBS,
                           'traffic': pmc_traffic('hod', dom) if (nh, npart) == (10_000_000, 10_000_000) else None,
                           'algorithmic_bytes': alg_bytes[dom],
                           'whole_step_GBs': (12.0 * nh + 12.0 * npart + 152.0 * ngal) / (dt / args.steps) / 1e9,
                           'whole_step_frac': (12.0 * nh + 12.0 * npart + 152.0 * ngal) / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                           # the same times in units of the float64 columns the reference streams (24 B per object):
                           # what a kernel reading the reference layout would need to sustain - NOT bytes this path moves
                           'reference_layout_GBs': (24.0 * nh + 24.0 * npart) / (kern[dom] * 1e-3) / 1e9,
                           'reference_layout_whole_step_GBs': (24.0 * nh + 24.0 * npart + 152.0 * ngal) / (dt / args.steps) / 1e9}
    st.free()
    if dist.rank == 0 and dist.world == 1 and not args.no_cpu:   # CPU baseline: rank 0 at N = 1 only
        out['cpu_baseline'] = cpu_baseline_hod(hd, pd, params, tracers, nh)
    return out


def cpu_baseline_hod(hd, pd, params, tracers, nh):
    """the oracle (C + OpenMP port of the reference's two-pass chunked algorithm) on the host cores"""
    from oracle import oracle
    cores = len(os.sched_getaffinity(0))
    os.environ['OMP_NUM_THREADS'] = str(cores)
    oracle.gen_gal_cat(hd, pd, tracers, params, Nthread=cores)  # warm-up
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        oracle.gen_gal_cat(hd, pd, tracers, params, Nthread=cores)
        ts.append(time.perf_counter() - t)
    return {'value': nh / min(ts), 'unit': 'halos/s', 'cores': cores, 'kind': 'port',
            'sample': f'full workload ({nh} halos + particles), min of 3 reps after 1 warm-up; '
                      f'mean {nh / (sum(ts) / 3):.3e} halos/s',
            'cpu_model': cpu_model()}


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def run_slab_children(args, dist, collectives):
    """every rank starts `bench_pk.py --slab-child` (own rendezvous port, chosen by rank 0), waits for it with a
    timeout and reports rank 0's result line; children are ended by exact PID on timeout"""
    import socket
    import subprocess
    port = 0
    if dist.rank == 0:
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
    port = dist.broadcast_int(port)
    env = {k: v for k, v in os.environ.items() if not k.startswith('TORCHELASTIC')}
    cmd = [sys.executable, os.path.join(REPO, 'bench_pk.py'), '--slab-child', '--store-port', str(port),
           '--collectives', collectives, '--nmesh', str(args.nmesh), '--npk', str(args.npk), '--steps', str(min(args.steps, 5))]
    res, ok = None, False
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.slab_timeout)
        for line in r.stdout.splitlines():
            if line.startswith('SLAB-RESULT '):
                res = json.loads(line[len('SLAB-RESULT '):])
        ok = r.returncode == 0
        if dist.rank == 0:
            if res is None:
                res = {'error': f'child exit code {r.returncode}: ' + r.stderr[-400:]}
            ok = ok and 'error' not in res
    except subprocess.TimeoutExpired:
        res = {'error': f'abandoned after {args.slab_timeout:.0f} s'}
    bad = int(dist.sum(0.0 if ok else 1.0))          # collective: every rank takes the same retry decision
    if bad and 'error' not in (res or {}):
        res = {'error': f'{bad} rank(s) failed'}
    return res or {}


def main():
    args = parse()
    dist = Dist()
    from abacusutils_amd import _lib
    ndev = max(_lib.device_count(), 1)
    _lib.set_device(dist.local_rank % ndev)
    if args.workload == 'hod':
        out = bench_hod(args, dist)
        if not args.no_pk:
            try:
                from bench_pk import bench_pk
                out['pk'] = bench_pk(args, dist, headline=False)
            except Exception as e:  # the secondary measurement must not take the headline down
                out['pk'] = {'error': repr(e)}
            try:
                from bench_pk import bench_pairs
                out['pairs'] = bench_pairs(args, dist)
            except Exception as e:
                out['pairs'] = {'error': repr(e)}
            try:
                from bench_pk import bench_catalog
                out['catalog'] = bench_catalog(args, dist)
            except Exception as e:
                out['catalog'] = {'error': repr(e)}
    else:
        from bench_pk import bench_pk
        out = bench_pk(args, dist, headline=True)
    if dist.world > 1 and not args.no_slab and not args.no_pk:
        # One nmesh^3 mesh decomposed over the N GPUs (BASELINE config 4): ghost exchange, all-to-all pencil transpose
        # and histogram all-reduce over RCCL.  Runs in a child process per rank: the headline above is complete, and a
        # fault or a stuck collective in this leg must not take the bench line down with it.
        res = run_slab_children(args, dist, 'device')
        if 'error' in res:
            first = res
            res = run_slab_children(args, dist, 'host')
            res['device_collectives_error'] = first['error']
        if dist.rank == 0:
            out['pk_slab'] = res
    if dist.rank == 0:
        print(json.dumps(out), flush=True)
    dist.finish()


if __name__ == '__main__':
    main()
