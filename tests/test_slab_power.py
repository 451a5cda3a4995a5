"""x-slab P(k) (abacusutils_amd/analysis/slab_power.py): host orchestration at world_size 1 and 2 (gloo) against the
oracle's calc_power on the union catalogue.  The CPU tests use the NumPy stand-in for the device side
(tests/slab_numpy_backend.py); the GPU tests run the HIP entry points (two ranks share the one GPU, gloo staging)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from abacusutils_amd.synth import synth_positions
from oracle import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
L = 500.0


def run_ranks(tmp_path, world, backend, port, flags=(), **kw):
    out = str(tmp_path / f'slab_{backend}_{world}')
    args = [f'--{k.replace("_", "-")}={v}' for k, v in kw.items()] + list(flags)
    worker = os.path.join(HERE, '_slab_worker.py')
    if world == 1 and not flags:
        cmd = [sys.executable, worker, '--backend', backend, '--out', out] + args
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}',
               '--master-addr', '127.0.0.1', '--master-port', str(port), worker, '--backend', backend, '--out', out] + args
    env = dict(os.environ, OMP_NUM_THREADS='2')
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return [np.load(f'{out}.rank{k}.npz') for k in range(world)]


def reference(nmesh=64, n=20000, interlaced=1, compensated=1, cross=0, kbins=16):
    pos = synth_positions(n, L, seed=11)
    w = np.random.default_rng(5).random(n, dtype=np.float32) + np.float32(0.5)
    kw = dict(kbins=kbins, mubins=4, paste='TSC', nmesh=nmesh, compensated=bool(compensated), interlaced=bool(interlaced),
              poles=[0, 2, 4], nthread=2, accum64=True)
    if cross:
        kw['pos2'] = synth_positions(n // 2, L, seed=12)
    return oracle.calc_power(pos, L, w=w, **kw)


def check(res, ref, world, n):
    assert sum(int(r['n_local']) for r in res) == n
    for r in res:
        np.testing.assert_array_equal(r['N_mode'], ref['N_mode'])
        np.testing.assert_array_equal(r['N_mode_poles'], ref['N_mode_poles'])
        scale = np.abs(ref['power']).max()
        np.testing.assert_allclose(r['power'], ref['power'], rtol=1e-5, atol=1e-5 * scale)
        np.testing.assert_allclose(r['poles'], ref['poles'], rtol=1e-5, atol=1e-5 * scale)
        np.testing.assert_allclose(r['k_avg'], ref['k_avg'], rtol=1e-6)
    for r in res[1:]:                          # every rank holds the same all-reduced result
        np.testing.assert_array_equal(r['power'], res[0]['power'])


@pytest.mark.parametrize('world', [1, 2])
def test_slab_orchestration_cpu(tmp_path, world):
    res = run_ranks(tmp_path, world, 'numpy', 29611 + world)
    check(res, reference(), world, 20000)


def test_slab_cross_cpu(tmp_path):
    res = run_ranks(tmp_path, 2, 'numpy', 29621, cross=1, interlaced=0)
    check(res, reference(cross=1, interlaced=0), 2, 20000)


@pytest.mark.gpu
@pytest.mark.parametrize('world,interlaced,compensated,cross', [(1, 1, 1, 0), (2, 1, 1, 0), (2, 0, 0, 1), (4, 1, 1, 1)])
def test_slab_hip(tmp_path, world, interlaced, compensated, cross):
    res = run_ranks(tmp_path, world, 'hip', 29631 + world, interlaced=interlaced, compensated=compensated, cross=cross)
    check(res, reference(interlaced=interlaced, compensated=compensated, cross=cross), world, 20000)


@pytest.mark.gpu
@pytest.mark.parametrize('world,interlaced,cross', [(1, 0, 0), (2, 1, 0), (4, 0, 1), (4, 0, 0)])
def test_slab_hip_fused_form_small(tmp_path, world, interlaced, cross):
    """the FUSED slab transform (y stage inside the z pass, x stage inside the unpack, n/2-point column passes, permuted
    rows undone by the binning) at a mesh the CPU oracle can check: nmesh 256 with `fft_fuse_small`, 1 / 2 / 4 ranks"""
    res = run_ranks(tmp_path, world, 'hip', 29651 + world, flags=('--option', 'fft_fuse_small=1'), nmesh=256, kbins=40,
                    interlaced=interlaced, compensated=1, cross=cross)
    check(res, reference(nmesh=256, interlaced=interlaced, compensated=1, cross=cross, kbins=40), world, 20000)


@pytest.mark.gpu
@pytest.mark.parametrize('world,compensated', [(2, 0), (4, 1)])
def test_slab_hip_fused_last_pass_on_y_slabs(tmp_path, world, compensated):
    """nmesh 1024, auto power of one non-interlaced field: after the pencil transpose every rank runs the last x pass fused
    with the binning on its y-slab (abacus_slab_xbin_dev, y0 > 0 on ranks > 0, N_mode / k_avg from the cached geometry on
    rank 0 only) - against the single-GPU calc_power on the same catalogue"""
    from abacusutils_amd.analysis.power_spectrum import calc_power
    res = run_ranks(tmp_path, world, 'hip', 29661 + world, nmesh=1024, kbins=128, interlaced=0, compensated=compensated)
    pos = synth_positions(20000, L, seed=11)
    w = np.random.default_rng(5).random(20000, dtype=np.float32) + np.float32(0.5)
    ref = calc_power(pos, L, kbins=128, mubins=4, paste='TSC', nmesh=1024, compensated=bool(compensated), interlaced=False,
                     poles=[0, 2, 4], w=w)
    for r in res:
        np.testing.assert_array_equal(r['N_mode'], np.asarray(ref['N_mode']))
        scale = np.abs(np.asarray(ref['power'])).max()
        np.testing.assert_allclose(r['power'], np.asarray(ref['power']), rtol=1e-5, atol=1e-6 * scale)
        np.testing.assert_allclose(r['poles'], np.asarray(ref['poles']), rtol=1e-5, atol=1e-6 * scale)
        np.testing.assert_allclose(r['k_avg'], np.asarray(ref['k_avg']), rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('kbins,mubins', [(32, 5), (900, 7)])
def test_slab_hip_matches_single_gpu_path(tmp_path, kbins, mubins):
    """world=1 slab path against the product's own calc_power at a larger mesh; 900 x 7 bins: more than the LDS histogram of
    one binning launch holds - the raw sums the ranks reduce are assembled from several passes over runs of k bins"""
    from abacusutils_amd.analysis import power_spectrum as ps
    from abacusutils_amd.analysis import slab_power as sp
    n, nmesh = 400000, 128
    pos = synth_positions(n, L, seed=21, clustered=True)
    kw = dict(kbins=kbins, mubins=mubins, paste='TSC', nmesh=nmesh, compensated=True, interlaced=True, poles=[0, 2, 4])
    a = ps.calc_power(pos.copy(), L, **kw)
    b = sp.calc_power_slab(pos.copy(), L, **kw)
    np.testing.assert_array_equal(a['N_mode'], b['N_mode'])
    np.testing.assert_allclose(b['power'], a['power'], rtol=2e-5, atol=1e-6 * np.abs(a['power']).max())
    np.testing.assert_allclose(b['poles'], a['poles'], rtol=2e-5, atol=1e-6 * np.abs(a['power']).max())


@pytest.mark.gpu
@pytest.mark.parametrize('cross', [0, 1])
def test_slab_hip_rccl_single_rank(tmp_path, cross):
    """the product transport on the one GPU of the test box: a one-rank RCCL communicator created through the C ABI (file
    rendezvous, ncclCommInitRank), with the device-side particle routing, ring exchange, chunk-wise all-to-all on the
    communicator's stream, join and the histogram all-reduce all executing; no torch in the process"""
    res = run_ranks(tmp_path, 1, 'hip', 29641, flags=('--rccl',), cross=cross, interlaced=1 - cross)
    check(res, reference(cross=cross, interlaced=1 - cross), 1, 20000)


def test_slab_forced_collectives_gloo_cpu(tmp_path):
    res = run_ranks(tmp_path, 1, 'numpy', 29642, flags=('--force-collectives',))
    check(res, reference(), 1, 20000)


@pytest.mark.gpu
@pytest.mark.parametrize('world,rank,nmesh,n', [(4, 1, 256, 9_000_000), (8, 7, 512, 17_000_000), (2, 0, 128, 4_400_000), (4, 3, 64, 5000)])
def test_two_window_deposit_of_a_rank_equals_the_planes_of_the_full_mesh(world, rank, nmesh, n):
    """abacus_slab_deposit_dev with two windows (the folded slab pair of rank `rank` of `world`, ghosts included) on the particles
    that rank owns (>= 2e6 of them: the multisplit list build with the slab fast path, and once with `tsc_noslabfast`; 5000:
    the small-n path): every window plane equals the same plane of the whole-mesh deposit of those particles - bit for bit in the owned
    planes' interior, and wherever both windows hold a plane the first one has the deposits and the second none"""
    import ctypes as C

    from abacusutils_amd import _lib
    from abacusutils_amd.analysis import slab_power as sp
    box, G = 700.0, sp.GHOST
    h = nmesh // (2 * world)
    rng = np.random.default_rng(world * 100 + rank)
    pos = (rng.random((n, 3), dtype=np.float32) * np.float32(box)).astype(np.float32)
    mine = sp.slab_owner(pos[:, 0], box, world, True) == rank
    pos = np.ascontiguousarray(pos[mine])
    w = rng.uniform(0.5, 1.5, len(pos)).astype(np.float32)
    be = sp.HipSlabBackend()
    pitch = be.pitch(nmesh)
    plane = nmesh * pitch
    win = h + 2 * G
    xa, xb = (rank * h - G) % nmesh, (rank * h + nmesh // 2 - G) % nmesh
    full = sp.HipBuf(nmesh * plane)
    be.deposit(be.upload_particles(pos, w), full, nmesh, 0, nmesh, box, 0.0, 1.0, 0, sub=0.0)
    F = full.get(0, nmesh * plane).reshape(nmesh, nmesh, pitch)[:, :, :nmesh]
    for noslabfast in (0, 1):
        _lib.set_option('tsc_noslabfast', noslabfast)
        two = sp.HipBuf(2 * win * plane)
        be.deposit(be.upload_particles(pos, w), two, nmesh, xa, win, box, 0.0, 1.0, 0, sub=0.0, xoff2=xb)
        T = two.get(0, 2 * win * plane).reshape(2 * win, nmesh, pitch)[:, :, :nmesh]
        _lib.set_option('tsc_noslabfast', 0)
        seen = set()
        for k, x0 in enumerate((xa, xb)):
            for i in range(win):
                gp = (x0 + i) % nmesh
                want = F[gp] if gp not in seen else np.zeros_like(F[gp])
                seen.add(gp)
                np.testing.assert_array_equal(T[k * win + i], want, err_msg=f'window {k} plane {i} (global {gp}), noslabfast={noslabfast}')
        two.free()
    full.free()


@pytest.mark.gpu
@pytest.mark.parametrize('world,rank', [(8, 7), (8, 0), (4, 2)])
def test_two_window_deposit_takes_the_block_record_lists(world, rank):
    """unweighted float32 TSC into a rank's two windows, the buffer padded to whole 16-plane tiles as calc_power_slab allocates it:
    the deposit runs the third-generation list build of the single-GPU path on the windows' local planes (csrc/tsc_lines3.hpp,
    L3Win; asserted from the profiler) and every window plane equals the same plane of the whole-mesh deposit of those
    particles - to the resolution of the per-tile fixed-point sums (the two meshes tile differently); rank W - 1 / rank 0: the
    windows that wrap around the box"""
    from abacusutils_amd import _lib
    from abacusutils_amd.analysis import slab_power as sp
    box, G, nmesh, n = 700.0, sp.GHOST, 1024, 20_000_000
    h = nmesh // (2 * world)
    rng = np.random.default_rng(world * 10 + rank)
    pos = (rng.random((n, 3), dtype=np.float32) * np.float32(box)).astype(np.float32)
    pos = np.ascontiguousarray(pos[sp.slab_owner(pos[:, 0], box, world, True) == rank])
    assert len(pos) >= 2_000_000
    be = sp.HipSlabBackend()
    pitch = be.pitch(nmesh)
    plane = nmesh * pitch
    win = h + 2 * G
    xa, xb = (rank * h - G) % nmesh, (rank * h + nmesh // 2 - G) % nmesh
    full = sp.HipBuf(nmesh * plane)
    be.deposit(be.upload_particles(pos, None), full, nmesh, 0, nmesh, box, 0.0, 1.0, 0, sub=0.0)
    F = full.get(0, nmesh * plane).reshape(nmesh, nmesh, pitch)[:, :, :nmesh]
    full.free()
    padded = -(-2 * win // 16) * 16
    two = sp.HipBuf(padded * plane)
    _lib.profile_reset()
    _lib.profile_enable(True)
    be.deposit(be.upload_particles(pos, None), two, nmesh, xa, win, box, 0.0, 1.0, 0, sub=0.0, xoff2=xb)
    _lib.profile_enable(False)
    prof = _lib.profile_get()
    assert 'tsc_lines_coarse' in prof and 'tsc_ms_coarse_scatter' not in prof, sorted(prof)
    T = two.get(0, padded * plane).reshape(padded, nmesh, pitch)[:, :, :nmesh]
    two.free()
    assert abs(float(T.sum(dtype='f8')) / len(pos) - 1) < 2e-6          # every owned particle's whole cloud is in the windows
    scale = float(F.max())
    for k, x0 in enumerate((xa, xb)):
        for i in range(win):
            np.testing.assert_allclose(T[k * win + i], F[(x0 + i) % nmesh], rtol=2e-6, atol=2e-6 * scale, err_msg=f'window {k} plane {i}')
    assert not T[2 * win:].any()                                         # the padding planes


@pytest.mark.gpu
@pytest.mark.parametrize('world,nmesh,comp', [(8, 1024, 0), (8, 256, 1), (4, 512, 1), (8, 2048, 1)])
def test_eight_ranks_as_threads_on_one_gpu(world, nmesh, comp):
    """the north-star rank count on the one GPU of the box: W ranks of calc_power_slab as THREADS (tests/thread_comm.py; the
    box allows six GPU processes), every rank with its own buffers, ghost ring over the 2 W slabs, all-to-all staged through
    the host, the fused last pass reading blocks of eight peers (h = nmesh / 16) - against the single-GPU calc_power on the
    union catalogue; 4 ranks of a 512^3 mesh: the plain three-pass form (unpack, x pass, spectrum_bin)"""
    from thread_comm import run_ranks

    from abacusutils_amd import _lib
    from abacusutils_amd.analysis import slab_power as sp
    from abacusutils_amd.analysis.power_spectrum import calc_power
    n = 600000 if nmesh < 2048 else 3_000_000       # 2048^3 on 8 ranks: BASELINE config 4's mesh, h = 128 planes per half
    pos = synth_positions(n, L, seed=31, clustered=True)
    w = np.random.default_rng(6).random(n, dtype=np.float32) + np.float32(0.5)
    kw = dict(kbins=64 if nmesh >= 256 else 12, mubins=4, paste='TSC', nmesh=nmesh, compensated=bool(comp), interlaced=False,
              poles=[0, 2, 4])
    _lib.set_option('fft_fuse_small', 1 if nmesh == 256 else 0)
    try:
        ref = calc_power(pos.copy(), L, w=w, **kw)

        def rank_fn(comm):
            mine = slice(comm.rank, None, comm.world)
            p1, w1 = sp.route_particles(pos[mine], w[mine], L, comm, fold=True)
            t = sp.calc_power_slab(p1, L, comm=comm, backend=sp.HipSlabBackend(), w=w1, **kw)
            return len(p1), {k: np.asarray(t[k]) for k in ('power', 'N_mode', 'poles', 'k_avg')}

        res = run_ranks(world, rank_fn)
    finally:
        _lib.set_option('fft_fuse_small', 0)
    assert sum(r[0] for r in res) == n
    for _, t in res:
        np.testing.assert_array_equal(t['N_mode'], np.asarray(ref['N_mode']))
        scale = np.abs(np.asarray(ref['power'])).max()
        np.testing.assert_allclose(t['power'], np.asarray(ref['power']), rtol=1e-5, atol=1e-6 * scale)
        np.testing.assert_allclose(t['poles'], np.asarray(ref['poles']), rtol=1e-5, atol=1e-6 * scale)
        np.testing.assert_allclose(t['k_avg'], np.asarray(ref['k_avg']), rtol=1e-6)
    for _, t in res[1:]:
        np.testing.assert_array_equal(t['power'], res[0][1]['power'])


@pytest.mark.gpu
@pytest.mark.parametrize('world,nmesh,kfrac', [(8, 1024, 1.0), (8, 2048, 1.0), (4, 1024, 0.6)])
def test_compact_transpose_sends_a_fifth_less(world, nmesh, kfrac):
    """the pencil transpose of a spectrum that only feeds a binning up to k_max leaves the columns beyond sqrt(k_max^2 - ky^2) at
    home (csrc/fft.hip slab_layout: rows packed with their live columns, 16-row groups): the same spectrum bit for bit as the
    regular layout (option slab_nocompact) - and as the single-GPU path to 1e-5 - with >= 20 % fewer floats through the
    all-to-all when the bins end at the Nyquist frequency (more when they end earlier)"""
    from thread_comm import run_ranks

    from abacusutils_amd import _lib
    from abacusutils_amd.analysis import slab_power as sp
    from abacusutils_amd.analysis.power_spectrum import calc_power
    n = 600000 if nmesh < 2048 else 2_500_000
    pos = synth_positions(n, L, seed=47, clustered=True)
    kw = dict(kbins=48, mubins=3, k_max=kfrac * np.pi * nmesh / L, paste='TSC', nmesh=nmesh, compensated=True, interlaced=False, poles=[0, 2, 4])
    ref = calc_power(pos.copy(), L, **kw)

    def rank_fn(comm):
        mine = slice(comm.rank, None, comm.world)
        p1, _ = sp.route_particles(pos[mine], None, L, comm, fold=True)
        t = sp.calc_power_slab(p1, L, comm=comm, backend=sp.HipSlabBackend(), **kw)
        return getattr(comm, 'floats_sent', 0), {k: np.asarray(t[k]) for k in ('power', 'N_mode', 'poles', 'k_avg')}

    out = {}
    for mode in ('compact', 'regular'):
        _lib.set_option('slab_nocompact', 1 if mode == 'regular' else 0)
        try:
            out[mode] = run_ranks(world, rank_fn)
        finally:
            _lib.set_option('slab_nocompact', 0)
    sent = {m: sum(r[0] for r in out[m]) for m in out}
    assert sent['compact'] <= (0.80 if kfrac == 1.0 else 0.5) * sent['regular'], sent
    for (_, a), (_, b) in zip(out['compact'], out['regular']):
        for k in ('power', 'N_mode', 'poles', 'k_avg'):
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    t = out['compact'][0][1]
    np.testing.assert_array_equal(t['N_mode'], np.asarray(ref['N_mode']))
    scale = np.abs(np.asarray(ref['power'])).max()
    np.testing.assert_allclose(t['power'], np.asarray(ref['power']), rtol=1e-5, atol=1e-6 * scale)
    np.testing.assert_allclose(t['poles'], np.asarray(ref['poles']), rtol=1e-5, atol=1e-6 * scale)


@pytest.mark.gpu
@pytest.mark.parametrize('world,nmesh,mode', [(8, 1024, 'cross'), (4, 2048, 'cross'), (1, 1024, 'cross'), (8, 1024, 'interlaced'),
                                              (1, 1024, 'interlaced'),    # (8, 2048, 'interlaced'): scripts/gpu_slab.sh (suite budget, r06)
                                              (8, 1024, 'interlaced_cross'), (1, 1024, 'interlaced_cross')])
def test_field_pairs_over_slabs_take_the_fused_last_pass(world, nmesh, mode):
    """calc_power_slab(pos, pos2=..., interlaced=False) - LRG x ELG of BASELINE config 5 - and calc_power_slab(pos,
    interlaced=True) - the reference's default mode: both fields of the pair stop after their y pass, each crosses the links in
    the compact layout into its own receive buffer, and ONE fused last pass bins Re(conj(a) b), or |a + a' exp(i pi m / n)|^2 / 4,
    from the pair (fft_x_bin2<.., INTER[, CROSS]>).  Against the single-GPU spectrum (1e-5, N_mode exact), bit-identical between
    the compact and the regular transpose, and no separate x pass / spectrum_bin launched"""
    from thread_comm import run_ranks

    from abacusutils_amd import _lib
    from abacusutils_amd.analysis import slab_power as sp
    from abacusutils_amd.analysis.power_spectrum import calc_power
    n = 500000 if nmesh < 2048 else 1_500_000
    pos = synth_positions(n, L, seed=61, clustered=True)
    pos2 = None
    if mode in ('cross', 'interlaced_cross'):
        pos2 = synth_positions(n // 2, L, seed=62, clustered=True)
        pos2[:n // 5] = pos[:n // 5]
    # interlaced_cross: calc_power's defaults with a second catalogue - four fields, four receive buffers, fft_x_bin2<.., QUAD>
    kw = dict(kbins=48, mubins=3, paste='TSC', nmesh=nmesh, compensated=True, interlaced=mode in ('interlaced', 'interlaced_cross'), poles=[0, 2, 4])
    ref = calc_power(pos.copy(), L, pos2=None if pos2 is None else pos2.copy(), **kw)

    def rank_fn(comm):
        mine = slice(comm.rank, None, comm.world)
        p1, _ = sp.route_particles(pos[mine], None, L, comm, fold=True)
        p2 = None if pos2 is None else sp.route_particles(pos2[mine], None, L, comm, fold=True)[0]
        t = sp.calc_power_slab(p1, L, comm=comm, backend=sp.HipSlabBackend(), pos2=p2, **kw)
        return getattr(comm, 'floats_sent', 0), {k: np.asarray(t[k]) for k in ('power', 'N_mode', 'poles', 'k_avg')}

    out = {}
    for mode in ('compact', 'regular'):
        _lib.set_option('slab_nocompact', 1 if mode == 'regular' else 0)
        _lib.profile_reset()
        _lib.profile_enable(True)
        try:
            out[mode] = run_ranks(world, rank_fn)
        finally:
            _lib.profile_enable(False)
            _lib.set_option('slab_nocompact', 0)
        prof = _lib.profile_get()
        assert 'fft_x_bin' in prof and 'spectrum_bin' not in prof and 'fft_cols_x' not in prof, sorted(prof)
    if world > 1:
        sent = {m: sum(r[0] for r in out[m]) for m in out}
        assert sent['compact'] <= 0.80 * sent['regular'], sent
    for (_, a), (_, b) in zip(out['compact'], out['regular']):
        for k in ('power', 'N_mode', 'poles', 'k_avg'):
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    scale = np.abs(np.asarray(ref['power'])).max()
    for _, t in out['compact']:
        np.testing.assert_array_equal(t['N_mode'], np.asarray(ref['N_mode']))
        np.testing.assert_allclose(t['power'], np.asarray(ref['power']), rtol=1e-5, atol=1e-6 * scale)
        np.testing.assert_allclose(t['poles'], np.asarray(ref['poles']), rtol=1e-5, atol=1e-6 * scale)


@pytest.mark.gpu
def test_config5_in_miniature_eight_ranks_as_threads():
    """BASELINE config 5's composition on eight ranks (threads on the one GPU): sharded multi-tracer HOD -> every rank keeps its
    galaxies -> folded-slab routing -> LRG x ELG cross P(k) over the slabs, and DD(r) of the ELGs over x-slabs - against the
    single-process pipeline (one HOD, calc_power, the pair counter) on the same catalogue"""
    from thread_comm import run_ranks

    from abacusutils_amd import synth
    from abacusutils_amd.analysis import slab_pairs, slab_power as sp
    from abacusutils_amd.analysis.power_spectrum import calc_power
    from abacusutils_amd.analysis.tpcf_corrfunc import _paircount
    from abacusutils_amd.hod import shard
    from abacusutils_amd.hod.GRAND_HOD import gen_gal_cat
    hd, pd, params = synth.synth_hod_inputs(400000, 600000, seed=91)
    box = float(params['Lbox'])
    tracers = {'LRG': dict(synth.LRG_PARAMS), 'ELG': dict(synth.ELG_PARAMS)}
    kw = dict(kbins=24, mubins=3, paste='TSC', nmesh=128, compensated=True, interlaced=True, poles=[0, 2])
    bins = np.geomspace(0.5, box / 8 * 0.9, 8).astype(np.float32)
    one = gen_gal_cat(hd, pd, tracers, params, rsd=True)
    xyz = {t: np.column_stack([one[t][c] for c in 'xyz']).astype(np.float32) for t in tracers}
    assert len(xyz['LRG']) > 500 and len(xyz['ELG']) > 2000
    ref_pk = calc_power(xyz['LRG'].copy(), box, pos2=xyz['ELG'].copy(), **kw)
    e = xyz['ELG']
    ref_dd = _paircount(0, e[:, 0], e[:, 1], e[:, 2], box, bins)

    def rank_fn(tc):
        local = shard.run_hod_sharded(hd, pd, tracers, params, comm=shard.HodComm(tc), rsd=True, gather=False)
        mine = {t: np.column_stack([local[t][c] for c in 'xyz']).astype(np.float32) for t in tracers}
        p1, _ = sp.route_particles(mine['LRG'], None, box, tc, fold=True)
        p2, _ = sp.route_particles(mine['ELG'], None, box, tc, fold=True)
        t = sp.calc_power_slab(p1, box, comm=tc, backend=sp.HipSlabBackend(), pos2=p2, **kw)
        dd = slab_pairs.paircount_slab('r', mine['ELG'], box, bins, comm=tc)
        return {k: np.asarray(t[k]) for k in ('power', 'N_mode', 'poles')}, dd

    for t, dd in run_ranks(8, rank_fn):
        np.testing.assert_array_equal(t['N_mode'], np.asarray(ref_pk['N_mode']))
        scale = np.abs(np.asarray(ref_pk['power'])).max()
        np.testing.assert_allclose(t['power'], np.asarray(ref_pk['power']), rtol=2e-5, atol=2e-6 * scale)
        np.testing.assert_allclose(t['poles'], np.asarray(ref_pk['poles']), rtol=2e-5, atol=2e-6 * scale)
        np.testing.assert_array_equal(dd, ref_dd)
