/*
 * abacus_hip.h - C ABI of libabacus_hip.so (MI355X / gfx950).
 *
 * The reference (abacusorg/abacusutils) has no FFI seam: its hot path is Python + Numba and the drop-in
 * boundary is the Python call signature.  This header is the C boundary underneath our
 * Python mirror of those signatures; each entry point names the reference function(s) it replaces
 * (paths relative to abacusnbody/).  INTEGRATION.md shows the ctypes binding a maintainer of the reference
 * would add.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; abacus_last_error() returns a thread-local message;
 *   - the caller owns all host buffers; the library reads/writes them only during the call;
 *   - device state lives behind explicit handles (`*_stage` / `*_free`), mirroring the residency of
 *     AbacusHOD.staging() (hod/abacus_hod.py:193-197): stage once, populate many times;
 *   - `*_dev` variants take DEVICE pointers and never synchronise with the host: they enqueue on the library
 *     stream (abacus_get_stream), for callers that keep data in HBM (bench, multi-GPU slab path);
 *   - calls are blocking unless stated; re-entrant per handle; HIP is initialised lazily on first use
 *     (fork-safe in the sense of NUMBA_THREADING_LAYER=forksafe, docs/hod.rst:226-240);
 *   - thread counts of the reference API (`Nthread`, `nthread`) have no meaning here and are not part of the ABI.
 */
#ifndef ABACUS_HIP_H
#define ABACUS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- runtime ------------------------------ */
const char *abacus_last_error(void);
int abacus_device_count(int *n);
int abacus_set_device(int device);          /* before any other call; default device 0 */
int abacus_device_name(char *buf, int len);
int abacus_device_sync(void);               /* wait for the library stream */
void *abacus_get_stream(void);              /* hipStream_t the library launches on */
int abacus_set_stream(void *hip_stream);    /* adopt a caller stream (e.g. torch's current stream) */

/* raw device memory for callers that keep inputs resident (bench.py, slab-parallel path) */
int abacus_malloc(void **dptr, uint64_t nbytes);
int abacus_free(void *dptr);
int abacus_memcpy_h2d(void *dst, const void *src, uint64_t nbytes);
int abacus_memcpy_d2h(void *dst, const void *src, uint64_t nbytes);
int abacus_memset(void *dptr, int value, uint64_t nbytes);
/* page-locked host memory (hipHostMalloc): device-to-host copies into it are single DMAs at link speed.  The Python side
 * hands catalogue columns out as NumPy views of such blocks and recycles them (abacusutils_amd/_lib.py pinned_empty) */
int abacus_host_alloc(void **hptr, uint64_t nbytes);
int abacus_host_free(void *hptr);
/* sum_i word[i] * (2 i + 1) mod 2^64 over n 8-byte words in HBM: position-dependent checksum of a device column (is a host
 * copy of the column still what the device holds?) */
int abacus_poshash_u64(const void *dptr, int64_t n, uint64_t *out);

/* HIP-event timers on the library stream (bench.py: roofline.achieved) */
int abacus_event_create(void **ev);
int abacus_event_record(void *ev);
int abacus_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises on `stop` */
int abacus_event_destroy(void *ev);
/* per-kernel timing: when enabled every kernel launch is bracketed by events on the library stream */
int abacus_profile_enable(int on);
int abacus_profile_reset(void);
/* bracket only the kernel called `name` (NULL or "": all kernels) - keeps the event overhead out of short steps */
int abacus_profile_select(const char *name);
/* writes up to `cap` entries; returns the number of distinct kernels (names are static strings) */
int abacus_profile_get(const char **names, double *total_ms, int64_t *launches, int cap);

/* Diagnostic options: comparator code paths that the parity tests and the A/B scripts switch on (the library reads no
 * environment variables).  Names: fft_nofuse (plain three-pass FFT), fft_hipfft, fft_fuse_small, pk_noxbin (separate last
 * pass + spectrum_bin), tsc_atomic, tsc_noshare, pairs_gen (1 / 2: older pair kernels), hod_nocls, hod_one_stage,
 * hod_f64filter, hod_norec, hod_nokeys, hod_pipe (1 / 2: pipelined hod_exact off / on), hod_eblock (256 / 512 threads per
 * hod_emit workgroup), hod_sbtiles (8 / 16 tiles per superblock), hod_nolazy / hod_noindex (no lazy
 * keep masks / no mass-sorted key index), dbg / dbg_fft / dbg_tsc (ablation bit masks).  Default 0 = the production path. */
int abacus_set_option(const char *name, int value);
int abacus_get_option(const char *name);

/* ---------------------------------------------------------------- HOD ---------------------------------- */
/*
 * Flat form of the three numba typed dicts gen_gals builds (hod/GRAND_HOD.py:1342-1468) plus the scalars it
 * passes to gen_cent / gen_sats (:1472-1475).  z-evolution and defaults are applied by the caller (Python
 * mirror of gen_gals) before filling this struct.
 */
typedef struct abacus_hod_params {
    int32_t want_LRG, want_ELG, want_QSO;
    int32_t rsd, has_origin, enable_ranks;
    int32_t pad0, pad1;
    double inv_velz2kms, lbox, origin[3];
    /* LRG_hod_dict */
    double L_logM_cut, L_logM1, L_sigma, L_alpha, L_kappa, L_alpha_c, L_alpha_s;
    double L_s, L_s_v, L_s_p, L_s_r, L_Acent, L_Asat, L_Bcent, L_Bsat, L_ic;
    /* ELG_hod_dict */
    double E_p_max, E_Q, E_logM_cut, E_kappa, E_sigma, E_logM1, E_alpha, E_gamma, E_A_s;
    double E_alpha_c, E_alpha_s, E_s, E_s_v, E_s_p, E_s_r;
    double E_Acent, E_Asat, E_Bcent, E_Bsat, E_Ccent, E_Csat, E_ic;
    double E_logM1_EE, E_alpha_EE, E_logM1_EL, E_alpha_EL;
    /* QSO_hod_dict */
    double Q_logM_cut, Q_kappa, Q_sigma, Q_logM1, Q_alpha, Q_alpha_c, Q_alpha_s;
    double Q_s, Q_s_v, Q_s_p, Q_s_r, Q_Acent, Q_Asat, Q_Bcent, Q_Bsat, Q_ic;
} abacus_hod_params;

/*
 * The staged halo / particle subsample: the arrays of AbacusHOD.halo_data / particle_data
 * (hod/abacus_hod.py:659-702) in their reference layout - float64, (N,3) C-order, int64 ids.
 * Optional arrays (NULL = absent, treated as zeros like gen_gals does, hod/GRAND_HOD.py:1485-1487,1541-1543):
 * hdeltac, hfenv, hshear, pdeltac, pfenv, pshear.  pranks* may be NULL when ranks are never enabled.
 * pinds[i] = index of particle i's host halo (hod/abacus_hod.py:588); the keep_cent[pinds] gather the
 * reference does on the host (hod/GRAND_HOD.py:1562) happens on the device.
 */
typedef struct abacus_hod_arrays {
    int64_t n_halo;
    const double *hpos, *hvel, *hmass;
    const int64_t *hid;
    const double *hmultis, *hrandoms, *hveldev, *hdeltac, *hfenv, *hshear;
    int64_t n_part;
    const double *ppos, *pvel, *phvel, *phmass;
    const int64_t *phid;
    const double *pweights, *prandoms, *pdeltac, *pfenv, *pshear;
    const double *pranks, *pranksv, *pranksp, *pranksr;
    const int64_t *pinds;
} abacus_hod_arrays;

typedef struct abacus_hod_state abacus_hod_state;

/* replaces: the host residency set up by AbacusHOD.staging() (hod/abacus_hod.py:253-704); uploads once.
 * `arrays_on_device` != 0: the pointers are device pointers that the handle adopts WITHOUT copying (the caller
 * keeps them alive until abacus_hod_free). */
int abacus_hod_stage(const abacus_hod_arrays *arrays, int arrays_on_device, abacus_hod_state **out);
/* re-upload one staged array after `reseed` rewrote it (hod/abacus_hod.py:824-835).
 * field: "hrandoms" | "hveldev" | "prandoms" (float64 host data, staged length) */
int abacus_hod_update(abacus_hod_state *st, const char *field, const double *host);
/* Device-side `reseed` (hod/abacus_hod.py:775-839): rewrites hrandoms, hveldev and prandoms in HBM from a counter-based
 * Philox4x32-10 generator keyed by `seed` - float32 U[0,1) uniforms, float32 N(0,1) (want_expvel: the two-sided
 * exponential of :799-801) times hsigma3d / sqrt(3) - with no host traffic.  A value depends only on (seed, global
 * object index): `halo_index0` / `part_index0` are the global indices of this catalogue's first halo / particle, so
 * the shards of a multi-GPU run draw what the unsharded catalogue would.  The reference's own stream comes from the
 * third-party parallel_numpy_rng (not in its tree): stream parity is unpinned, distributions and dtypes are the same.
 * abacus_hod_set_sigma3d stages the per-halo hsigma3d once; abacus_hod_fetch_field copies a rewritten array back
 * ("hrandoms" | "hveldev" | "prandoms") for callers that need the reference's host-side mutation (:824-835). */
int abacus_hod_set_sigma3d(abacus_hod_state *st, const double *hsigma3d, int on_device);
int abacus_hod_reseed(abacus_hod_state *st, uint64_t seed, int want_expvel, int64_t halo_index0, int64_t part_index0);
int abacus_hod_fetch_field(abacus_hod_state *st, const char *field, double *host);

/* `AbacusHOD.compute_ngal` (hod/abacus_hod.py:861-1179): expected numbers of centrals and satellites per tracer from the
 * weighted halo histogram.  The reference's sum over the 100^3 / 100^4 histogram cells is evaluated as the identical
 * sum over halos, sum_i multis[i] * n(centre of the cell of halo i).  bins4[i] = {logM, deltac, fenv, shear} cell of
 * halo i under np.histogramdd's edge rule (255 = outside: dropped, as histogramdd drops it); centres = [4][nbin]:
 * 10**logM centres, then the deltac, fenv and shear centres.  out = Ncent[3], Nsat[3] (LRG, ELG, QSO); `p` is the
 * marshalled parameter struct (z-evolution applied). */
int abacus_hod_set_ngal_bins(abacus_hod_state *st, const uint8_t *bins4, const double *centres, int nbin);
int abacus_hod_ngal(abacus_hod_state *st, const abacus_hod_params *p, double out[6]);

/* NFW satellites: `gen_gal_cat(..., nfw=True, NFW_draw=...)` (gen_sats_nfw, compute_fast_NFW, getPointsOnSphere,
 * hod/GRAND_HOD.py:417-822).  Centrals are decided exactly as on the particle path; satellites are Poisson(n_sat(M) ic)
 * per halo and tracer, placed isotropically at r = NFW_draw[k] / c * Rvir (k random with NFW_draw[k] <= c) around
 * their halo with N(v_halo, (0.577 f_sigv vrms)^2) velocities; box RSD only.  The reference draws from NumPy's
 * unseeded per-thread generators (never reproducible): parity is statistical; here the draws are
 * Philox streams of (seed, global halo index, tracer, satellite rank).  Needs abacus_hod_set_sigma3d (vrms) and
 * abacus_hod_set_profile (hc = r98/r25 concentration, hrvir) once per staged catalogue. */
typedef struct abacus_nfw_params {
    uint64_t seed;
    double f_sigv[3];                         /* LRG, ELG, QSO (:566,605,623) */
    double exp_frac, exp_scale, nfw_rescale;  /* ELG_hod_dict values, applied to every tracer as the reference does (:606-608) */
    int64_t halo_index0;                      /* global index of this catalogue's first halo (sharding) */
} abacus_nfw_params;
int abacus_hod_set_profile(abacus_hod_state *st, const double *hc, const double *hrvir);
int abacus_hod_populate_nfw(abacus_hod_state *st, const abacus_hod_params *p, const abacus_nfw_params *nfw,
                            const double *NFW_draw, int64_t n_draw, int64_t counts[6]);
/*
 * replaces: gen_cent + gen_sats + fast_concatenate (hod/GRAND_HOD.py:139-414, 825-1262, 1265-1299) as called from
 * gen_gals (:1477-1589).  Decides and emits on the device; galaxies of tracer t are left in device buffers in the
 * reference's order: centrals (halo order) then satellites (particle order).
 * counts[0..2] = Ncent (LRG, ELG, QSO), counts[3..5] = Nsat.
 */
int abacus_hod_populate(abacus_hod_state *st, const abacus_hod_params *p, int64_t counts[6]);
/* as above but enqueue only (no host sync, counts stay on the device until abacus_hod_counts) */
int abacus_hod_populate_async(abacus_hod_state *st, const abacus_hod_params *p);
int abacus_hod_counts(abacus_hod_state *st, int64_t counts[6]); /* syncs; re-runs emission if buffers grew */
/* diagnostic: how many halos (out[0]) and particles (out[1]) the last populate's filter passed on to the exact
 * float64 decision (no reference counterpart: the reference evaluates every object, hod/GRAND_HOD.py:210,954) */
int abacus_hod_candidates(abacus_hod_state *st, int64_t out[2]);
/* copy tracer t's catalog (x,y,z,vx,vy,vz,mass: float64; id: int64), each of length Ncent+Nsat, to the host */
int abacus_hod_fetch(abacus_hod_state *st, int tracer, double *x, double *y, double *z, double *vx, double *vy,
                     double *vz, double *mass, int64_t *id);
/* the same catalogue as ONE device-to-host transfer: out8 = [8][n] (x, y, z, vx, vy, vz, mass as float64, id as int64),
 * n = Ncent + Nsat of the tracer (checked against n_expected) */
int abacus_hod_fetch_block(abacus_hod_state *st, int tracer, void *out8, int64_t n_expected);
/* device pointers of tracer t's catalog columns (7 float64 + 1 int64), valid until the next populate */
int abacus_hod_device_columns(abacus_hod_state *st, int tracer, void *cols[8]);
/* the int8 masks gen_cent / gen_sats compute (hod/GRAND_HOD.py:210,954); either pointer may be NULL */
int abacus_hod_fetch_keep(abacus_hod_state *st, int8_t *keep_cent, int8_t *keep_sat);
int abacus_hod_free(abacus_hod_state *st);

/* ---------------------------------------------------------------- TSC / CIC ---------------------------- */
/* dtype codes */
#define ABACUS_F32 0
#define ABACUS_F64 1

/*
 * replaces: tsc_parallel = _wrap_inplace + partition_parallel + _tsc_parallel/_tsc_scatter
 * (analysis/tsc.py:10-206, 219-226, 259-384, 229-256, 394-507).
 * pos: (n,3) host array of `pos_dtype`; weights: (n,) same dtype or NULL; grid: (gx,gy,gz) host array of
 * `grid_dtype`, ACCUMULATED into (not zeroed), as the reference does (tsc.py:45-50).
 * wrap != 0: positions are wrapped to [0,box) and the wrapped values are written back to `pos` (the reference
 * mutates the caller's array, tsc.py:171-173).
 */
int abacus_tsc_deposit(void *pos, int64_t n, const void *weights, int pos_dtype, void *grid, int gx, int gy,
                       int gz, int grid_dtype, double box, double offset, int wrap);
/* device-resident variant: pos/weights/grid are device pointers; float32 positions and grid only.
 * zero_grid != 0 overwrites the grid instead of accumulating (saves the separate zeroing pass of get_field,
 * analysis/power_spectrum.py:842).  Enqueues on the library stream. */
int abacus_tsc_deposit_dev(float *pos, int64_t n, const float *weights, float *grid, int gx, int gy, int gz,
                           double box, double offset, int wrap, int zero_grid, int cic);
/* diagnostic (option tsc_lines_clk = 1): shader-clock ticks per phase of the list build's split rounds, summed over workgroups since
 * the last call; out32[0..9] coarse pass (count, barrier, owner, barrier, owner + carries, barrier, place, barrier, write-out,
 * geometry + loads), out32[16..24] fine pass.  Reads and clears. */
int abacus_tsc_lines_clocks(unsigned long long *out32);
/* replaces: cic_serial (analysis/cic.py:13-125): float64 math, float32 grid, no wrap */
int abacus_cic_deposit(const void *pos, int64_t n, const void *weights, int pos_dtype, float *grid, int gx, int gy,
                       int gz, double box);
/* replaces: partition_parallel (analysis/tsc.py:259-384), sort=False: stable counting sort into `npartition`
 * stripes along `coord`.  psort (n,3), starts (npartition+1) int64, wsort (n,) or NULL. */
int abacus_partition(const void *pos, int64_t n, const void *weights, int dtype, int npartition, double box,
                     int coord, void *psort, int64_t *starts, void *wsort);

/* ---------------------------------------------------------------- power spectrum ----------------------- */
/*
 * replaces: get_field_fft (analysis/power_spectrum.py:1001-1070) = get_field (:808-857: zero mesh, deposit,
 * normalize_field :860-901) + scipy.fft.rfftn (:980,986,1059) + _normalize (:1073-1078) or the interlacing
 * combine shift_field_fft (:904-948) + compensation divide (:1063-1069).
 * pos (n,3) float32 host; w (n,) float32 or NULL; W (nmesh,) float32 window or NULL (= not compensated);
 * out: (nmesh, nmesh, nmesh/2+1) complex64 host.  paste: 0 TSC, 1 CIC.
 */
/* get_field (analysis/power_spectrum.py:808-857): overdensity mesh delta = rho * f32(M / len(pos)) - 1 of the particles,
 * deposit and normalisation fused on the device; field: (nmesh, nmesh, nmesh) float32 host array.  TSC wraps pos in place. */
int abacus_field(float *pos, int64_t n, const float *w, double Lbox, int nmesh, int paste, double offset, float *field);
int abacus_field_fft(float *pos, int64_t n, const float *w, double Lbox, int nmesh, int paste, const float *W,
                     int interlaced, void *out_c64);
/*
 * float64 MESHES: get_field / get_field_fft / calc_power with dtype=np.float64 (analysis/power_spectrum.py:808-857, 1001-1070,
 * 1131-1319), non-interlaced (the reference's interlaced branch ignores dtype, :1048-1052).  pos_f64: the positions (and weights)
 * are float64 - the cloud weights are evaluated in the dtype of the positions (analysis/tsc.py:400).  The mesh is deposited,
 * normalised and transformed in float64 (csrc/gfft.hip: even sizes up to 2560 with factors 2, 3, 5, 7, 11, 13 - what fits the 160-KiB LDS tile in double; hipFFT's D2Z otherwise); field:
 * (nmesh,)*3 float64, out_c128: (nmesh, nmesh, nmesh/2+1) complex128; abacus_power_f64 returns what bin_kmu returns, times L^3.
 */
int abacus_field_f64(void *pos, int pos_f64, int64_t n, const void *w, double Lbox, int nmesh, int paste, double offset, double *field);
int abacus_field_fft_f64(void *pos, int pos_f64, int64_t n, const void *w, double Lbox, int nmesh, int paste, const float *W,
                         void *out_c128);
int abacus_power_f64(void *pos, int pos_f64, int64_t n, const void *w, void *pos2, int64_t n2, const void *w2, double Lbox, int nmesh,
                     int paste, const float *W, const double *kedges, int Nk, const double *muedges, int Nmu, const int64_t *poles,
                     int Np, float *power, int64_t *N_mode, float *binned_poles, int64_t *N_mode_poles, float *k_avg);
/*
 * replaces: calc_pk_from_deltak = get_raw_power + bin_kmu (analysis/power_spectrum.py:730-805, 707-727, 150-300)
 * on host spectra.  field2 may be NULL (auto power).  Outputs as bin_kmu returns them, already multiplied by L^3:
 * power (Nk,Nmu) f32, N_mode (Nk,Nmu) i64, binned_poles (Np,Nk) f32, N_mode_poles (Nk) i64, k_avg (Nk,Nmu) f32.
 */
int abacus_pk_from_deltak(const void *field_c64, const void *field2_c64, int nmesh, double Lbox,
                          const double *kedges, int Nk, const double *muedges, int Nmu, const int64_t *poles, int Np,
                          float *power, int64_t *N_mode, float *binned_poles, int64_t *N_mode_poles, float *k_avg);
/*
 * replaces: the whole calc_power chain (analysis/power_spectrum.py:1131-1319) without the spectrum ever leaving
 * HBM: deposit(s) -> FFT(s) -> [interlace combine] -> fused scale/compensate/|delta_k|^2/bin pass.
 * pos2 == NULL -> auto power.  Host inputs; `pos`/`pos2` are wrapped in place like the reference.
 */
int abacus_power_from_particles(float *pos, int64_t n, const float *w, float *pos2, int64_t n2, const float *w2,
                                double Lbox, int nmesh, int paste, const float *W, int interlaced,
                                const double *kedges, int Nk, const double *muedges, int Nmu, const int64_t *poles,
                                int Np, float *power, int64_t *N_mode, float *binned_poles, int64_t *N_mode_poles,
                                float *k_avg);
/* the same for float64 positions (and weights): the cloud weights are evaluated in float64, as the reference computes them
 * in the dtype of the positions (analysis/tsc.py:400 `ftype = positions.dtype.type`); the mesh and the transform stay float32 */
int abacus_power_from_particles_f64(double *pos, int64_t n, const double *w, double *pos2, int64_t n2, const double *w2,
                                    double Lbox, int nmesh, int paste, const float *W_host, int interlaced,
                                    const double *kedges, int Nk, const double *muedges, int Nmu, const int64_t *poles,
                                    int Np, float *power, int64_t *N_mode, float *binned_poles, int64_t *N_mode_poles,
                                    float *k_avg);
/* same with DEVICE particle arrays (bench / HOD-to-P(k) without leaving HBM); outputs are host arrays */
int abacus_power_from_particles_dev(float *pos, int64_t n, const float *w, float *pos2, int64_t n2,
                                    const float *w2, double Lbox, int nmesh, int paste, const float *W_host,
                                    int interlaced, const double *kedges, int Nk, const double *muedges, int Nmu,
                                    const int64_t *poles, int Np, float *power, int64_t *N_mode,
                                    float *binned_poles, int64_t *N_mode_poles, float *k_avg);

/* Multi-tracer spectra (hod/abacus_hod.py:1338-1472 `compute_power`: every auto and cross pair of the tracers) without
 * repeated work and without leaving HBM: `abacus_power_field_soa64` deposits + transforms ONE tracer's galaxies - given as
 * the float64 x | y | z columns of abacus_hod_device_columns, cast to float32 and wrapped like calc_power does - into field
 * slot `slot` (< 8), which stays resident; `abacus_power_from_fields` bins the auto (slot_a == slot_b) or cross spectrum of
 * two resident fields.  LRG x ELG: 2 deposits + FFTs instead of the 4 of three calc_power calls. */
int abacus_power_field_soa64(int slot, const double *x, const double *y, const double *z, int64_t n, double Lbox, int nmesh,
                             int paste, int interlaced);
int abacus_power_from_fields(int slot_a, int slot_b, const float *W_host, const double *kedges, int Nk, const double *muedges,
                             int Nmu, const int64_t *poles, int Np, float *power, int64_t *N_mode, float *binned_poles,
                             int64_t *N_mode_poles, float *k_avg);
int abacus_power_fields_release(void);
/* releases cached FFT plans / work meshes / the cached binning geometry of fft_x_bin */
int abacus_power_release(void);
/* frees the idle scratch blocks the host-array entry points (abacus_prepare_*, abacus_argsort_i64, abacus_searchsorted_i64,
 * abacus_fenv_rank) keep between calls instead of paying hipMalloc + hipFree per temporary */
int abacus_scratch_release(void);
/* milliseconds of the one-off geometry pass behind the most recent fused last pass (abacus_power_from_particles[_dev],
 * auto power, non-interlaced, nmesh 1024 / 2048): N_mode, k_avg and the (k, mu) bin of every mode depend on (nmesh, edges)
 * alone (power_spectrum.py:233-256), so they are computed once per (nmesh, edges) and cached; 0 if none was built */
double abacus_power_geometry_ms(void);
/* diagnostic: batches in which the most recent abacus_power_from_particles uploaded its (first) host position array - more than one
 * when the upload runs on a copy stream behind the deposits of the batches before (csrc/power.hip, HostSrc; option pk_nobatch = 1
 * forces one) */
double abacus_power_last_batches(void);
/* which fused last pass served the most recent spectrum: 2 = cached-geometry kernel, 1 = first generation (bin walk in the
 * kernel: more than 8 mu bins, edges the cell table cannot resolve, option pk_xbin_gen = 1), 0 = none yet */
int abacus_power_xbin_generation(void);


/* ---------------------------------------------------------------- multi-GPU slab building blocks ------- */
/*
 * The reference has no distributed mesh (docs/tutorials/analysis/tsc.ipynb:19); this is new functionality behind
 * calc_power: x-slab mesh decomposition, ghost-plane exchange-add, local z/y FFT passes, all-to-all pencil transpose,
 * x FFT pass and binning on y-slabs, all-reduce of the (k, mu) histogram.  The collectives are the
 * abacus_comm_* entry points below (RCCL); these entry points are the device-side pieces.
 * All pointers are DEVICE pointers; nmesh must be a power of two in [64, 2048] and divisible by twice the number of ranks.
 * Mesh rows have abacus_slab_pitch(nmesh) floats (128-B aligned rows; complex rows of pitch/2).
 */
int abacus_slab_pitch(int nmesh);
/* deposit into planes [xoff, xoff + nx_local) (mod nmesh) of the global mesh as rho*norm - sub (ghost planes included);
 * xoff2 >= 0: TWO windows of nx_local planes each, starting at xoff and xoff2 and stored back to back (the two slabs of a
 * rank's folded pair with their ghosts) - a plane both windows hold receives its deposits in the first.  Particles whose
 * clouds lie outside are skipped.
 * sub = 1: every cell already carries the "-1" of the overdensity (normalize_field, power_spectrum.py:860-901) - a ghost
 * block is then added to its owner as `ghost + 1` (abacus_slab_axpy_dev with add = 1) and no pass over the mesh is spent
 * on the subtraction; xoff2 < 0 and nx_local == nmesh with xoff == 0 is the whole periodic mesh (one rank: no ghosts) */
int abacus_slab_deposit_dev(float *pos, int64_t n, const float *w, float *mesh, int nmesh, int xoff, int xoff2, int nx_local,
                            double Lbox, double offset, double norm, int paste, double sub);
/* the same with a `mesh` buffer the caller padded to nx_alloc planes, nx_alloc the window planes (nx_local, or 2 nx_local with two
 * windows) rounded up to a multiple of 16: unweighted float32 TSC of >= 2e6 particles then takes the third-generation list build
 * of the single-GPU path on the windows' local planes (csrc/tsc_lines3.hpp, L3Win); the padding planes are overwritten.
 * nx_alloc = 0: no padding, the first-generation lists (what abacus_slab_deposit_dev runs). */
int abacus_slab_deposit_padded_dev(float *pos, int64_t n, const float *w, float *mesh, int nmesh, int xoff, int xoff2, int nx_local,
                                   double Lbox, double offset, double norm, int paste, double sub, int nx_alloc);
/* dst[i] += src[i] + add  (ghost-plane accumulation; a constant alone with src == NULL) */
int abacus_slab_axpy_dev(float *dst, const float *src, int64_t nfloat, float add);
/* FOLDED SLABS.  With W ranks the mesh is cut into 2 W slabs of h = nmesh / (2 W) planes and rank r owns slabs r and
 * r + W: the planes x and x + nmesh/2 of a pair sit on ONE rank, so the fused form of the transform (first radix-2 stage of
 * y and x inside the z pass, fft.hip) runs on a slab exactly as on the whole mesh.  A rank's buffer holds its two halves
 * `xsep` planes apart (ghost planes in between).
 *
 * z and y passes of the plane pairs [p0, p0 + pc) of the rank's h pairs: `mesh` = first plane of the first half (global
 * plane xg0; the second half is global plane xg0 + nmesh/2).  send == NULL: in place.  send != NULL: the result goes to the
 * send buffer of the pencil transpose, send[peer][s h + p][y_local][k] (s = 0 / 1: first / second half) - written by the y
 * pass itself where the fused form runs with more than one rank, by a pack pass otherwise */
int abacus_slab_fft_zy_dev(float *mesh, void *send, int nmesh, int world, int64_t xsep, int xg0, int p0, int pc);
/* COMPACT pencil transpose (new functionality like the rest of the slab path; csrc/fft.hip `slab_layout`).  When the spectrum goes
 * to a binning that ends at k_last and nowhere else, the columns of a y row beyond sqrt(k_last^2 - ky^2) never cross the links:
 * a row keeps its first row_len columns (a multiple of 16, decided per aligned group of 16 rows), the rows of a peer's plane block
 * are packed back to back - 21 % fewer bytes when the bins end at the Nyquist frequency.
 * abacus_slab_transpose_layout: P_out[p] = complex elements per plane of the block that goes to peer p (so a rank sends
 * 2 h P_out[p] elements to p and receives 2 h P_out[rank] from everybody); returns 1 when this mesh / rank count has no compact
 * form (then use the regular layout).  abacus_slab_fft_zy_compact_dev writes that send buffer; the receive buffer is read by
 * abacus_slab_xbin_dev(from_transpose = 2) */
int abacus_slab_transpose_layout(int nmesh, int world, double Lbox, double k_last, int64_t *P_out);
int abacus_slab_fft_zy_compact_dev(float *mesh, void *send, int nmesh, int world, int64_t xsep, int xg0, int p0, int pc, double Lbox,
                                   double k_last);
/* the pack pass alone (pairs [p0, p0 + pc)), and recv[q][s h + p][y_local][k] -> out[y_local][s nmesh/2 + q h + p][k]: row
 * s nmesh/2 + i of x is plane i of half s - the plane itself in the plain form; in the fused form the sum (s = 0) or the
 * twiddled difference (s = 1) of planes i and i + nmesh/2, i.e. the inputs of the two nmesh/2-point transforms */
int abacus_slab_pack_dev(const void *data, void *send, int nmesh, int world, int64_t xsep, int p0, int pc);
int abacus_slab_unpack_dev(const void *recv, void *out, int nmesh, int world);
/* x pass on a y-slab in the (y_local, x, k) layout, in place */
int abacus_slab_fft_x_dev(float *data, int nmesh, int ny_local);
/* raw (un-normalised) bin sums of a y-slab: counts u64[Nk*Nmu], sum P f64[Nk*Nmu], sum k f64[Nk*Nmu],
 * pole sums f64[Np'*Nk]; `raw_out` is a HOST buffer of abacus_bin_raw_bytes() bytes (to be all-reduced) */
int abacus_slab_bin_dev(const void *a, const void *as, const void *b, const void *bs, int nmesh, int y0, int ny_local,
                        double Lbox, const float *W_host, int interlaced, const double *kedges, int Nk,
                        const double *muedges, int Nmu, const int64_t *poles, int Np, void *raw_out);
int64_t abacus_bin_raw_bytes(int Nk, int Nmu, const int64_t *poles, int Np);
/* 1 if the slab entry points above run the FUSED form of the transform for this mesh (nmesh 1024 / 2048, like the single-GPU
 * path: n/2-point column passes with 128-B row segments; rows of x and y then come out in the order
 * f = 2 (r mod n/2) + (r div n/2), which abacus_slab_bin_dev and abacus_slab_xbin_dev undo) */
int abacus_slab_fused(int nmesh);
/* last x pass FUSED with the binning on a y-slab (auto power of one non-interlaced field): replaces abacus_slab_fft_x_dev +
 * abacus_slab_bin_dev.  from_transpose = 0: `mesh` is the unpacked (y_local, x, k) block; from_transpose = 1: `mesh` is the
 * RECEIVE buffer of the pencil transpose of a `world`-rank run as it arrived, (peer, 2 h, y_local, k) - no unpack at all,
 * the kernel gathers the rows of a column from the peers' blocks while staging (one rank: the slab itself after its z / y
 * passes).  Returns 0 (raw sums in the host buffer raw_out), 1 when this mesh / histogram is not served (use the two-step
 * form), < 0 on error.  put_geom = 1 on exactly one rank: N_mode and sum |k| are mesh-wide quantities taken from the
 * cached geometry of (nmesh, edges), not from the y-slab */
int abacus_slab_xbin_dev(const void *mesh, int nmesh, int world, int y0, int ny_local, double Lbox, const float *W_host,
                         const double *kedges, int Nk, const double *muedges, int Nmu, const int64_t *poles, int Np,
                         int put_geom, int from_transpose, void *raw_out);
/* the same over two fields in the same layout.  pair_mode 2: their cross power Re(conj(a) b) (calc_power(pos, pos2=...,
 * interlaced=False), analysis/power_spectrum.py:707-727 with field2_fft); pair_mode 1: mesh2 is the half-cell-shifted deposit of
 * the same particles - the auto power of the interlaced combination (:951-998); pair_mode 0 is abacus_slab_xbin_dev (mesh2
 * ignored).  The query form (mesh == NULL) answers for the pair_mode given */
int abacus_slab_xbin_pair_dev(const void *mesh, const void *mesh2, int pair_mode, int nmesh, int world, int y0, int ny_local, double Lbox,
                              const float *W_host, const double *kedges, int Nk, const double *muedges, int Nmu, const int64_t *poles,
                              int Np, int put_geom, int from_transpose, void *raw_out);
/* ... and over four: pair_mode 3 = the cross power of two INTERLACED fields (calc_power(pos, pos2=..., interlaced=True) over slabs,
 * analysis/power_spectrum.py:1200-1260): mesh / mesh2 the first catalogue's unshifted / shifted deposit, mesh3 / mesh4 the second
 * catalogue's, all in the layout of `mesh`; pair_mode 0 - 2 as abacus_slab_xbin_pair_dev (mesh3 / mesh4 ignored) */
int abacus_slab_xbin_quad_dev(const void *mesh, const void *mesh2, const void *mesh3, const void *mesh4, int pair_mode, int nmesh, int world,
                              int y0, int ny_local, double Lbox, const float *W_host, const double *kedges, int Nk, const double *muedges,
                              int Nmu, const int64_t *poles, int Np, int put_geom, int from_transpose, void *raw_out);
/* particle routing on the device: stable bucket sort of (pos (n,3) float32, w or NULL) by the rank that owns the wrapped x
 * (fold = 0: x-slabs of width Lbox / world; fold = 1: the folded slabs above, rank = slab mod world of 2 world slabs);
 * counts[world] on the host.  The blocks then travel with abacus_comm_all_to_all_v. */
int abacus_slab_route_dev(const float *pos, int64_t n, const float *w, double Lbox, int world, int fold, float *pos_out,
                          float *w_out, int64_t *counts);
/* bin_kmu's normalisation (analysis/power_spectrum.py:276-293, 789-792) of reduced raw sums; host only */
int abacus_bin_finalize(const void *raw, double Lbox, int Nk, int Nmu, const int64_t *poles, int Np, float *power,
                        int64_t *N_mode, float *binned_poles, int64_t *N_mode_poles, float *k_avg);

/* ---------------------------------------------------------------- multi-GPU communicator (RCCL) -------- */
/*
 * One process per GPU; the collectives of the slab-decomposed P(k), the sharded HOD and the slab pair counts
 * (SURVEY.md 8e).  New functionality: the reference is single-node shared-memory (its unit of decomposition is the slab
 * chunk, hod/abacus_hod.py:301-312).  librccl is opened on first use.  Rank 0 creates the 128-byte id and distributes it
 * out of band (abacusutils_amd/comm.py: file rendezvous); abacus_comm_init = ncclCommInitRank on the device selected with
 * abacus_set_device.  Device-buffer collectives are enqueued on the library stream and never synchronise with the host;
 * `async` != 0 runs them on the communicator's own stream behind the kernels enqueued so far - abacus_comm_join puts the
 * library stream behind them again (pencil-transpose chunks overlap the FFT passes of the next chunk).
 * dtype codes: 0 int64, 1 float64, 2 float32, 3 uint64; op: 0 sum, 1 max.
 */
#define ABACUS_COMM_ID_BYTES 128
typedef struct abacus_comm abacus_comm;
int abacus_comm_unique_id(void *id, int len);
int abacus_comm_init(int rank, int world, const void *id, int len, abacus_comm **out);
int abacus_comm_info(const abacus_comm *c, int *rank, int *world, int *rccl_version, uint64_t *bytes_sent);
int abacus_comm_free(abacus_comm *c);
int abacus_comm_abort(abacus_comm *c);   /* ncclCommAbort: after a failed or abandoned collective */
int abacus_comm_join(abacus_comm *c);
/* block p of `send` (bytes_per_peer each) goes to rank p, block p of `recv` comes from rank p: ONE group of
 * ncclSend / ncclRecv, every xGMI link busy at once */
int abacus_comm_all_to_all(abacus_comm *c, const void *send, void *recv, uint64_t bytes_per_peer, int async);
/* the same for the piece [offset, offset + bytes) of every peer block (blocks `peer_stride` bytes apart): chunks of the
 * pencil transpose */
int abacus_comm_all_to_all_strided(abacus_comm *c, const void *send, void *recv, uint64_t peer_stride, uint64_t offset,
                                   uint64_t bytes, int async);
/* variable blocks (device-side particle routing): byte counts and offsets per peer, host arrays of `world` entries */
int abacus_comm_all_to_all_v(abacus_comm *c, const void *send, const uint64_t *send_bytes, const uint64_t *send_off, void *recv,
                             const uint64_t *recv_bytes, const uint64_t *recv_off);
/* the same on the communicator's own stream behind an event fork when async != 0 (chunks of the compact pencil transpose) */
int abacus_comm_all_to_all_v_async(abacus_comm *c, const void *send, const uint64_t *send_bytes, const uint64_t *send_off, void *recv,
                                   const uint64_t *recv_bytes, const uint64_t *recv_off, int async);
/* ghost blocks: to_left -> rank-1, to_right -> rank+1; from_right <- what rank+1 sent left, from_left <- what rank-1 sent
 * right (one rank: its own blocks come back, the periodic box) */
int abacus_comm_ring_exchange(abacus_comm *c, const void *to_left, const void *to_right, void *from_right, void *from_left,
                              uint64_t bytes);
int abacus_comm_allreduce_dev(abacus_comm *c, void *buf, int64_t count, int dtype, int op);   /* in place, device buffer */
/* host-buffer conveniences for the few-KB messages (histograms, counts, timings): staged through device scratch, blocking */
int abacus_comm_allreduce_host(abacus_comm *c, void *buf, int64_t count, int dtype, int op);
int abacus_comm_allgather_host(abacus_comm *c, const void *send, void *recv, uint64_t bytes);
int abacus_comm_barrier(abacus_comm *c);

/* ---------------------------------------------------------------- ZCV-facing spectrum helpers ----------- */
/*
 * replaces the remaining Numba kernels of analysis/power_spectrum.py that the control-variates code calls
 * (zcv/tools_cv.py:320,805-923, zcv/tracer_power.py:456).  Grids are host arrays in the rfftn layout (n, n, n/2+1)
 * unless a z-extent `zdim` is given ((n, n, n) real-space grids: zdim = n, only k <= n/2 is read, like the reference).
 */
/* bin_kmu (:150-300) of a caller-supplied real grid: project_3d_to_poles (:415-448) with one mu bin and scale = L^3.
 * fourier = 0 bins a configuration-space grid (dk = L/n, :212).  scale multiplies the means (<= 0: 1). */
int abacus_bin_weights(const float *weights, int n1d, int zdim, double Lbox, int fourier, const double *kedges, int Nk,
                       const double *muedges, int Nmu, const int64_t *poles, int Np, double scale, float *power,
                       int64_t *N_mode, float *binned_poles, int64_t *N_mode_poles, float *k_avg);
/* pk_to_xi (:620-660): Xi = irfftn(Pk) on the device (hipFFT C2R), multipoles of Xi in r bins, times nmesh^3 */
int abacus_pk_to_xi(const float *Pk, int n, double Lbox, const double *redges, int Nr, const int64_t *poles, int Np,
                    float *binned_poles, int64_t *N_poles);
/* bin_kppi (:303-412): mean and mode count per (k_perp, pi) bin, pi edges linspace(0, pimax, Npi+1).  The reference's j
 * loop breaks at the first k_perp beyond the last edge (the rest of that i row is never visited): kept. */
int abacus_bin_kppi(const float *weights, int n1d, int zdim, double Lbox, const double *kedges, int Nk, double pimax, int Npi,
                    int fourier, float *mean, int64_t *counts);
/* get_raw_power (:707-727): |field|^2, or Re(conj(field) field2), of `total` complex64 values -> float32 */
int abacus_raw_power(const void *field_c64, const void *field2_c64, int64_t total, float *out);
/* shift_field_fft (:904-948), in place on `field`: (field + shift * exp(i (d / 2)(kx + ky + kz))) * 0.5 / n1d^3, float32
 * wavenumbers as the reference forms them */
int abacus_shift_field_fft(void *field_c64, const void *shift_c64, int n1d, double Lbox, double d);
/* get_smoothing (:527-577), get_delta_mu2 (:580-617), expand_poles_to_3d (:451-505) */
int abacus_get_smoothing(int n1d, double Lbox, double R, float *out);
int abacus_get_delta_mu2(const void *delta_c64, int n1d, void *out_c64);
int abacus_expand_poles_to_3d(const double *k_ell, const double *P_ell, int nk, int n1d, double Lbox, const int64_t *poles,
                              int Np, float *out);

/* ---------------------------------------------------------------- pair counting ------------------------ */
/*
 * replaces: Corrfunc.theory.DD / DDrppi / DDsmu as called at analysis/tpcf_corrfunc.py:144-156,164-179,
 * 240-273,328-362 and scripts/emulator/generate_cfs/generate_cf.py:65-74 (third-party C, periodic box,
 * float32 coordinates).  mode 0: DD(r), 1: DD(rp,pi) with pi bins of width pimax/npibins, 2: DD(s,mu) with
 * nmubins in [0,mu_max).  x2 == NULL -> autocorrelation (ordered pairs, no self pairs).
 * npairs: nbins * (1 | npibins | nmubins) uint64.
 */
int abacus_paircount(int mode, const float *x1, const float *y1, const float *z1, int64_t n1, const float *x2,
                     const float *y2, const float *z2, int64_t n2, float boxsize, const float *bins, int nbins,
                     float pimax, int npibins, float mu_max, int nmubins, uint64_t *npairs);
/* the same with the coordinate columns already in HBM (pos_dtype ABACUS_F32, or ABACUS_F64: cast to float32 on the
 * device as analysis/tpcf_corrfunc.py:134-139 casts on the host) - e.g. the catalogue columns of
 * abacus_hod_device_columns: the HOD -> clustering step of hod/abacus_hod.py:1181-1336 without a PCIe round trip.
 * Coordinates may lie in any interval of one box length ([0, L), [-L/2, L/2), ...), as Corrfunc accepts them. */
int abacus_paircount_dev(int mode, const void *x1, const void *y1, const void *z1, int64_t n1, const void *x2,
                         const void *y2, const void *z2, int64_t n2, int pos_dtype, float boxsize, const float *bins,
                         int nbins, float pimax, int npibins, float mu_max, int nmubins, uint64_t *npairs);

/* bookkeeping of the last pair-count call (bench.py's roofline): candidate pair separations the kernel evaluated (0 when an
 * older-generation kernel ran), cells per dimension in xy / z, stencil half-width in cells (1 or 2; 0: older kernel) */
int abacus_paircount_stats(uint64_t *candidates, int *ncell_xy, int *ncell_z, int *stencil_R);

/* ---------------------------------------------------------------- staging (the step in front of the HOD) ---- */
/*
 * Data-parallel pieces of AbacusHOD.staging() (hod/abacus_hod.py:253-704); host arrays in and out, staging runs once.
 * abacus_argsort_i64: stable ascending argsort (replaces `np.argsort(hid)` of the halo sort, :566-585).
 * abacus_searchsorted_i64: np.searchsorted(sorted, query, side='left') (replaces `_searchsorted_parallel`, :588).
 * abacus_fenv_rank: `calc_fenv_opt` (:1961-1970): out[i] = rank of Menv[i] among the halos with
 *   mbins[b] < halosM < mbins[b+1] of i's bin, / (N_bin - 1) - 0.5; 0 for halos in no bin or alone in theirs.
 */
int abacus_argsort_i64(const int64_t *keys, int64_t n, int64_t *order);
int abacus_searchsorted_i64(const int64_t *sorted, int64_t n, const int64_t *query, int64_t m, int64_t *out);
int abacus_fenv_rank(const double *Menv, const double *halosM, int64_t n, const double *mbins, int n_edges, double *out);

/* ---------------------------------------------------------------- prepare_sim (subsample preparation) ---- */
/*
 * replaces the data-parallel core of abacusnbody/hod/prepare_sim.py `prepare_slab` (:296-1052); host arrays in and out
 * (runs once per slab).  Host side and file layout: abacusutils_amd/hod/prepare_sim.py.
 *
 * abacus_prepare_halo_factors: p_halos[i] = subsample_halos(N[i] * Mpart, MT) (:83-108); mask[i] = u[i] < p_halos[i] (:449)
 *   when `mask` / `u` are given; ntarget[i] = the number of subsample particles submask_particles keeps for a halo of that
 *   mass with pnum[i] particles (:152-174) when `ntarget` / `pnum` are given.
 * abacus_prepare_particles: for every kept halo (hmask) with pnum > 0, keep a subset of its particle slice
 *   [pstart, pstart + pnum) - either the caller's `submask_in` (npart bytes, drawn on the host in the reference's order) or,
 *   with submask_in == NULL, a uniformly random subset of ntarget[j] particles from counter-based Philox keys (seed, particle
 *   index part_index0 + q).  Outputs: pstart_new / pnum_new per halo as the reference stores them (float64; -1 for dropped
 *   halos and halos without particles, :895-897, :979-981); *n_sel kept particles; if cap_sel >= *n_sel also their input index
 *   (ascending), host halo index, Np (:881) and, with want_ranks, the five rank columns ranks / ranksv / ranksp / ranksr /
 *   ranksc (:899-977).  The number of kept particles is known beforehand - the set bytes of `submask_in`, or the sum of
 *   ntarget over the kept halos - so one call with cap_sel of that size does it all; a call with cap_sel = 0 only sizes the
 *   outputs (pass its `submask_out` as `submask_in` of the second call, so that both see one draw).  Halo slices must be
 *   ordered like the halos (CompaSO's layout).
 */
int abacus_prepare_halo_factors(const uint32_t *N, int64_t n, double Mpart, int MT, const double *u, const int64_t *pnum,
                                double *p_halos, uint8_t *mask, int32_t *ntarget);
/* the reference's two-argument form, subsample_halos(m, MT) on float64 MASSES (hod/prepare_sim.py:83-108) */
int abacus_prepare_halo_factors_mass(const double *mass, int64_t n, int MT, double *p_halos);
int abacus_prepare_particles(int64_t nh, const uint8_t *hmask, const int64_t *pstart, const int64_t *pnum, const uint32_t *N,
                             const float *hpos, const float *hvel, const float *r25, const float *r98, int64_t npart,
                             const float *pos, const float *vel, const uint8_t *submask_in, const int32_t *ntarget, uint64_t seed,
                             int64_t part_index0, double Mpart, double h, int want_ranks, double *pstart_new, double *pnum_new,
                             int64_t *n_sel, int64_t cap_sel, int64_t *sel_idx, int64_t *sel_host, double *sel_np, double *ranks,
                             double *ranksv, double *ranksp, double *ranksr, double *ranksc, uint8_t *submask_out);

/*
 * The same slab prepared WITHOUT a column crossing PCIe twice (what hod/prepare_sim.py:296-1052 does between its loader and its
 * writer, `rng = <seed>` form): abacus_prepare_slab takes the CompaSO columns prepare_slab loads (:404-425; host pointers are
 * uploaded once, device pointers - what the reader's unpack kernels produce - are used in place), draws the halo mask, selects the
 * particles, ranks them and gathers the kept rows of every column of both tables in HBM; abacus_prepare_slab_fetch copies the
 * columns out, one copy each.  mbins / n_edges: the mass bins of the concentration rank (n_edges = 0: deltac_rank = 0, the
 * reference's want_AB = False); fenv_rank / shear_rank: per-halo columns the caller computed (light cones, shear) or NULL = 0.
 * mask_out (nh bytes, host, may be NULL): the halo mask.  Streams and values are those of abacus_prepare_halo_factors /
 * _particles / _randoms with the same seed: both paths give identical tables.
 */
typedef struct abacus_prepare_slab_args {
    int64_t nh, npart;
    const uint32_t *N;                           /* (nh) */
    const float *x, *v;                          /* (nh, 3) x_L2com, v_L2com */
    const float *r25, *r90, *r98, *sigmav;       /* (nh) */
    const int64_t *npstartA, *npoutA;            /* (nh) */
    const void *id;                              /* (nh) 8-byte ids */
    const float *pos, *vel;                      /* (npart, 3) subsample-A particles in halo order */
    const double *fenv_rank, *shear_rank;        /* (nh) or NULL */
    const double *mbins;                         /* (n_edges) host */
    int32_t n_edges, MT, want_ranks, pad_;
    double Mpart, h;
    uint64_t seed;
    int64_t halo_index0, part_index0;
} abacus_prepare_slab_args;
enum {  /* columns of the halo table, in the order of halo_cols[] */
    ABACUS_PREP_H_N = 0, ABACUS_PREP_H_X, ABACUS_PREP_H_V, ABACUS_PREP_H_R25, ABACUS_PREP_H_R90, ABACUS_PREP_H_R98, ABACUS_PREP_H_NPSTART,
    ABACUS_PREP_H_NPOUT, ABACUS_PREP_H_ID, ABACUS_PREP_H_SIGMAV, ABACUS_PREP_H_MASK, ABACUS_PREP_H_MULTI, ABACUS_PREP_H_FENV, ABACUS_PREP_H_DELTAC,
    ABACUS_PREP_H_SHEAR, ABACUS_PREP_H_RANDOMS, ABACUS_PREP_H_REXP, ABACUS_PREP_H_RGAUS, ABACUS_PREP_HALO_COLS
};
enum {  /* columns of the particle table, in the order of part_cols[]; RANKS .. RANKS + 4 = ranks, ranksv, ranksp, ranksr, ranksc */
    ABACUS_PREP_P_POS = 0, ABACUS_PREP_P_VEL, ABACUS_PREP_P_RANKS, ABACUS_PREP_P_DOWNSAMPLE = ABACUS_PREP_P_RANKS + 5, ABACUS_PREP_P_HALO_VEL,
    ABACUS_PREP_P_HALO_MASS, ABACUS_PREP_P_NP, ABACUS_PREP_P_HALO_ID, ABACUS_PREP_P_RANDOMS, ABACUS_PREP_P_DELTAC, ABACUS_PREP_P_FENV,
    ABACUS_PREP_P_SHEAR, ABACUS_PREP_PART_COLS
};
int abacus_prepare_slab(const abacus_prepare_slab_args *args, int64_t *n_halo_kept, int64_t *n_part_kept, uint8_t *mask_out);
int abacus_prepare_slab_fetch(void *const *halo_cols, void *const *part_cols);

/* the random columns of the prepare_sim tables drawn on the device (Philox4x32-10, counter = global object index: shard
 * invariant; NOT the reference's NumPy stream - prepare_sim.py:984-996,1029 - but its distributions and dtypes).  Row r stands
 * for the object index0 + (index ? index[r] : r).  stream_id 4, with scale / randoms_exp / randoms_gaus: the halo columns
 * `randoms` (n) U[0,1), `randoms_exp` (n,3) = +-Exp(1) * scale[r], `randoms_gaus_vrms` (n,3) = N(0,1) * scale[r]; stream_id
 * 5 .. 255 without them: one U[0,1) per object (5: the particles' `randoms`, 6: the draws of the halo mask).  Host pointers. */
int abacus_prepare_randoms(int64_t n, const int64_t *index, int64_t index0, uint64_t seed, int stream_id, const double *scale,
                            double *randoms, double *randoms_exp, double *randoms_gaus);

/* ---------------------------------------------------------------- catalogue side (upstream of the HOD) ---- */
/*
 * replaces: abacusnbody/data/bitpacked.py:32-116 `unpack_rvint` / `_unpack_rvint`.  intdata: (n,3) int32, 20-bit
 * position | 12-bit velocity per word.  pos = (x >> 12) * (boxsize / 1e6), vel = ((x & 0xFFF) - 2048) * 6000 / 2048,
 * formed in float64 and rounded once into float32 (out_f64 = 0) or kept float64 (out_f64 = 1).  posout / velout:
 * (n,3) or NULL (skip, the reference's `False`).  Every pointer of this block may be host or device memory.
 */
int abacus_unpack_rvint(const int32_t *intdata, int64_t n, double boxsize, int out_f64, void *posout, void *velout);
/*
 * replaces: abacusnbody/data/bitpacked.py:118-330 `unpack_pids` / `_unpack_pids`.  packed: (n) uint64 aux words.
 * Outputs (each may be NULL): pid (n) int64 = packed & 0x7FFF7FFF7FFF; lagr_idx (n,3) int16; lagr_pos (n,3) =
 * idx * float_dtype(box / ppd) - float_dtype(box / 2); tagged (n) uint8 = bit 48; density (n) = (bits 49..58)^2.
 */
int abacus_unpack_pids(const uint64_t *packed, int64_t n, double box, int64_t ppd, int out_f64, int64_t *pid,
                       void *lagr_pos, int16_t *lagr_idx, uint8_t *tagged, void *density);
/*
 * replaces: abacusnbody/data/pack9.py:16-123 `unpack_pack9` / `_unpack_pack9`.  data: (nrec, 9) bytes; a record whose
 * first byte is 0xFF is a cell header for the particles after it.  posout / velout: (nrec, 3) or NULL; the first
 * *npart rows are written (records before the first header decode to NaN, like the reference's initial state).
 */
int abacus_unpack_pack9(const uint8_t *data, int64_t nrec, double boxsize, double velzspace_to_kms, int out_f64,
                        void *posout, void *velout, int64_t *npart);
/*
 * replaces: abacusnbody/hod/menv.py:19-87 `do_Menv_from_tree` (scipy KDTree ball queries + gather sums; callers
 * hod/prepare_sim.py:603-612,728-737).  Menv[i] = sum of mass within r_outer of halo i minus the sum within r_inner,
 * over ALL halos (itself included in both), for halos with mass > mcut; 0 for the others.  pos: (n,3) in [-Lbox/2,
 * Lbox/2) or wherever the caller has them: when periodic the kernels apply menv.py:39 `(pos + Lbox/2) % Lbox` in the
 * dtype of pos (NumPy remainder), so the distances are formed from the values the reference's tree sees.
 * r_inner / r_outer: n_inner / n_outer = 1 (scalar) or n values of precision r_f64; r_outer_max = their maximum (sets the
 * cell size).  periodic = 0: open geometry (`halo_lc`), lo / hi = bounding box of pos.  Menv: (n) float64.
 */
int abacus_menv(const void *pos, int pos_f64, const void *mass, int mass_f64, int64_t n, const void *r_inner,
                int64_t n_inner, const void *r_outer, int64_t n_outer, int r_f64, double r_outer_max, double Lbox,
                int periodic, const double *lo, const double *hi, double mcut, double *Menv);

#ifdef __cplusplus
}
#endif
#endif /* ABACUS_HIP_H */
