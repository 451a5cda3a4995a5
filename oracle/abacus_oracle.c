/*
 * CPU ORACLE - TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C (+OpenMP) restatement of the reference's algorithms for the hot path
 * (abacusorg/abacusutils, Python+Numba).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the shipped path
 * (abacusutils_amd/ + libabacus_hip.so) never does.
 *
 * Parity status: PINNED.  Checked in tests/test_oracle_*.py against
 *   - the reference's own fixtures (tests/ref_hod ECSV catalogs, tests/ref_tsc grids),
 *   - golden vectors produced by running the reference's functions in the build
 *     container (oracle/make_golden.py -> tests/golden/).
 * Pair counting (section at the end) restates Corrfunc's published binning
 * convention; Corrfunc is an un-vendored third-party dependency (Corrfunc>=2,
 * pyproject.toml:51) and no reference test covers it: that part is "parity
 * unpinned" (checked only against the brute-force counter in this file).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/abacusnbody/).  Arithmetic order and dtypes follow the
 * reference; build with -ffp-contract=off so no FMA contraction changes it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* HOD                                                                        */
/* ------------------------------------------------------------------------- */

/* flat form of the three numba typed dicts built in gen_gals (hod/GRAND_HOD.py:1342-1468) */
typedef struct {
    int32_t want_LRG, want_ELG, want_QSO;
    int32_t rsd, has_origin, enable_ranks;
    int32_t pad0, pad1;
    double inv_velz2kms, lbox, origin[3];
    /* LRG */
    double L_logM_cut, L_logM1, L_sigma, L_alpha, L_kappa, L_alpha_c, L_alpha_s;
    double L_s, L_s_v, L_s_p, L_s_r, L_Acent, L_Asat, L_Bcent, L_Bsat, L_ic;
    /* ELG */
    double E_p_max, E_Q, E_logM_cut, E_kappa, E_sigma, E_logM1, E_alpha, E_gamma, E_A_s;
    double E_alpha_c, E_alpha_s, E_s, E_s_v, E_s_p, E_s_r;
    double E_Acent, E_Asat, E_Bcent, E_Bsat, E_Ccent, E_Csat, E_ic;
    double E_logM1_EE, E_alpha_EE, E_logM1_EL, E_alpha_EL;
    /* QSO */
    double Q_logM_cut, Q_kappa, Q_sigma, Q_logM1, Q_alpha, Q_alpha_c, Q_alpha_s;
    double Q_s, Q_s_v, Q_s_p, Q_s_r, Q_Acent, Q_Asat, Q_Bcent, Q_Bsat, Q_ic;
} oracle_hod_params;

/* hod/GRAND_HOD.py:37-42 */
static double n_cen_LRG(double M_h, double logM_cut, double sigma) {
    return 0.5 * erfc((logM_cut - log10(M_h)) / (1.41421356 * sigma));
}
/* hod/GRAND_HOD.py:23-34 */
static double n_sat_LRG_modified(double M_h, double logM_cut, double M_cut, double M_1, double sigma, double alpha,
                                 double kappa) {
    if (M_h - kappa * M_cut < 0) return 0;
    return pow((M_h - kappa * M_cut) / M_1, alpha) * 0.5 * erfc((logM_cut - log10(M_h)) / (1.41421356 * sigma));
}
/* hod/GRAND_HOD.py:45-52 (A_s default 1.0) and :55-65 */
static double N_sat_generic(double M_h, double M_cut, double kappa, double M_1, double alpha, double A_s) {
    if (M_h - kappa * M_cut < 0) return 0;
    return A_s * pow((M_h - kappa * M_cut) / M_1, alpha);
}
/* hod/GRAND_HOD.py:120-125 */
static double Gaussian_fun(double x, double mean, double sigma) {
    double d = x - mean;
    return 0.3989422804014327 / sigma * exp(-(d * d) / 2 / (sigma * sigma));
}
/* hod/GRAND_HOD.py:68-78,101-117 */
static double N_cen_ELG_v1(double M_h, double p_max, double Q, double logM_cut, double sigma, double gamma) {
    double logM_h = log10(M_h);
    double phi = Gaussian_fun(logM_h, logM_cut, sigma);
    double x = gamma * (logM_h - logM_cut) / sigma;
    double Phi = 0.5 * (1 + erf(x / sqrt(2.0)));
    return 2.0 * (p_max - 1.0 / Q) * phi * Phi / 1;
}
/* hod/GRAND_HOD.py:93-98 */
static double N_cen_QSO(double M_h, double logM_cut, double sigma) {
    return 0.5 * (1 + erf((log10(M_h) - logM_cut) / 1.41421356 / sigma));
}
/* hod/GRAND_HOD.py:128-136 */
static double wrap_box(double x, double L) {
    double L2 = L / 2;
    if (x >= L2) return x - L;
    if (x < -L2) return x + L;
    return x;
}

/* chunk starts: np.rint(np.linspace(0, H, Nthread+1)) (hod/GRAND_HOD.py:206-208) */
static void chunk_starts(int64_t H, int nthread, int64_t *hstart) {
    double step = (double)H / nthread;
    for (int t = 0; t < nthread; t++) hstart[t] = (int64_t)rint(t * step);
    hstart[nthread] = H;
}

/* centrals, pass 1: marker chain + keep mask + per-chunk counts (hod/GRAND_HOD.py:213-252) */
static int8_t cent_decide(const oracle_hod_params *p, double mass, double multis, double randoms, double deltac,
                          double fenv, double shear) {
    double LRG_marker = 0;
    if (p->want_LRG) {
        double lc = p->L_logM_cut + p->L_Acent * deltac + p->L_Bcent * fenv;
        LRG_marker += n_cen_LRG(mass, lc, p->L_sigma) * p->L_ic * multis;
    }
    double ELG_marker = LRG_marker;
    if (p->want_ELG) {
        double lc = p->E_logM_cut + p->E_Acent * deltac + p->E_Bcent * fenv + p->E_Ccent * shear;
        ELG_marker += N_cen_ELG_v1(mass, p->E_p_max, p->E_Q, lc, p->E_sigma, p->E_gamma) * p->E_ic * multis;
    }
    double QSO_marker = ELG_marker;
    if (p->want_QSO) {
        double lc = p->Q_logM_cut + p->Q_Acent * deltac + p->Q_Bcent * fenv;
        QSO_marker += N_cen_QSO(mass, lc, p->Q_sigma) * p->Q_ic * multis;
    }
    if (randoms <= LRG_marker) return 1;
    if (randoms <= ELG_marker) return 2;
    if (randoms <= QSO_marker) return 3;
    return 0;
}

/* satellites, pass 1 (hod/GRAND_HOD.py:957-1088) */
static int8_t sat_decide(const oracle_hod_params *p, double hmass, double weights, double randoms, double d, double f,
                         double sh, double r, double rv, double rp, double rr, int8_t keep_cent) {
    double LRG_marker = 0;
    if (p->want_LRG) {
        double M1 = pow(10.0, p->L_logM1 + p->L_Asat * d + p->L_Bsat * f);
        double lc = p->L_logM_cut + p->L_Acent * d + p->L_Bcent * f;
        double base = n_sat_LRG_modified(hmass, lc, pow(10.0, lc), M1, p->L_sigma, p->L_alpha, p->L_kappa) * weights *
                      p->L_ic;
        double exp_sat = base;
        if (p->enable_ranks) {
            double dec = 1 + p->L_s * r + p->L_s_v * rv + p->L_s_p * rp + p->L_s_r * rr;
            exp_sat = base * dec;
        }
        LRG_marker += exp_sat;
    }
    double ELG_marker = LRG_marker;
    if (p->want_ELG) {
        double M1 = pow(10.0, p->E_logM1 + p->E_Asat * d + p->E_Bsat * f + p->E_Csat * sh);
        double lc = p->E_logM_cut + p->E_Acent * d + p->E_Bcent * f + p->E_Ccent * sh;
        double base = N_sat_generic(hmass, pow(10.0, lc), p->E_kappa, M1, p->E_alpha, p->E_A_s) * weights * p->E_ic;
        if (keep_cent == 1) { /* ELG conformity :1006-1035 (no Csat term in these branches) */
            M1 = pow(10.0, p->E_logM1_EL + p->E_Asat * d + p->E_Bsat * f);
            base = N_sat_generic(hmass, pow(10.0, lc), p->E_kappa, M1, p->E_alpha_EL, p->E_A_s) * weights * p->E_ic;
        } else if (keep_cent == 2) {
            M1 = pow(10.0, p->E_logM1_EE + p->E_Asat * d + p->E_Bsat * f);
            base = N_sat_generic(hmass, pow(10.0, lc), p->E_kappa, M1, p->E_alpha_EE, p->E_A_s) * weights * p->E_ic;
        }
        if (p->enable_ranks) {
            double dec = 1 + p->E_s * r + p->E_s_v * rv + p->E_s_p * rp + p->E_s_r * rr;
            base = base * dec;
        }
        ELG_marker += base;
    }
    double QSO_marker = ELG_marker;
    if (p->want_QSO) {
        double M1 = pow(10.0, p->Q_logM1 + p->Q_Asat * d + p->Q_Bsat * f);
        double lc = p->Q_logM_cut + p->Q_Acent * d + p->Q_Bcent * f;
        double base = N_sat_generic(hmass, pow(10.0, lc), p->Q_kappa, M1, p->Q_alpha, 1.0) * weights * p->Q_ic;
        double exp_sat = base;
        if (p->enable_ranks) {
            double dec = 1 + p->Q_s * r + p->Q_s_v * rv + p->Q_s_p * rp + p->Q_s_r * rr;
            exp_sat = base * dec;
        }
        QSO_marker += exp_sat;
    }
    if (randoms <= LRG_marker) return 1;
    if (randoms <= ELG_marker) return 2;
    if (randoms <= QSO_marker) return 3;
    return 0;
}

/* galaxy emission shared by centrals (:298-325) and satellites (:1134-1165) */
static void emit_one(const oracle_hod_params *p, double x, double y, double z, double vx, double vy, double vz,
                     double mass, int64_t id, double **out, int64_t *idout, int64_t j) {
    if (p->rsd && p->has_origin) {
        double nx = x - p->origin[0], ny = y - p->origin[1], nz = z - p->origin[2];
        double inv_norm = 1.0 / sqrt(nx * nx + ny * ny + nz * nz);
        nx *= inv_norm;
        ny *= inv_norm;
        nz *= inv_norm;
        double proj = p->inv_velz2kms * (vx * nx + vy * ny + vz * nz);
        x = x + proj * nx;
        y = y + proj * ny;
        z = z + proj * nz;
    } else if (p->rsd) {
        z = wrap_box(z + vz * p->inv_velz2kms, p->lbox);
    }
    out[0][j] = x;
    out[1][j] = y;
    out[2][j] = z;
    out[3][j] = vx;
    out[4][j] = vy;
    out[5][j] = vz;
    out[6][j] = mass;
    idout[j] = id;
}

/*
 * gen_cent (hod/GRAND_HOD.py:139-414): two chunked passes.  Call with out == NULL to
 * run pass 1 only (fills keep[] and counts[3]); then allocate and call again with
 * out[t*7+c] (t tracer 0..2, c column x,y,z,vx,vy,vz,mass) and idout[t].
 */
int oracle_gen_cent(int64_t H, const double *pos, const double *vel, const double *mass, const int64_t *ids,
                    const double *multis, const double *randoms, const double *vdev, const double *deltac,
                    const double *fenv, const double *shear, const oracle_hod_params *p, int nthread, int8_t *keep,
                    int64_t *counts, double **out, int64_t **idout) {
    if (nthread < 1) nthread = 1;
    int64_t *hstart = malloc((nthread + 1) * sizeof(int64_t));
    int64_t *Nout = calloc((size_t)nthread * 3, sizeof(int64_t));
    chunk_starts(H, nthread, hstart);
#pragma omp parallel for num_threads(nthread) schedule(static, 1)
    for (int tid = 0; tid < nthread; tid++) {
        for (int64_t i = hstart[tid]; i < hstart[tid + 1]; i++) {
            int8_t k = cent_decide(p, mass[i], multis[i], randoms[i], deltac ? deltac[i] : 0.0, fenv ? fenv[i] : 0.0,
                                   shear ? shear[i] : 0.0);
            keep[i] = k;
            if (k) Nout[tid * 3 + (k - 1)]++;
        }
    }
    /* gstart = cumsum of per-chunk counts (:255-259) */
    int64_t *gstart = malloc((size_t)(nthread + 1) * 3 * sizeof(int64_t));
    gstart[0] = gstart[1] = gstart[2] = 0;
    for (int t = 0; t < nthread; t++)
        for (int c = 0; c < 3; c++) gstart[(t + 1) * 3 + c] = gstart[t * 3 + c] + Nout[t * 3 + c];
    for (int c = 0; c < 3; c++) counts[c] = gstart[nthread * 3 + c];
    if (out) {
        const double ac[3] = {p->L_alpha_c, p->E_alpha_c, p->Q_alpha_c};
#pragma omp parallel for num_threads(nthread) schedule(static, 1)
        for (int tid = 0; tid < nthread; tid++) {
            int64_t j[3] = {gstart[tid * 3], gstart[tid * 3 + 1], gstart[tid * 3 + 2]};
            for (int64_t i = hstart[tid]; i < hstart[tid + 1]; i++) {
                int t = keep[i] - 1;
                if (t < 0) continue;
                double vx = vel[3 * i] + ac[t] * vdev[3 * i];
                double vy = vel[3 * i + 1] + ac[t] * vdev[3 * i + 1];
                double vz = vel[3 * i + 2] + ac[t] * vdev[3 * i + 2];
                emit_one(p, pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], vx, vy, vz, mass[i], ids[i], out + 7 * t,
                         idout[t], j[t]);
                j[t]++;
            }
        }
    }
    free(hstart);
    free(Nout);
    free(gstart);
    return 0;
}

/* gen_sats (hod/GRAND_HOD.py:825-1262); keep_cent is keep_cent[pinds] gathered by the caller (:1562) */
int oracle_gen_sats(int64_t H, const double *ppos, const double *pvel, const double *hvel, const double *hmass,
                    const int64_t *hid, const double *weights, const double *randoms, const double *hdeltac,
                    const double *hfenv, const double *hshear, const double *ranks, const double *ranksv,
                    const double *ranksp, const double *ranksr, const int8_t *keep_cent, const oracle_hod_params *p,
                    int nthread, int8_t *keep, int64_t *counts, double **out, int64_t **idout) {
    if (nthread < 1) nthread = 1;
    int64_t *hstart = malloc((nthread + 1) * sizeof(int64_t));
    int64_t *Nout = calloc((size_t)nthread * 3, sizeof(int64_t));
    chunk_starts(H, nthread, hstart);
#pragma omp parallel for num_threads(nthread) schedule(static, 1)
    for (int tid = 0; tid < nthread; tid++) {
        for (int64_t i = hstart[tid]; i < hstart[tid + 1]; i++) {
            int8_t k = sat_decide(p, hmass[i], weights[i], randoms[i], hdeltac ? hdeltac[i] : 0.0,
                                  hfenv ? hfenv[i] : 0.0, hshear ? hshear[i] : 0.0, ranks ? ranks[i] : 1.0,
                                  ranksv ? ranksv[i] : 1.0, ranksp ? ranksp[i] : 1.0, ranksr ? ranksr[i] : 1.0,
                                  keep_cent ? keep_cent[i] : 0);
            keep[i] = k;
            if (k) Nout[tid * 3 + (k - 1)]++;
        }
    }
    int64_t *gstart = malloc((size_t)(nthread + 1) * 3 * sizeof(int64_t));
    gstart[0] = gstart[1] = gstart[2] = 0;
    for (int t = 0; t < nthread; t++)
        for (int c = 0; c < 3; c++) gstart[(t + 1) * 3 + c] = gstart[t * 3 + c] + Nout[t * 3 + c];
    for (int c = 0; c < 3; c++) counts[c] = gstart[nthread * 3 + c];
    if (out) {
        const double as[3] = {p->L_alpha_s, p->E_alpha_s, p->Q_alpha_s};
#pragma omp parallel for num_threads(nthread) schedule(static, 1)
        for (int tid = 0; tid < nthread; tid++) {
            int64_t j[3] = {gstart[tid * 3], gstart[tid * 3 + 1], gstart[tid * 3 + 2]};
            for (int64_t i = hstart[tid]; i < hstart[tid + 1]; i++) {
                int t = keep[i] - 1;
                if (t < 0) continue;
                double vx = hvel[3 * i] + as[t] * (pvel[3 * i] - hvel[3 * i]);
                double vy = hvel[3 * i + 1] + as[t] * (pvel[3 * i + 1] - hvel[3 * i + 1]);
                double vz = hvel[3 * i + 2] + as[t] * (pvel[3 * i + 2] - hvel[3 * i + 2]);
                emit_one(p, ppos[3 * i], ppos[3 * i + 1], ppos[3 * i + 2], vx, vy, vz, hmass[i], hid[i], out + 7 * t,
                         idout[t], j[t]);
                j[t]++;
            }
        }
    }
    free(hstart);
    free(Nout);
    free(gstart);
    return 0;
}

/* keep_cent[pinds] (hod/GRAND_HOD.py:1562: a NumPy fancy-index gather on the host); threaded here so that the
 * CPU baseline of bench.py times compiled code only */
void oracle_gather_i8(const int8_t *src, const int64_t *idx, int64_t n, int8_t *dst, int nthread) {
    if (nthread < 1) nthread = 1;
#pragma omp parallel for num_threads(nthread) schedule(static)
    for (int64_t i = 0; i < n; i++) dst[i] = src[idx[i]];
}

/* ------------------------------------------------------------------------- */
/* TSC / CIC / partition                                                      */
/* ------------------------------------------------------------------------- */

/* _wrap_inplace (analysis/tsc.py:219-226), single +-box shift, in the position dtype */
void oracle_wrap_inplace_f32(float *pos, int64_t n, double box) {
#pragma omp parallel for
    for (int64_t i = 0; i < 3 * n; i++) {
        if (pos[i] >= box) pos[i] -= box; /* numpy: f32 (op)= python float -> f32 arithmetic */
        else if (pos[i] < 0) pos[i] += box;
    }
}
void oracle_wrap_inplace_f64(double *pos, int64_t n, double box) {
#pragma omp parallel for
    for (int64_t i = 0; i < 3 * n; i++) {
        if (pos[i] >= box) pos[i] -= box;
        else if (pos[i] < 0) pos[i] += box;
    }
}

static inline int rightwrap(int x, int L) { /* analysis/tsc.py:387-391 + numpy negative indexing */
    if (x >= L) return x - L;
    if (x < 0) return x + L;
    return x;
}

/* _tsc_scatter (analysis/tsc.py:394-507): math in the position dtype FT, accumulate into grid dtype GT */
#define DEFINE_TSC_SCATTER(NAME, FT, GT, RINT)                                                                  \
    void NAME(const FT *positions, int64_t n, GT *density, int gx, int gy, int gz, double boxsize,              \
              const FT *weights, double offset_) {                                                              \
        const FT inv_hx = (FT)(gx / boxsize), inv_hy = (FT)(gy / boxsize), inv_hz = (FT)(gz / boxsize);         \
        const FT offset = (FT)offset_;                                                                          \
        const FT HALF = (FT)0.5, P75 = (FT)0.75;                                                                \
        FT W = (FT)1.0;                                                                                         \
        for (int64_t i = 0; i < n; i++) {                                                                       \
            if (weights) W = weights[i];                                                                        \
            FT px = (positions[3 * i] + offset) * inv_hx;                                                       \
            FT py = (positions[3 * i + 1] + offset) * inv_hy;                                                   \
            FT pz = (positions[3 * i + 2] + offset) * inv_hz;                                                   \
            int ix = (int16_t)RINT(px), iy = (int16_t)RINT(py), iz = (int16_t)RINT(pz);                         \
            FT dx = (FT)ix - px, dy = (FT)iy - py, dz = (FT)iz - pz;                                            \
            FT wx[3], wy[3], wz[3];                                                                             \
            wx[1] = P75 - dx * dx;                                                                              \
            wx[0] = HALF * ((HALF + dx) * (HALF + dx));                                                         \
            wx[2] = HALF * ((HALF - dx) * (HALF - dx));                                                         \
            wy[1] = P75 - dy * dy;                                                                              \
            wy[0] = HALF * ((HALF + dy) * (HALF + dy));                                                         \
            wy[2] = HALF * ((HALF - dy) * (HALF - dy));                                                         \
            wz[1] = P75 - dz * dz;                                                                              \
            wz[0] = HALF * ((HALF + dz) * (HALF + dz));                                                         \
            wz[2] = HALF * ((HALF - dz) * (HALF - dz));                                                         \
            int jx[3] = {rightwrap(ix - 1, gx), rightwrap(ix, gx), rightwrap(ix + 1, gx)};                      \
            int jy[3] = {rightwrap(iy - 1, gy), rightwrap(iy, gy), rightwrap(iy + 1, gy)};                      \
            int jz[3] = {rightwrap(iz - 1, gz), rightwrap(iz, gz), rightwrap(iz + 1, gz)};                      \
            /* same 27 updates; the reference does the 9 centre-z cells first (:471-479) then the +-z ones */   \
            for (int a = 0; a < 3; a++)                                                                         \
                for (int b = 0; b < 3; b++) {                                                                   \
                    GT *row = density + ((int64_t)jx[a] * gy + jy[b]) * gz;                                     \
                    row[jz[1]] = (GT)(row[jz[1]] + wx[a] * wy[b] * wz[1] * W);                                  \
                }                                                                                               \
            for (int a = 0; a < 3; a++)                                                                         \
                for (int b = 0; b < 3; b++) {                                                                   \
                    GT *row = density + ((int64_t)jx[a] * gy + jy[b]) * gz;                                     \
                    row[jz[0]] = (GT)(row[jz[0]] + wx[a] * wy[b] * wz[0] * W);                                  \
                    row[jz[2]] = (GT)(row[jz[2]] + wx[a] * wy[b] * wz[2] * W);                                  \
                }                                                                                               \
        }                                                                                                       \
    }
DEFINE_TSC_SCATTER(oracle_tsc_scatter_f32_f32, float, float, rintf)
DEFINE_TSC_SCATTER(oracle_tsc_scatter_f64_f32, double, float, rint)
DEFINE_TSC_SCATTER(oracle_tsc_scatter_f32_f64, float, double, rintf)
DEFINE_TSC_SCATTER(oracle_tsc_scatter_f64_f64, double, double, rint)

/* cic_serial (analysis/cic.py:13-125): float64 math whatever the input dtype; no wrap of positions */
#define DEFINE_CIC(NAME, FT)                                                                                    \
    void NAME(const FT *positions, int64_t n, float *density, int gx, int gy, int gz, double boxsize,           \
              const FT *weights) {                                                                              \
        double W = 1.0;                                                                                         \
        for (int64_t i = 0; i < n; i++) {                                                                       \
            if (weights) W = weights[i];                                                                        \
            double px = (positions[3 * i] / boxsize) * gx;                                                      \
            double py = (positions[3 * i + 1] / boxsize) * gy;                                                  \
            double pz = (positions[3 * i + 2] / boxsize) * gz;                                                  \
            int ix = (int)rint(px), iy = (int)rint(py), iz = (int)rint(pz);                                     \
            double d[3] = {ix - px, iy - py, iz - pz};                                                          \
            double w[3][3];                                                                                     \
            for (int a = 0; a < 3; a++) {                                                                       \
                w[a][1] = 1.0 - fabs(d[a]);                                                                     \
                if (d[a] > 0.0) {                                                                               \
                    w[a][0] = d[a];                                                                             \
                    w[a][2] = 0.0;                                                                              \
                } else {                                                                                        \
                    w[a][2] = -d[a];                                                                            \
                    w[a][0] = 0.0;                                                                              \
                }                                                                                               \
            }                                                                                                   \
            int jx[3] = {rightwrap(ix - 1, gx), rightwrap(ix, gx), rightwrap(ix + 1, gx)};                      \
            int jy[3] = {rightwrap(iy - 1, gy), rightwrap(iy, gy), rightwrap(iy + 1, gy)};                      \
            int jz[3] = {rightwrap(iz - 1, gz), rightwrap(iz, gz), rightwrap(iz + 1, gz)};                      \
            for (int a = 0; a < 3; a++)                                                                         \
                for (int b = 0; b < 3; b++) {                                                                   \
                    float *row = density + ((int64_t)jx[a] * gy + jy[b]) * gz;                                  \
                    row[jz[1]] = (float)(row[jz[1]] + w[0][a] * w[1][b] * w[2][1] * W);                         \
                }                                                                                               \
            for (int a = 0; a < 3; a++)                                                                         \
                for (int b = 0; b < 3; b++) {                                                                   \
                    float *row = density + ((int64_t)jx[a] * gy + jy[b]) * gz;                                  \
                    row[jz[0]] = (float)(row[jz[0]] + w[0][a] * w[1][b] * w[2][0] * W);                         \
                    row[jz[2]] = (float)(row[jz[2]] + w[0][a] * w[1][b] * w[2][2] * W);                         \
                }                                                                                               \
        }                                                                                                       \
    }
DEFINE_CIC(oracle_cic_f32, float)
DEFINE_CIC(oracle_cic_f64, double)

/* partition_parallel (analysis/tsc.py:259-384): stable counting sort by stripe key */
#define DEFINE_PARTITION(NAME, FT)                                                                              \
    void NAME(const FT *pos, int64_t n, int npartition, double boxsize, const FT *weights, int coord,           \
              int nthread, FT *psort, int64_t *starts, FT *wsort) {                                             \
        if (nthread < 1) nthread = 1;                                                                           \
        const FT inv_pwidth = (FT)(npartition / boxsize);                                                       \
        int32_t *keys = malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));                                      \
        int64_t *counts = calloc((size_t)nthread * npartition, sizeof(int64_t));                                \
        int64_t *tstart = malloc((nthread + 1) * sizeof(int64_t));                                              \
        for (int t = 0; t <= nthread; t++) tstart[t] = (int64_t)((double)t * ((double)n / nthread));            \
        tstart[nthread] = n;                                                                                    \
        _Pragma("omp parallel for num_threads(nthread) schedule(static,1)")                                     \
        for (int t = 0; t < nthread; t++)                                                                       \
            for (int64_t i = tstart[t]; i < tstart[t + 1]; i++) {                                               \
                int32_t k = (int32_t)(pos[3 * i + coord] * inv_pwidth);                                         \
                if (k > npartition - 1) k = npartition - 1;                                                     \
                keys[i] = k;                                                                                    \
                counts[(size_t)t * npartition + k]++;                                                           \
            }                                                                                                   \
        /* pointers: cumsum over (partition-major, thread-minor) (:339-342) */                                  \
        int64_t run = 0;                                                                                        \
        for (int k = 0; k < npartition; k++) {                                                                  \
            starts[k] = run;                                                                                    \
            for (int t = 0; t < nthread; t++) {                                                                 \
                int64_t c = counts[(size_t)t * npartition + k];                                                 \
                counts[(size_t)t * npartition + k] = run;                                                       \
                run += c;                                                                                       \
            }                                                                                                   \
        }                                                                                                       \
        starts[npartition] = n;                                                                                 \
        _Pragma("omp parallel for num_threads(nthread) schedule(static,1)")                                     \
        for (int t = 0; t < nthread; t++)                                                                       \
            for (int64_t i = tstart[t]; i < tstart[t + 1]; i++) {                                               \
                int64_t s = counts[(size_t)t * npartition + keys[i]]++;                                         \
                psort[3 * s] = pos[3 * i];                                                                      \
                psort[3 * s + 1] = pos[3 * i + 1];                                                              \
                psort[3 * s + 2] = pos[3 * i + 2];                                                              \
                if (weights) wsort[s] = weights[i];                                                             \
            }                                                                                                   \
        free(keys);                                                                                             \
        free(counts);                                                                                           \
        free(tstart);                                                                                           \
    }
DEFINE_PARTITION(oracle_partition_f32, float)
DEFINE_PARTITION(oracle_partition_f64, double)

/* _tsc_parallel (analysis/tsc.py:229-256): even stripes in parallel, then odd stripes */
#define DEFINE_TSC_PARALLEL(NAME, SCATTER, FT, GT)                                                              \
    void NAME(const FT *ppart, const int64_t *starts, int npartition, GT *dens, int gx, int gy, int gz,         \
              double box, const FT *weights, double offset, int nthread) {                                      \
        if (nthread < 1) nthread = 1;                                                                           \
        for (int parity = 0; parity < 2; parity++) {                                                            \
            if (parity == 1 && npartition <= 1) break;                                                          \
            _Pragma("omp parallel for num_threads(nthread) schedule(dynamic,1)")                                \
            for (int i = 0; i < (npartition + 1) / 2; i++) {                                                    \
                int s = 2 * i + parity;                                                                         \
                if (s >= npartition) continue;                                                                  \
                SCATTER(ppart + 3 * starts[s], starts[s + 1] - starts[s], dens, gx, gy, gz, box,                \
                        weights ? weights + starts[s] : NULL, offset);                                          \
            }                                                                                                   \
        }                                                                                                       \
    }
DEFINE_TSC_PARALLEL(oracle_tsc_parallel_f32_f32, oracle_tsc_scatter_f32_f32, float, float)
DEFINE_TSC_PARALLEL(oracle_tsc_parallel_f64_f32, oracle_tsc_scatter_f64_f32, double, float)

/* normalize_field (analysis/power_spectrum.py:860-901): delta = rho*f32(M/N) - 1 in place */
void oracle_normalize_field_f32(float *field, int64_t size, double tot_weight) {
    const float norm = (float)((double)size / tot_weight);
#pragma omp parallel for
    for (int64_t i = 0; i < size; i++) field[i] = field[i] * norm - 1.0f;
}

/* ------------------------------------------------------------------------- */
/* spectrum post-processing                                                   */
/* ------------------------------------------------------------------------- */
typedef struct {
    float re, im;
} c64;

/* _normalize (analysis/power_spectrum.py:1073-1078): complex64 *= float32 */
void oracle_scale_c64(c64 *f, int64_t n, float a) {
#pragma omp parallel for
    for (int64_t i = 0; i < n; i++) {
        f[i].re *= a;
        f[i].im *= a;
    }
}

/* shift_field_fft (analysis/power_spectrum.py:904-948).  exp(fac*(kx+ky+kz)) with fac = f32(0.5*d)*1j is a
 * complex64 exp of a pure-imaginary argument: (cosf(t), sinf(t)), t = f32(0.5 d)*(kx+ky+kz) in float32. */
void oracle_shift_field_fft(c64 *f, const c64 *fs, int n1d, double L, double d_) {
    const int kzlen = n1d / 2 + 1;
    const float dk = (float)(2.0 * M_PI / L);
    const float d = (float)d_;
    const float norm = (float)(0.5 / ((double)n1d * n1d * n1d));
    const float fac = (float)(0.5 * d); /* dtype(0.5*d) with d already f32 -> computed in python float then cast */
#pragma omp parallel for
    for (int i = 0; i < n1d; i++) {
        float kx = i < n1d / 2 ? (float)i * dk : (float)(i - n1d) * dk;
        for (int j = 0; j < n1d; j++) {
            float ky = j < n1d / 2 ? (float)j * dk : (float)(j - n1d) * dk;
            for (int k = 0; k < kzlen; k++) {
                float kz = (float)k * dk;
                float t = fac * (kx + ky + kz);
                float c = cosf(t), s = sinf(t);
                int64_t idx = ((int64_t)i * n1d + j) * kzlen + k;
                c64 a = fs[idx];
                float re = a.re * c - a.im * s, im = a.re * s + a.im * c;
                re = f[idx].re + re;
                im = f[idx].im + im;
                f[idx].re = re * norm;
                f[idx].im = im * norm;
            }
        }
    }
}

/* compensation divide (analysis/power_spectrum.py:1063-1069): f /= (Wx*Wy)*Wz in float32, complex/real */
void oracle_compensate(c64 *f, int n1d, const float *W) {
    const int kzlen = n1d / 2 + 1;
#pragma omp parallel for
    for (int i = 0; i < n1d; i++)
        for (int j = 0; j < n1d; j++) {
            float wxy = W[i] * W[j];
            for (int k = 0; k < kzlen; k++) {
                float w = wxy * W[k];
                int64_t idx = ((int64_t)i * n1d + j) * kzlen + k;
                /* NumPy's complex64/complex64 loop with a zero imaginary divisor (Smith's algorithm):
                 * rat = 0/w; scl = 1/(w + 0*rat); out = ((re + im*rat)*scl, (im - re*rat)*scl) */
                float scl = 1.0f / w;
                f[idx].re = f[idx].re * scl;
                f[idx].im = f[idx].im * scl;
            }
        }
}

/* get_raw_power (analysis/power_spectrum.py:707-727) */
void oracle_raw_power(const c64 *f, const c64 *f2, int64_t n, float *out) {
#pragma omp parallel for
    for (int64_t i = 0; i < n; i++) {
        if (f2) out[i] = f[i].re * f2[i].re + f[i].im * f2[i].im; /* Re(conj(a)*b) */
        else {
            /* np.abs(c64)**2: hypot in float32, then squared */
            float a = hypotf(f[i].re, f[i].im);
            out[i] = a * a;
        }
    }
}

static const int64_t FACT[21] = {1,
                                 1,
                                 2,
                                 6,
                                 24,
                                 120,
                                 720,
                                 5040,
                                 40320,
                                 362880,
                                 3628800,
                                 39916800,
                                 479001600,
                                 6227020800,
                                 87178291200,
                                 1307674368000,
                                 20922789888000,
                                 355687428096000,
                                 6402373705728000,
                                 121645100408832000,
                                 2432902008176640000};
static int64_t n_choose_k(int n, int k) { return FACT[n] / (FACT[k] * FACT[n - k]); }
/* P_n (analysis/power_spectrum.py:121-147): Legendre P_n in powers of x = mu^2, float32 */
static float P_n_f32(float x, int n) {
    float sum = 0.0f;
    for (int k = 0; k <= n / 2; k++) {
        float factor = (float)(n_choose_k(n, k) * n_choose_k(2 * n - 2 * k, n));
        float term = factor * powf(x, (float)(0.5 * (n - 2 * k)));
        if (k % 2 == 0) sum += term;
        else sum -= term;
    }
    sum *= (float)pow(0.5, n);
    return sum;
}

/*
 * bin_kmu (analysis/power_spectrum.py:150-300), fourier=True/False.
 * accum64 = 0: float32 accumulators in the reference's nthread=1 order (bitwise target for goldens).
 * accum64 = 1: float64 accumulators, OpenMP over kx planes (the yardstick for the GPU path).
 * Outputs: power[Nk*Nmu] f32, counts[Nk*Nmu] i64, poles[Np*Nk] f32, counts_poles[Nk] i64, kavg[Nk*Nmu] f32.
 */
/* weight of mode (i, j, k) when the spectrum is known in closed form: |amp * sum_t X_t[i] Y_t[j] Z_t[k]|^2 /
 * (cx[i] cy[j] cz[k])^2 - a sum of T separable complex terms (tabs: [T][3][n1d] complex128 as re, im pairs), an optional
 * real separable divisor (comp: [3][n1d] or NULL).  Used for the analytic known answers of the full-size tests: the
 * discrete Fourier transform of the TSC / CIC cloud of a handful of particles is such a sum (one term per particle). */
typedef struct {
    int T;
    const double *tabs;
    const double *comp;
    double amp;
} sep_spec;

static inline float sep_weight(const sep_spec *sp, int n1d, int i, int j, int k) {
    double re = 0, im = 0;
    for (int t = 0; t < sp->T; t++) {
        const double *X = sp->tabs + ((size_t)(t * 3 + 0) * n1d + i) * 2;
        const double *Y = sp->tabs + ((size_t)(t * 3 + 1) * n1d + j) * 2;
        const double *Z = sp->tabs + ((size_t)(t * 3 + 2) * n1d + k) * 2;
        const double xr = X[0] * Y[0] - X[1] * Y[1], xi = X[0] * Y[1] + X[1] * Y[0];
        re += xr * Z[0] - xi * Z[1];
        im += xr * Z[1] + xi * Z[0];
    }
    re *= sp->amp, im *= sp->amp;
    if (sp->comp) {
        const double c = sp->comp[i] * sp->comp[n1d + j] * sp->comp[2 * (size_t)n1d + k];
        re /= c, im /= c;
    }
    return (float)(re * re + im * im);
}

static int bin_kmu_core(int n1d, double L, const double *kedges, int Nk, const double *muedges, int Nmu,
                        const float *weights, const sep_spec *sep, const int64_t *poles, int Np, int fourier, int accum64,
                        int nthread, float *power, int64_t *counts, float *binned_poles, int64_t *counts_poles, float *kavg);

int oracle_bin_kmu(int n1d, double L, const double *kedges, int Nk, const double *muedges, int Nmu,
                   const float *weights, const int64_t *poles, int Np, int fourier, int accum64, int nthread,
                   float *power, int64_t *counts, float *binned_poles, int64_t *counts_poles, float *kavg) {
    return bin_kmu_core(n1d, L, kedges, Nk, muedges, Nmu, weights, NULL, poles, Np, fourier, accum64, nthread, power, counts,
                        binned_poles, counts_poles, kavg);
}

/* bin_kmu (float64 accumulation) of a spectrum given as a sum of separable terms: nothing of mesh size is stored, so the
 * 2048^3 known answers cost seconds on the host cores */
int oracle_bin_kmu_separable(int n1d, double L, const double *kedges, int Nk, const double *muedges, int Nmu, int T,
                             const double *tabs, const double *comp, double amp, const int64_t *poles, int Np, int nthread,
                             float *power, int64_t *counts, float *binned_poles, int64_t *counts_poles, float *kavg) {
    sep_spec sp = {T, tabs, comp, amp};
    return bin_kmu_core(n1d, L, kedges, Nk, muedges, Nmu, NULL, &sp, poles, Np, 1, 1, nthread, power, counts, binned_poles,
                        counts_poles, kavg);
}

static int bin_kmu_core(int n1d, double L, const double *kedges, int Nk, const double *muedges, int Nmu,
                        const float *weights, const sep_spec *sep, const int64_t *poles, int Np, int fourier, int accum64,
                        int nthread, float *power, int64_t *counts, float *binned_poles, int64_t *counts_poles, float *kavg) {
    const int kzlen = n1d / 2 + 1;
    const double dk = fourier ? 2.0 * M_PI / L : L / n1d;
    float *kedges2 = malloc((Nk + 1) * sizeof(float));
    float *muedges2 = malloc((Nmu + 1) * sizeof(float));
    for (int b = 0; b <= Nk; b++) kedges2[b] = (float)((kedges[b] / dk) * (kedges[b] / dk));
    for (int b = 0; b <= Nmu; b++) muedges2[b] = (float)(muedges[b] * muedges[b]);
    if (nthread < 1) nthread = 1;
    if (!accum64) nthread = 1;
    const size_t nb = (size_t)Nk * Nmu, npk = (size_t)Np * Nk;
    int64_t *cnt = calloc(nb * nthread, sizeof(int64_t));
    double *wc64 = calloc(nb * nthread, sizeof(double)), *wk64 = calloc(nb * nthread, sizeof(double));
    double *wp64 = calloc((npk ? npk : 1) * nthread, sizeof(double));
    float *wc32 = calloc(nb, sizeof(float)), *wk32 = calloc(nb, sizeof(float));
    float *wp32 = calloc(npk ? npk : 1, sizeof(float));
#pragma omp parallel for num_threads(nthread) schedule(static)
    for (int i = 0; i < n1d; i++) {
#ifdef _OPENMP
        int tid = omp_get_thread_num();
#else
        int tid = 0;
#endif
        int64_t i2 = i < n1d / 2 ? (int64_t)i * i : (int64_t)(i - n1d) * (i - n1d);
        for (int j = 0; j < n1d; j++) {
            int bk = 0, bmu = 0;
            int64_t j2 = j < n1d / 2 ? (int64_t)j * j : (int64_t)(j - n1d) * (j - n1d);
            for (int k = 0; k < kzlen; k++) {
                float kmag2 = (float)(i2 + j2 + (int64_t)k * k);
                float mu2;
                if (kmag2 > 0) {
                    float inv = 1.0f / kmag2; /* kmag2**-1 */
                    mu2 = (float)((int64_t)k * k) * inv;
                } else
                    mu2 = 0.0f;
                if (kmag2 < kedges2[0]) continue;
                if (kmag2 >= kedges2[Nk]) break;
                while (kmag2 > kedges2[bk + 1]) bk++;
                while (bmu + 1 < Nmu && mu2 > muedges2[bmu + 1]) bmu++;
                float w = weights ? weights[((int64_t)i * n1d + j) * kzlen + k] : sep_weight(sep, n1d, i, j, k);
                size_t b = (size_t)bk * Nmu + bmu;
                cnt[tid * nb + b] += k == 0 ? 1 : 2;
                if (accum64) {
                    wc64[tid * nb + b] += k == 0 ? (double)w : 2.0 * (double)w;
                    wk64[tid * nb + b] += k == 0 ? sqrt((double)kmag2) * dk : 2.0 * sqrt((double)kmag2) * dk;
                } else {
                    wc32[b] += k == 0 ? w : 2.0f * w;
                    /* np.sqrt(f32)*dk(f64): float64 product under numba typing, stored into a float32 accumulator */
                    double kv = k == 0 ? (double)sqrtf(kmag2) * dk : (double)(2.0f * sqrtf(kmag2)) * dk;
                    wk32[b] = (float)((double)wk32[b] + kv);
                }
                for (int ip = 0; ip < Np; ip++) {
                    int pole = (int)poles[ip];
                    if (pole != 0) {
                        float pw = (float)(2 * pole + 1) * P_n_f32(mu2, pole);
                        if (accum64)
                            wp64[tid * npk + (size_t)ip * Nk + bk] +=
                                k == 0 ? (double)w * (double)pw : 2.0 * (double)w * (double)pw;
                        else
                            wp32[(size_t)ip * Nk + bk] += k == 0 ? w * pw : 2.0f * w * pw;
                    }
                }
            }
        }
    }
    /* reductions and normalisation (:276-293) */
    for (size_t b = 0; b < nb; b++) {
        int64_t c = 0;
        double s = 0, sk = 0;
        for (int t = 0; t < nthread; t++) {
            c += cnt[t * nb + b];
            s += wc64[t * nb + b];
            sk += wk64[t * nb + b];
        }
        counts[b] = c;
        if (accum64) {
            wc64[b] = s;
            wk64[b] = sk;
        }
    }
    for (int i = 0; i < Nk; i++) {
        int64_t c = 0;
        for (int j = 0; j < Nmu; j++) c += counts[(size_t)i * Nmu + j];
        counts_poles[i] = c;
    }
    for (int ip = 0; ip < Np; ip++)
        for (int i = 0; i < Nk; i++) {
            size_t o = (size_t)ip * Nk + i;
            if (accum64) {
                double s = 0;
                for (int t = 0; t < nthread; t++) s += wp64[t * npk + o];
                if (poles[ip] == 0) {
                    s = 0;
                    for (int j = 0; j < Nmu; j++) s += wc64[(size_t)i * Nmu + j];
                }
                binned_poles[o] = (float)(counts_poles[i] ? s / (double)counts_poles[i] : s);
            } else {
                float s = wp32[o];
                if (poles[ip] == 0) {
                    s = 0;
                    for (int j = 0; j < Nmu; j++) s += wc32[(size_t)i * Nmu + j]; /* np.sum over mu, f32 */
                }
                binned_poles[o] = counts_poles[i] ? s / (float)counts_poles[i] : s;
            }
        }
    for (size_t b = 0; b < nb; b++) {
        if (accum64) {
            power[b] = (float)(counts[b] ? wc64[b] / (double)counts[b] : wc64[b]);
            kavg[b] = (float)(counts[b] ? wk64[b] / (double)counts[b] : wk64[b]);
        } else {
            power[b] = counts[b] ? wc32[b] / (float)counts[b] : wc32[b];
            kavg[b] = counts[b] ? wk32[b] / (float)counts[b] : wk32[b];
        }
    }
    free(kedges2);
    free(muedges2);
    free(cnt);
    free(wc64);
    free(wk64);
    free(wp64);
    free(wc32);
    free(wk32);
    free(wp32);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* pair counting  -- PARITY UNPINNED (see header)                             */
/* ------------------------------------------------------------------------- */
/*
 * Restates the conventions of Corrfunc.theory.DD / DDrppi / DDsmu (Corrfunc >= 2, third party, not in
 * /root/reference) as used at analysis/tpcf_corrfunc.py:144-156,240-252 and
 * scripts/emulator/generate_cfs/generate_cf.py:65-74: float32 coordinates, periodic minimum image,
 * r-bin b holds rmin_b <= r < rmax_b (compared on squared separations), pi-bin = floor(|dz|/dpi) with
 * dpi = 1 and |dz| < pimax, mu = |dz|/s, mu-bin = floor(mu*nmu/mu_max) with mu < mu_max;
 * autocorrelations count ordered pairs (i,j) and (j,i), self pairs excluded.
 * mode: 0 = DD(r), 1 = DDrppi, 2 = DDsmu.  npairs has nbins*(1|npibins|nmubins) entries.
 * Brute force O(N1*N2); the O(N) cell-list version lives in the product, this is its checker.
 */
int oracle_paircount_brute(int mode, int autocorr, const float *x1, const float *y1, const float *z1, int64_t n1,
                           const float *x2, const float *y2, const float *z2, int64_t n2, float boxsize,
                           const float *bins, int nbins, float pimax, int npibins, float mu_max, int nmubins,
                           int nthread, uint64_t *npairs) {
    if (autocorr) {
        x2 = x1;
        y2 = y1;
        z2 = z1;
        n2 = n1;
    }
    const int nsub = mode == 0 ? 1 : (mode == 1 ? npibins : nmubins);
    const size_t ntot = (size_t)nbins * nsub;
    if (nthread < 1) nthread = 1;
    uint64_t *loc = calloc(ntot * nthread, sizeof(uint64_t));
    float *b2 = malloc((nbins + 1) * sizeof(float));
    for (int b = 0; b <= nbins; b++) b2[b] = bins[b] * bins[b];
    const float half = boxsize * 0.5f;
    const float dpi = npibins > 0 ? pimax / (float)npibins : 1.0f;
    const float inv_dmu = nmubins > 0 ? (float)nmubins / mu_max : 1.0f;
#pragma omp parallel for num_threads(nthread) schedule(dynamic, 64)
    for (int64_t i = 0; i < n1; i++) {
#ifdef _OPENMP
        uint64_t *h = loc + (size_t)omp_get_thread_num() * ntot;
#else
        uint64_t *h = loc;
#endif
        for (int64_t j = 0; j < n2; j++) {
            if (autocorr && i == j) continue;
            float dx = x1[i] - x2[j], dy = y1[i] - y2[j], dz = z1[i] - z2[j];
            if (dx > half) dx -= boxsize;
            else if (dx < -half) dx += boxsize;
            if (dy > half) dy -= boxsize;
            else if (dy < -half) dy += boxsize;
            if (dz > half) dz -= boxsize;
            else if (dz < -half) dz += boxsize;
            float r2;
            int sub = 0;
            if (mode == 1) {
                float adz = fabsf(dz);
                if (adz >= pimax) continue;
                r2 = dx * dx + dy * dy;
                sub = (int)(adz / dpi);
                if (sub >= npibins) continue;
            } else {
                r2 = dx * dx + dy * dy + dz * dz;
            }
            if (r2 < b2[0] || r2 >= b2[nbins]) continue;
            int b = 0;
            while (r2 >= b2[b + 1]) b++;
            if (mode == 2) {
                float s = sqrtf(r2);
                float mu = s > 0 ? fabsf(dz) / s : 0.0f;
                if (mu >= mu_max) continue;
                sub = (int)(mu * inv_dmu);
                if (sub >= nmubins) continue;
            }
            h[(size_t)b * nsub + sub]++;
        }
    }
    for (size_t b = 0; b < ntot; b++) {
        uint64_t s = 0;
        for (int t = 0; t < nthread; t++) s += loc[(size_t)t * ntot + b];
        npairs[b] = s;
    }
    free(loc);
    free(b2);
    return 0;
}

/*
 * The same counts from a CELL LIST (cells >= the reach, 27-cell stencil, OpenMP over the cells of set 1, per-thread
 * histograms): the algorithm class of Corrfunc's kernels (without their AVX inner loops), used as the CPU baseline beside
 * the HIP pair counter in bench.py and held to the brute-force counter above by tests/test_oracle_pinned.py.  The per-pair
 * expressions are those of the brute-force loop, so the integers are identical.
 */
static inline int oracle_cell_of(float v, int nc, float inv_box) {
    float f = v * inv_box;
    f -= floorf(f);
    int c = (int)(f * (float)nc);
    return c >= nc ? nc - 1 : c;
}

int oracle_paircount_cells(int mode, int autocorr, const float *x1, const float *y1, const float *z1, int64_t n1,
                           const float *x2, const float *y2, const float *z2, int64_t n2, float boxsize,
                           const float *bins, int nbins, float pimax, int npibins, float mu_max, int nmubins,
                           int nthread, uint64_t *npairs) {
    if (autocorr) x2 = x1, y2 = y1, z2 = z1, n2 = n1;
    const int nsub = mode == 0 ? 1 : (mode == 1 ? npibins : nmubins);
    const size_t ntot = (size_t)nbins * nsub;
    if (nthread < 1) nthread = 1;
    const float rmax = bins[nbins], reach_z = mode == 1 ? pimax : rmax;
    const float inv_box = 1.0f / boxsize;
    int ncxy = (int)floorf(boxsize / rmax * 0.9999f), ncz = (int)floorf(boxsize / reach_z * 0.9999f);
    if (ncxy > 256) ncxy = 256;
    if (ncz > 256) ncz = 256;
    if (ncxy < 3) ncxy = 1;
    if (ncz < 3) ncz = 1;
    const int64_t ncell = (int64_t)ncxy * ncxy * ncz;
    /* counting sort of both sets into the cells */
    int64_t *start[2];
    float *sx[2], *sy[2], *sz[2];
    const float *px[2] = {x1, x2}, *py[2] = {y1, y2}, *pz[2] = {z1, z2};
    const int64_t pn[2] = {n1, n2};
    const int nset = autocorr ? 1 : 2;
    for (int s = 0; s < nset; s++) {
        const int64_t n = pn[s];
        start[s] = calloc((size_t)ncell + 1, sizeof(int64_t));
        int *cid = malloc((size_t)(n > 0 ? n : 1) * sizeof(int));
        sx[s] = malloc((size_t)(n > 0 ? n : 1) * 4), sy[s] = malloc((size_t)(n > 0 ? n : 1) * 4), sz[s] = malloc((size_t)(n > 0 ? n : 1) * 4);
        for (int64_t i = 0; i < n; i++) {
            cid[i] = (oracle_cell_of(px[s][i], ncxy, inv_box) * ncxy + oracle_cell_of(py[s][i], ncxy, inv_box)) * ncz +
                     oracle_cell_of(pz[s][i], ncz, inv_box);
            start[s][cid[i] + 1]++;
        }
        for (int64_t c = 0; c < ncell; c++) start[s][c + 1] += start[s][c];
        int64_t *cur = malloc((size_t)ncell * sizeof(int64_t));
        memcpy(cur, start[s], (size_t)ncell * sizeof(int64_t));
        for (int64_t i = 0; i < n; i++) {
            const int64_t d = cur[cid[i]]++;
            sx[s][d] = px[s][i], sy[s][d] = py[s][i], sz[s][d] = pz[s][i];
        }
        free(cur);
        free(cid);
    }
    const int T = autocorr ? 0 : 1;
    uint64_t *loc = calloc(ntot * nthread, sizeof(uint64_t));
    float *b2 = malloc((nbins + 1) * sizeof(float));
    for (int b = 0; b <= nbins; b++) b2[b] = bins[b] * bins[b];
    const float half = boxsize * 0.5f;
    const float dpi = npibins > 0 ? pimax / (float)npibins : 1.0f;
    const float inv_dmu = nmubins > 0 ? (float)nmubins / mu_max : 1.0f;
    const int rxy = ncxy >= 3 ? 1 : 0, rz = ncz >= 3 ? 1 : 0;
#pragma omp parallel for num_threads(nthread) schedule(dynamic, 16)
    for (int64_t c1 = 0; c1 < ncell; c1++) {
#ifdef _OPENMP
        uint64_t *h = loc + (size_t)omp_get_thread_num() * ntot;
#else
        uint64_t *h = loc;
#endif
        const int64_t i0 = start[0][c1], i1 = start[0][c1 + 1];
        if (i0 == i1) continue;
        const int cz = (int)(c1 % ncz), cy = (int)((c1 / ncz) % ncxy), cx = (int)(c1 / ((int64_t)ncz * ncxy));
        for (int ox = -rxy; ox <= rxy; ox++)
            for (int oy = -rxy; oy <= rxy; oy++)
                for (int oz = -rz; oz <= rz; oz++) {
                    const int nx = (cx + ox + ncxy) % ncxy, ny = (cy + oy + ncxy) % ncxy, nz = (cz + oz + ncz) % ncz;
                    const int64_t c2 = ((int64_t)nx * ncxy + ny) * ncz + nz;
                    const int64_t j0 = start[T][c2], j1 = start[T][c2 + 1];
                    for (int64_t i = i0; i < i1; i++) {
                        const float xi = sx[0][i], yi = sy[0][i], zi = sz[0][i];
                        for (int64_t j = j0; j < j1; j++) {
                            if (autocorr && i == j) continue;
                            float dx = xi - sx[T][j], dy = yi - sy[T][j], dz = zi - sz[T][j];
                            if (dx > half) dx -= boxsize;
                            else if (dx < -half) dx += boxsize;
                            if (dy > half) dy -= boxsize;
                            else if (dy < -half) dy += boxsize;
                            if (dz > half) dz -= boxsize;
                            else if (dz < -half) dz += boxsize;
                            float r2;
                            int sub = 0;
                            if (mode == 1) {
                                float adz = fabsf(dz);
                                if (adz >= pimax) continue;
                                r2 = dx * dx + dy * dy;
                                sub = (int)(adz / dpi);
                                if (sub >= npibins) continue;
                            } else {
                                r2 = dx * dx + dy * dy + dz * dz;
                            }
                            if (r2 < b2[0] || r2 >= b2[nbins]) continue;
                            int b = 0;
                            while (r2 >= b2[b + 1]) b++;
                            if (mode == 2) {
                                float sq = sqrtf(r2);
                                float mu = sq > 0 ? fabsf(dz) / sq : 0.0f;
                                if (mu >= mu_max) continue;
                                sub = (int)(mu * inv_dmu);
                                if (sub >= nmubins) continue;
                            }
                            h[(size_t)b * nsub + sub]++;
                        }
                    }
                }
    }
    for (size_t b = 0; b < ntot; b++) {
        uint64_t s = 0;
        for (int t = 0; t < nthread; t++) s += loc[(size_t)t * ntot + b];
        npairs[b] = s;
    }
    free(loc);
    free(b2);
    for (int s = 0; s < nset; s++) free(start[s]), free(sx[s]), free(sy[s]), free(sz[s]);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* reseed stream (hod/abacus_hod.py:775-839)                                  */
/* ------------------------------------------------------------------------- */
/* The reference redraws hrandoms / hveldev / prandoms from parallel_numpy_rng.MTGenerator(PCG64(seed)) (:778-823), a
 * third-party generator that is not in the tree and not installed (parallel_numpy_rng >= 0.2.0, pyproject.toml:33): the
 * REFERENCE's stream cannot be restated.  What is restated here is the stream the build draws instead - same
 * distributions, dtypes and scalings as the reference (:780 float32 uniforms; :799-801 two-sided exponential of
 * `want_expvel`; :803-818 float32 standard normals; :826-833 `r2 * hsigma3d / sqrt(3)`) - so that "device == CPU bit for
 * bit" is a checked property (tests/test_reseed_gpu.py):
 *   Philox4x32-10 (Salmon, Moraes, Dror & Shaw, SC'11; Random123), pinned by its published known-answer vectors
 *   (tests/test_oracle_reseed.py); halo with global index g: a = Philox(ctr (g lo, g hi, 0, 0), key (seed lo, seed hi)),
 *   b = Philox(ctr (g lo, g hi, 1, 0)); hrandoms = u(a0); normals by Box-Muller, (r0, r1) from (a1, a2), r2 from
 *   (b0, b1); exponential deviates from a1, a2, a3; particle 4 q + d = word d of Philox(ctr (q lo, q hi, 2, 0)).
 *   u(w) = (w >> 8) * 2^-24 in float32.  log / sin / cos are fixed float64 polynomial evaluations (+ - * / sqrt only,
 *   compiled without FMA contraction), so the result does not depend on a math library. */
static void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1,
                       n3 = (uint32_t)p0;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u; /* golden ratio */
        k1 += 0xBB67AE85u; /* sqrt(3) - 1 */
    }
    out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}
void oracle_philox4x32_10(const uint32_t *ctr, const uint32_t *key, uint32_t *out) { philox4x32_10(ctr, key, out); }

static float rs_u01(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-08f; }

/* log of a positive finite double: x = m 2^e, m in [sqrt(1/2), sqrt 2); log m = 2 atanh((m-1)/(m+1)), series to s^15 */
static double rs_log(double x) {
    uint64_t bits;
    memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m;
    memcpy(&m, &bits, 8);
    if (m > 1.4142135623730951) {
        m *= 0.5;
        e += 1;
    }
    const double s = (m - 1.0) / (m + 1.0), z = s * s;
    double p = 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    return 2.0 * s * p + (double)e * 0.6931471805599453;
}
/* sin, cos of 2 pi t for t in [0, 1): quadrant + Taylor series on [0, pi/2) */
static void rs_sincos2pi(double t, double *sn, double *cs) {
    const double t4 = 4.0 * t;
    const int q = (int)t4;
    const double a = (t4 - (double)q) * 1.5707963267948966, z = a * a;
    double ps = -1.0 / 121645100408832000.0;
    ps = ps * z + 1.0 / 355687428096000.0;
    ps = ps * z - 1.0 / 1307674368000.0;
    ps = ps * z + 1.0 / 6227020800.0;
    ps = ps * z - 1.0 / 39916800.0;
    ps = ps * z + 1.0 / 362880.0;
    ps = ps * z - 1.0 / 5040.0;
    ps = ps * z + 1.0 / 120.0;
    ps = ps * z - 1.0 / 6.0;
    ps = ps * z + 1.0;
    const double s0 = a * ps;
    double pc = 1.0 / 6402373705728000.0;
    pc = pc * z - 1.0 / 20922789888000.0;
    pc = pc * z + 1.0 / 87178291200.0;
    pc = pc * z - 1.0 / 479001600.0;
    pc = pc * z + 1.0 / 3628800.0;
    pc = pc * z - 1.0 / 40320.0;
    pc = pc * z + 1.0 / 720.0;
    pc = pc * z - 1.0 / 24.0;
    pc = pc * z + 0.5;
    const double c0 = 1.0 - z * pc;
    switch (q & 3) {
        case 0: *sn = s0, *cs = c0; break;
        case 1: *sn = c0, *cs = -s0; break;
        case 2: *sn = -s0, *cs = -c0; break;
        default: *sn = -c0, *cs = s0; break;
    }
}
double oracle_rs_log(double x) { return rs_log(x); }
void oracle_rs_sincos2pi(double t, double *sn, double *cs) { rs_sincos2pi(t, sn, cs); }

static float rs_laplace(float rt) { /* (:799-801) */
    return rt >= 0.5f ? (float)(-rs_log(2.0 * (1.0 - (double)rt))) : (float)rs_log(2.0 * (double)rt);
}
static float rs_max(float a, float b) { return a > b ? a : b; }

/* halos [index0, index0 + n) of the global catalogue: hrandoms (n) float32 values as float64 (staging keeps float64
 * columns), hveldev (n, 3) = float32 deviate * hsigma3d / sqrt(3) in float64 (:826-833) */
void oracle_reseed_halos(int64_t n, int64_t index0, uint64_t seed, const double *sigma3d, int expvel, double *hrandoms,
                         double *hveldev) {
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        const uint64_t g = (uint64_t)(index0 + i);
        const uint32_t ca[4] = {(uint32_t)g, (uint32_t)(g >> 32), 0u, 0u}, cb[4] = {(uint32_t)g, (uint32_t)(g >> 32), 1u, 0u};
        uint32_t a[4], b[4];
        philox4x32_10(ca, key, a);
        philox4x32_10(cb, key, b);
        hrandoms[i] = (double)rs_u01(a[0]);
        float r[3];
        if (expvel) {
            for (int d = 0; d < 3; d++) r[d] = rs_laplace(rs_max(rs_u01(a[1 + d]), 1e-30f));
        } else {
            const double m0 = sqrt(-2.0 * rs_log(1.0 - (double)rs_u01(a[1]))), m1 = sqrt(-2.0 * rs_log(1.0 - (double)rs_u01(b[0])));
            double s0, c0, s1, c1;
            rs_sincos2pi((double)rs_u01(a[2]), &s0, &c0);
            rs_sincos2pi((double)rs_u01(b[1]), &s1, &c1);
            r[0] = (float)(m0 * c0), r[1] = (float)(m0 * s0), r[2] = (float)(m1 * c1);
        }
        const double sg = sigma3d ? sigma3d[i] : 0.0;
        for (int d = 0; d < 3; d++) hveldev[3 * i + d] = (double)r[d] * sg / 1.7320508075688772;
    }
}
/* particles [index0, index0 + n): prandoms */
void oracle_reseed_particles(int64_t n, int64_t index0, uint64_t seed, double *prandoms) {
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        const uint64_t g = (uint64_t)(index0 + i), q = g >> 2;
        const uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), 2u, 0u};
        uint32_t a[4];
        philox4x32_10(c, key, a);
        prandoms[i] = (double)rs_u01(a[g & 3]);
    }
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
