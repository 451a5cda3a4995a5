// AbacusHOD population on MI355X (gfx950): replaces gen_cent / gen_sats / fast_concatenate of the reference
// (abacusnbody/hod/GRAND_HOD.py:139-414, 825-1262, 1265-1299), called as in gen_gals (:1477-1589).
//
// Data layout in HBM: the staged arrays keep the reference's layout (float64 SoA scalars, (N,3) C-order
// pos/vel, int64 ids) and stay resident across populate calls (the MCMC use case of staging()).
//
// plus what the kernels actually read: a packed 16-bit filter key per object, packed records of whole 128-B memory lines
// for the sparse phases, and for LRG-only runs a (mass bin, q code)-sorted key index.
//
// Kernels (no MFMA - there is no contraction on this path); a populate is filter | deal -> exact -> emit:
//   hod_filter_key  streams the keys (2 B per object): `code > threshold[bin]` proves keep = 0 for the bulk of the objects
//        (an envelope table bounds the marker chain over the bin's masses and the catalogue's environment ranges); the
//        rest is queued per 2048-object tile.  hod_filter / hod_filter32 are the comparator and fallback filters
//        (float64 columns of caller-owned catalogues, float32 shadow columns).
//   hod_deal        LRG alone, from the second populate on: the candidates are prefixes of the sorted key index, found on
//        the host without reading a key; the kernel hands their indices to the tile queues.
//   hod_exact       one workgroup per superblock (16 tiles for LRG alone, 8 for mixes with ELG / QSO): a float32 interval
//        classifier settles a candidate from one packed record line; the reference's float64 chain (erfc, log10, pow) runs
//        only where the random lies inside a marker's band.  Writes the int8 keep bytes (for LRG alone it first un-keeps
//        what the previous populate kept - the filter no longer zeroes the masks) and - through per-tracer LDS bitmaps and
//        a popcount scan - the superblock's kept list in index order plus its three counts.
//   hod_emit        one workgroup per superblock: sums the counters of the superblocks in front of it (a few hundred
//        L2-resident ints) for its output offset - satellites start at Ncent, so centrals||satellites land
//        concatenated (no fast_concatenate pass) - and gathers / writes the kept rows in input order (stable
//        compaction = the reference's order for any Nthread).
// What bounds them is in DESIGN.md section 4: the filter the key stream, the sparse phases the 128-B lines they gather.
// All FP64 arithmetic that reaches an OUTPUT (velocity bias, RSD) is + - * / sqrt in the reference's order,
// compiled with -ffp-contract=off, so outputs are bit-identical to the CPU; the keep decision compares
// randoms against erfc/log10/pow-based markers whose last-ulp differences (ocml vs libm) only matter for a
// random within ~1e-16 of its marker.
#include <cmath>
#include <cstring>
#include <cstddef>
#include <vector>

#include "../../include/abacus_hip.h"
#include "common.hpp"
#include "hod_classify.hpp"

namespace abacus {
int exclusive_scan_u32(unsigned int *counters, int64_t n, int64_t *out, DevBuf &scratch, int zero_counters);
int sort_pairs_u16(const unsigned short *keys_in, unsigned short *keys_out, const unsigned int *val_in, unsigned int *val_out,
                   int64_t n, DevBuf &tmp);   // staging.hip
}
using namespace abacus;

namespace {

constexpr int TILE = 2048;    // objects per workgroup
constexpr int BLOCK = 256;    // threads per workgroup (4 waves)
constexpr int PER_THREAD = TILE / BLOCK;
// Superblock = the tiles one hod_exact / hod_emit workgroup owns.  Chosen per populate (template parameter SBT): 16 tiles
// for LRG alone (a hundred candidates per superblock: fewer, fatter workgroups - 51.8 us per step at 1e7 + 1e7 against 55.2
// with 8 tiles and 55.7 with 32), 8 for mixes with ELG / QSO (thousands per superblock: 308 us against 380 with 16 and 316
// with 4).  An in-superblock index fits uint16.
constexpr int SB_TILES_SPARSE = 16, SB_TILES_DENSE = 8;
constexpr int SB_TILES_MIN = 8, SB_TILES_MAX = 16;

// ---- occupation functions (hod/GRAND_HOD.py:23-136) ------------------------------------------------------
__device__ __forceinline__ double n_cen_LRG(double M_h, double logM_cut, double sigma) {
    return 0.5 * erfc((logM_cut - log10(M_h)) / (1.41421356 * sigma));
}
// x**alpha; x**1.0 is x exactly (also what libm and ocml return), and alpha = 1 is the usual HOD choice
__device__ __forceinline__ double powa(double x, double alpha) { return alpha == 1.0 ? x : pow(x, alpha); }
__device__ __forceinline__ double n_sat_LRG_modified(double M_h, double logM_cut, double M_cut, double M_1,
                                                     double sigma, double alpha, double kappa) {
    if (M_h - kappa * M_cut < 0) return 0;
    return powa((M_h - kappa * M_cut) / M_1, alpha) * 0.5 * erfc((logM_cut - log10(M_h)) / (1.41421356 * sigma));
}
__device__ __forceinline__ double N_sat_generic(double M_h, double M_cut, double kappa, double M_1, double alpha,
                                                double A_s) {
    if (M_h - kappa * M_cut < 0) return 0;
    return A_s * powa((M_h - kappa * M_cut) / M_1, alpha);
}
__device__ __forceinline__ double N_cen_ELG_v1(double M_h, double p_max, double Q, double logM_cut, double sigma,
                                               double gamma) {
    double logM_h = log10(M_h);
    double d = logM_h - logM_cut;
    double phi = 0.3989422804014327 / sigma * exp(-(d * d) / 2 / (sigma * sigma));
    double x = gamma * (logM_h - logM_cut) / sigma;
    double Phi = 0.5 * (1 + erf(x / 1.4142135623730951));
    return 2.0 * (p_max - 1.0 / Q) * phi * Phi / 1;
}
__device__ __forceinline__ double N_cen_QSO(double M_h, double logM_cut, double sigma) {
    return 0.5 * (1 + erf((log10(M_h) - logM_cut) / 1.41421356 / sigma));
}
__device__ __forceinline__ double wrap_box(double x, double L) {
    double L2 = L / 2;
    if (x >= L2) return x - L;
    if (x < -L2) return x + L;
    return x;
}

__device__ __forceinline__ int8_t pick(double r, double m1, double m2, double m3) {
    if (r <= m1) return 1;
    if (r <= m2) return 2;
    if (r <= m3) return 3;
    return 0;
}

// marker chain of gen_cent pass 1 (hod/GRAND_HOD.py:213-252)
__device__ __forceinline__ int8_t cent_decide(const abacus_hod_params &p, double mass, double multis, double randoms,
                                              double deltac, double fenv, double shear) {
    double LRG_marker = 0;
    if (p.want_LRG) {
        double lc = p.L_logM_cut + p.L_Acent * deltac + p.L_Bcent * fenv;
        LRG_marker += n_cen_LRG(mass, lc, p.L_sigma) * p.L_ic * multis;
    }
    double ELG_marker = LRG_marker;
    if (p.want_ELG) {
        double lc = p.E_logM_cut + p.E_Acent * deltac + p.E_Bcent * fenv + p.E_Ccent * shear;
        ELG_marker += N_cen_ELG_v1(mass, p.E_p_max, p.E_Q, lc, p.E_sigma, p.E_gamma) * p.E_ic * multis;
    }
    double QSO_marker = ELG_marker;
    if (p.want_QSO) {
        double lc = p.Q_logM_cut + p.Q_Acent * deltac + p.Q_Bcent * fenv;
        QSO_marker += N_cen_QSO(mass, lc, p.Q_sigma) * p.Q_ic * multis;
    }
    return pick(randoms, LRG_marker, ELG_marker, QSO_marker);
}

using abacus_cls::SatPre;   // particle-independent 10**x values (hod_classify.hpp)

// marker chain of gen_sats pass 1 (hod/GRAND_HOD.py:957-1088)
__device__ __forceinline__ int8_t sat_decide(const abacus_hod_params &p, const SatPre &pre, double hmass,
                                             double weights, double randoms, double d, double f, double sh, double r,
                                             double rv, double rp, double rr, int8_t keep_cent) {
    double LRG_marker = 0;
    if (p.want_LRG) {
        double lc = p.L_logM_cut + p.L_Acent * d + p.L_Bcent * f;
        double M1 = pre.L_const ? pre.L_M1 : pow(10.0, p.L_logM1 + p.L_Asat * d + p.L_Bsat * f);
        double Mcut = pre.L_const ? pre.L_Mcut : pow(10.0, lc);
        double base = n_sat_LRG_modified(hmass, lc, Mcut, M1, p.L_sigma, p.L_alpha, p.L_kappa) * weights * p.L_ic;
        double exp_sat = base;
        if (p.enable_ranks) {
            double dec = 1 + p.L_s * r + p.L_s_v * rv + p.L_s_p * rp + p.L_s_r * rr;
            exp_sat = base * dec;
        }
        LRG_marker += exp_sat;
    }
    double ELG_marker = LRG_marker;
    if (p.want_ELG) {
        double lc = p.E_logM_cut + p.E_Acent * d + p.E_Bcent * f + p.E_Ccent * sh;
        double alpha = p.E_alpha, M1;
        if (keep_cent == 1) {  // ELG conformity (:1006-1035); these branches carry no Csat term
            M1 = pre.E_const ? pre.E_M1_EL : pow(10.0, p.E_logM1_EL + p.E_Asat * d + p.E_Bsat * f);
            alpha = p.E_alpha_EL;
        } else if (keep_cent == 2) {
            M1 = pre.E_const ? pre.E_M1_EE : pow(10.0, p.E_logM1_EE + p.E_Asat * d + p.E_Bsat * f);
            alpha = p.E_alpha_EE;
        } else {
            M1 = pre.E_const ? pre.E_M1 : pow(10.0, p.E_logM1 + p.E_Asat * d + p.E_Bsat * f + p.E_Csat * sh);
        }
        double Mcut = pre.E_const ? pre.E_Mcut : pow(10.0, lc);
        double base = N_sat_generic(hmass, Mcut, p.E_kappa, M1, alpha, p.E_A_s) * weights * p.E_ic;
        if (p.enable_ranks) {
            double dec = 1 + p.E_s * r + p.E_s_v * rv + p.E_s_p * rp + p.E_s_r * rr;
            base = base * dec;
        }
        ELG_marker += base;
    }
    double QSO_marker = ELG_marker;
    if (p.want_QSO) {
        double lc = p.Q_logM_cut + p.Q_Acent * d + p.Q_Bcent * f;
        double M1 = pre.Q_const ? pre.Q_M1 : pow(10.0, p.Q_logM1 + p.Q_Asat * d + p.Q_Bsat * f);
        double Mcut = pre.Q_const ? pre.Q_Mcut : pow(10.0, lc);
        double base = N_sat_generic(hmass, Mcut, p.Q_kappa, M1, p.Q_alpha, 1.0) * weights * p.Q_ic;
        double exp_sat = base;
        if (p.enable_ranks) {
            double dec = 1 + p.Q_s * r + p.Q_s_v * rv + p.Q_s_p * rp + p.Q_s_r * rr;
            exp_sat = base * dec;
        }
        QSO_marker += exp_sat;
    }
    return pick(randoms, LRG_marker, ELG_marker, QSO_marker);
}

// 16-byte coalesced pair load with tail / null handling
__device__ __forceinline__ void load2(const double *a, int64_t i, int64_t n, double fill, double &v0, double &v1) {
    if (a == nullptr) {
        v0 = v1 = fill;
    } else if (i + 1 < n) {
        double2 t = *reinterpret_cast<const double2 *>(a + i);
        v0 = t.x;
        v1 = t.y;
    } else {
        v0 = i < n ? a[i] : fill;
        v1 = fill;
    }
}
__device__ __forceinline__ double load1(const double *a, int64_t i, double fill) { return a ? a[i] : fill; }

// ---- float32 rejection filter ----------------------------------------------------------------------------
// Most objects host no galaxy: their random exceeds the last marker by orders of magnitude.  A float32 UPPER BOUND
// U >= (largest marker of the chain) is ~10x cheaper than the FP64 erfc/log10/pow chain; `random > U` proves keep = 0
// without changing any decision.  Everything else (including every accepted object) goes through the exact FP64
// path.  The bounds carry explicit slack for float32 rounding (1e-6 per operand), for the ocml float functions
// (1e-4 relative) and for the propagated argument error of erfc; `filter` is switched off by the host for parameter
// sets the bounds do not cover (negative ic / A_s / kappa-free cases, non-finite values).
struct Filt {
    int cent_ok, sat_ok;   // sat_ok: the arithmetic satellite bound applies (particle-independent M1 / M_cut)
    int sat_basic, pad_;   // finite parameters, alpha >= 0, A_s >= 0, ic >= 0 (what the envelope table needs)
    float L_lc, L_Ac, L_Bc, L_inv_s, L_ic;                      // centrals + LRG satellites share lc, sigma
    float E_lc, E_Ac, E_Bc, E_Cc, E_c_phi, E_half_inv_s2, E_ic;  // c_phi = max(2(pmax-1/Q),0) * 0.39894/sigma
    float E_gs;                                                  // gamma / sigma / sqrt(2)
    float Q_lc, Q_Ac, Q_Bc, Q_inv_s, Q_ic;
    // satellites (only used when the tracer's 10**x values are particle independent, SatPre::*_const)
    float L_invM1, L_alpha, L_s[4];
    float E_invM1[3], E_alpha[3], E_As, E_s[4];                  // [default, cent is LRG (EL), cent is ELG (EE)]
    float Q_invM1, Q_alpha, Q_s[4];
    double L_kMcut, E_kMcut, Q_kMcut;                            // kappa * M_cut
};

// upper bound of 0.5*erfc(t_true): t = num*inv_s with |num_true - num| <= dnum; erfc is decreasing.
// Abramowitz & Stegun 7.1.13, x >= 0:  1 / (x + sqrt(x^2 + 2)) < exp(x^2) int_x^inf exp(-t^2) dt <= 1 / (x + sqrt(x^2 + 4/pi)),
// so 0.5 erfc(x) <= exp(-x^2) / (sqrt(pi) (x + sqrt(x^2 + 4/pi))) (exact at 0, 5 % high at 1, tighter further out) and
// 0.5 erfc(-x) = 1 - 0.5 erfc(x) <= 1 - exp(-x^2) / (sqrt(pi) (x + sqrt(x^2 + 2))).  Hardware exp / sqrt / rcp (1 ulp) with slack.
__device__ __forceinline__ float half_erfc_ub(float num, float dnum, float inv_s) {
    const float t = num * inv_s;
    const float tl = t - (dnum * fabsf(inv_s) * 1.001f + fabsf(t) * 2e-6f);
    if (tl >= 9.0f) return 4e-36f;    // 0.5*exp(-81) / 32 = 1e-37
    if (!(tl > -9.0f)) return 1.0f;   // also NaN
    const float x = fabsf(tl), x2 = x * x;
    if (tl >= 0.f)
        return __expf(-x2 * 0.9999f) * (0.5641896f * 1.0003f) * __builtin_amdgcn_rcpf(x + __builtin_amdgcn_sqrtf(x2 + 1.2732395f));
    return 1.0f - __expf(-x2 * 1.0001f) * (0.5641895f * 0.9997f) * __builtin_amdgcn_rcpf(x + __builtin_amdgcn_sqrtf(x2 + 2.0f));
}

// T = double: the staged float64 columns; T = float: their float32 shadows (mass rounded up, randoms rounded down,
// the rest to nearest - the slack terms below already cover one float32 rounding of every operand)
template <class T>
__device__ __forceinline__ bool cent_reject(const abacus_hod_params &p, const Filt &F, T mass, T multis, T randoms,
                                            T deltac, T fenv, T shear) {
    if (!(multis >= (T)0)) return false;
    const float lM = __log10f((float)mass);   // v_log_f32 (1 ulp); its error is inside `dn` below
    const float mu = (float)multis * 1.00001f;
    const float d = (float)deltac, f = (float)fenv, sh = (float)shear;
    float U = 0.f;
    if (p.want_LRG) {
        const float a1 = F.L_Ac * d, a2 = F.L_Bc * f;
        const float lc = F.L_lc + a1 + a2;
        const float dn = 1e-6f * (fabsf(F.L_lc) + fabsf(a1) + fabsf(a2) + fabsf(lM) + 4.f);
        U += half_erfc_ub(lc - lM, dn, F.L_inv_s) * F.L_ic * mu;
    }
    if (p.want_ELG) {
        const float a1 = F.E_Ac * d, a2 = F.E_Bc * f, a3 = F.E_Cc * sh;
        const float lc = F.E_lc + a1 + a2 + a3;
        const float dn = 1e-6f * (fabsf(F.E_lc) + fabsf(a1) + fabsf(a2) + fabsf(a3) + fabsf(lM) + 4.f);
        const float dl = fmaxf(fabsf(lM - lc) - dn, 0.f);            // |logM - logM_cut| is at least this
        const float phi = F.E_c_phi * __expf(-(dl * dl) * F.E_half_inv_s2 * 0.9999f) * 1.0002f + 1e-37f;
        const float Phi = half_erfc_ub(lc - lM, dn, F.E_gs);         // 0.5 (1 + erf(y)) = 0.5 erfc(-y), y = gs (logM - logM_cut)
        U += phi * Phi * F.E_ic * mu;
    }
    if (p.want_QSO) {  // 0.5*(1+erf(u)) = 0.5*erfc(-u)
        const float a1 = F.Q_Ac * d, a2 = F.Q_Bc * f;
        const float lc = F.Q_lc + a1 + a2;
        const float dn = 1e-6f * (fabsf(F.Q_lc) + fabsf(a1) + fabsf(a2) + fabsf(lM) + 4.f);
        U += half_erfc_ub(lc - lM, dn, F.Q_inv_s) * F.Q_ic * mu;
    }
    return randoms > (T)(U * 1.001f);
}

__device__ __forceinline__ float pow_ub(float x, float alpha) {   // upper bound of x_true**alpha, x within 1e-6
    if (alpha == 1.0f) return x * 1.00001f;
    return powf(x, alpha) * (1.0002f + fabsf(alpha) * 4e-6f);
}
// upper bound of max(1 + s r + s_v r_v + s_p r_p + s_r r_r, 0): the value itself (float32 coefficients) plus its rounding
// error; a negative factor makes the tracer's term negative, and the chain's largest marker is bounded by the positive terms
template <class T>
__device__ __forceinline__ float dec_ub(const float s[4], T r, T rv, T rp, T rr) {
    const float t0 = s[0] * (float)r, t1 = s[1] * (float)rv, t2 = s[2] * (float)rp, t3 = s[3] * (float)rr;
    const float v = (((1.f + t0) + t1) + t2) + t3;
    return fmaxf(v + 2e-6f * (1.f + fabsf(t0) + fabsf(t1) + fabsf(t2) + fabsf(t3)), 0.f) * 1.00001f;
}
template <class T>
__device__ __forceinline__ bool sat_reject(const abacus_hod_params &p, const Filt &F, T hmass, T weights, T randoms, T r,
                                           T rv, T rp, T rr, int8_t keep_cent) {
    if (!(weights >= (T)0)) return false;
    const float w = (float)weights * 1.00001f;
    float U = 0.f;
    if (p.want_LRG) {
        const double xd = (double)hmass - F.L_kMcut;   // the exact FP64 test of n_sat_LRG_modified (:28); a shadow mass is >= the true one
        if (!(xd < 0)) {
            const float lM = __log10f((float)hmass);
            const float dn = 1e-6f * (fabsf(F.L_lc) + fabsf(lM) + 4.f);
            float term = pow_ub((float)xd * F.L_invM1, F.L_alpha) * half_erfc_ub(F.L_lc - lM, dn, F.L_inv_s) * w * F.L_ic;
            if (p.enable_ranks) term *= dec_ub(F.L_s, r, rv, rp, rr);
            U += term;
        }
    }
    if (p.want_ELG) {
        const double xd = (double)hmass - F.E_kMcut;
        if (!(xd < 0)) {
            float pw;
            if (keep_cent < 0) {   // host decision not known yet (filter ahead of the central exact pass): largest variant
                pw = fmaxf(pow_ub((float)xd * F.E_invM1[0], F.E_alpha[0]),
                           fmaxf(pow_ub((float)xd * F.E_invM1[1], F.E_alpha[1]), pow_ub((float)xd * F.E_invM1[2], F.E_alpha[2])));
            } else {
                const int v = keep_cent == 1 ? 1 : (keep_cent == 2 ? 2 : 0);
                pw = pow_ub((float)xd * F.E_invM1[v], F.E_alpha[v]);
            }
            float term = F.E_As * pw * w * F.E_ic;
            if (p.enable_ranks) term *= dec_ub(F.E_s, r, rv, rp, rr);
            U += term;
        }
    }
    if (p.want_QSO) {
        const double xd = (double)hmass - F.Q_kMcut;
        if (!(xd < 0)) {
            float term = pow_ub((float)xd * F.Q_invM1, F.Q_alpha) * w * F.Q_ic;
            if (p.enable_ranks) term *= dec_ub(F.Q_s, r, rv, rp, rr);
            U += term;
        }
    }
    return randoms > (T)(U * 1.001f);
}

// Decide = two launches per object kind:
//   hod_filter_*  streams the per-object scalars (16-B coalesced loads) through the float32 filter, zeroes the int8
//                 mask, and writes the tile-local indices of the survivors to the tile's slice of a queue;
//   hod_exact_*   evaluates the exact FP64 marker chain for the queued objects with all lanes busy (typically 1-2 %
//                 of the objects), writes their mask bytes and the tile / superblock counters.
// Splitting keeps the streaming kernel light (no FP64 transcendental code, few registers, 8 waves per SIMD).
constexpr int FBLOCK = 256;

// All device pointers of the staged catalogue + work arrays, passed by value to the fused kernels
// Packed per-object records (owned catalogues), laid out by 128-B LINE - the granularity the memory system fetches at:
// scripts/ubench/gather.hip reads 64 B or 128 B of a sparse, ascending subset of 128-B records at the same 4.7e10 records/s
// (= 6 TB/s of whole lines) whatever the density, and 3.1e10 /s when a 128-B read straddles two lines.  The kernels of a
// dense tracer mix are bound by exactly that - lines touched per candidate / galaxy:
//   halo      one line: what hod_exact reads for a candidate (mass, multiplicity, the random, environment) AND what hod_emit
//             reads for a galaxy (id, position, velocity, the halo's velocity deviate);
//   particle  line 0 = everything hod_exact reads (host mass, weight, random, environment, the four ranks, and the host
//             halo's index `pinds` for the conformity look-up), line 1 = everything hod_emit reads.
// (With 64-B "lines" a particle was 192 B: 1.5 memory lines per candidate, 1.5 per galaxy, plus one for pinds[i]; five
// arrays in five pages before that.)  The random and hveldev change with a reseed / update: those fields are rewritten
// then (hod_refresh_recs), the rest is built once.  Absent optional columns are stored as the value the exact chain
// substitutes for them (0, ranks 1, pinds -1).
struct __attribute__((aligned(128))) HaloRec {
    double mass, multis, rnd, deltac, fenv, shear;
    long long id;
    double pos0;
    double pos1, pos2, vel[3], vdev[3];
};
struct __attribute__((aligned(128))) PartRec {
    double mass, weights, rnd, deltac, fenv, shear, rank0, rank1, rank2, rank3;   // line 0: decide
    long long pinds;
    double pad0[5];
    long long id;                                                                 // line 1: emit
    double mass2, pos[3], vel0, vel1, vel2, hvel[3];
    double pad1[5];
};
static_assert(sizeof(HaloRec) == 128 && sizeof(PartRec) == 256 && offsetof(PartRec, id) == 128, "record sizes");

struct RecSrc {
    const double *hpos, *hvel, *hvdev, *hmass, *hmultis, *hrandoms, *hdeltac, *hfenv, *hshear;
    const int64_t *hid;
    const double *ppos, *pvel, *phvel, *phmass, *pweights, *prandoms, *pdeltac, *pfenv, *pshear, *pranks[4];
    const int64_t *phid, *pinds;
};
// RAND_ONLY: rewrite just the fields a reseed / update changes (the random; the halo's velocity deviate)
template <bool RAND_ONLY>
__global__ void hod_build_recs(int64_t nh, int64_t np, RecSrc c, HaloRec *__restrict__ hrec, PartRec *__restrict__ prec) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = t0; i < nh; i += stride) {
        if (RAND_ONLY) {
            hrec[i].rnd = c.hrandoms[i];
            for (int d = 0; d < 3; d++) hrec[i].vdev[d] = c.hvdev[3 * i + d];
            continue;
        }
        HaloRec r;
        r.mass = c.hmass[i], r.multis = c.hmultis[i], r.rnd = c.hrandoms[i], r.id = c.hid[i];
        r.deltac = c.hdeltac ? c.hdeltac[i] : 0.0, r.fenv = c.hfenv ? c.hfenv[i] : 0.0, r.shear = c.hshear ? c.hshear[i] : 0.0;
        r.pos0 = c.hpos[3 * i], r.pos1 = c.hpos[3 * i + 1], r.pos2 = c.hpos[3 * i + 2];
        for (int d = 0; d < 3; d++) r.vel[d] = c.hvel[3 * i + d], r.vdev[d] = c.hvdev[3 * i + d];
        hrec[i] = r;
    }
    for (int64_t i = t0; i < np; i += stride) {
        if (RAND_ONLY) {
            prec[i].rnd = c.prandoms[i];
            continue;
        }
        PartRec r;
        r.mass = r.mass2 = c.phmass[i], r.weights = c.pweights[i], r.rnd = c.prandoms[i], r.id = c.phid[i];
        r.deltac = c.pdeltac ? c.pdeltac[i] : 0.0, r.fenv = c.pfenv ? c.pfenv[i] : 0.0, r.shear = c.pshear ? c.pshear[i] : 0.0;
        r.rank0 = c.pranks[0] ? c.pranks[0][i] : 1.0, r.rank1 = c.pranks[1] ? c.pranks[1][i] : 1.0;
        r.rank2 = c.pranks[2] ? c.pranks[2][i] : 1.0, r.rank3 = c.pranks[3] ? c.pranks[3][i] : 1.0;
        for (int d = 0; d < 3; d++) r.pos[d] = c.ppos[3 * i + d], r.hvel[d] = c.phvel[3 * i + d];
        r.vel0 = c.pvel[3 * i], r.vel1 = c.pvel[3 * i + 1], r.vel2 = c.pvel[3 * i + 2];
        r.pinds = c.pinds ? c.pinds[i] : -1;
        for (int d = 0; d < 5; d++) r.pad0[d] = r.pad1[d] = 0.0;
        prec[i] = r;
    }
}

struct HodPtrs {
    const HaloRec *hrec = nullptr;   // packed per-object records (owned catalogues), or nullptr: gather from the columns
    const PartRec *prec = nullptr;
    int64_t nh, np;
    int ntile_c, ntile_s, nsb_c, nsb_s;
    const double *hmass, *hmultis, *hrandoms, *hdeltac, *hfenv, *hshear;
    const double *phmass, *pweights, *prandoms, *pdeltac, *pfenv, *pshear, *pranks, *pranksv, *pranksp, *pranksr;
    const int64_t *pinds;
    int8_t *keep_c, *keep_s;
    int *q_count;                       // [ntile_c + ntile_s]
    unsigned short *queue_c, *queue_s;  // one TILE-sized slice per tile
    unsigned short *kept_c, *kept_s;    // one SB_OBJ-sized slice per superblock: kept objects, tracer-major, index order
    int *sb_counts;                     // [nsb_c + nsb_s][4]
};

// Superblock S of a kind owns the tiles [S ntile / nsb, (S + 1) ntile / nsb): nsb >= ntile / SBT balanced runs of at most SBT
// tiles (the host picks nsb, set_superblocks: a whole number of workgroups per CU for the dense mixes)
__device__ __forceinline__ int sb_first_tile(int S, int ntile, int nsb) { return (int)((int64_t)S * ntile / nsb); }

// One launch filters central tiles (global tile id < ntile_c) and satellite tiles alike; `first_tile` lets the host
// split it in two when the satellite filter needs the exact central decisions (ELG conformity reads keep_cent[pinds]).
// `need_env`: some wanted tracer has a non-zero Acent/Bcent (Ccent for the shear); otherwise deltac / fenv / shear
// enter the reference's formula as `0 * x` and are not read (a non-finite x makes the exact marker NaN = never kept,
// which a filter that does not reject is consistent with).
__global__ __launch_bounds__(FBLOCK) void hod_filter(HodPtrs a, int first_tile, int want_LRG, int want_ELG,
                                                     int want_QSO, int enable_ranks, int need_env, int need_shear,
                                                     Filt F) {
    __shared__ int nq;
    __shared__ unsigned short q[TILE];
    const int tid = threadIdx.x;
    if (tid == 0) nq = 0;
    __syncthreads();
    const int g = (int)blockIdx.x + first_tile;
    const bool sat = g >= a.ntile_c;
    const int T = sat ? g - a.ntile_c : g;
    const int64_t n = sat ? a.np : a.nh;
    const int64_t tile0 = (int64_t)T * TILE;
    int8_t *keep = sat ? a.keep_s : a.keep_c;
    abacus_hod_params pw;   // only the want_* / enable_ranks flags are read by *_reject
    pw.want_LRG = want_LRG, pw.want_ELG = want_ELG, pw.want_QSO = want_QSO, pw.enable_ranks = enable_ranks;
    if (!sat) {
#pragma unroll 2
        for (int k = 0; k < PER_THREAD / 2; k++) {
            const int loc = k * (2 * FBLOCK) + 2 * tid;
            const int64_t i = tile0 + loc;
            if (i < n) {
                bool need0 = true, need1 = i + 1 < n;
                if (F.cent_ok) {
                    double m0, m1, mu0, mu1, r0, r1, d0 = 0, d1 = 0, f0 = 0, f1 = 0, s0 = 0, s1 = 0;
                    load2(a.hmass, i, n, 1.0, m0, m1);
                    load2(a.hmultis, i, n, 0.0, mu0, mu1);
                    load2(a.hrandoms, i, n, 2.0, r0, r1);
                    if (need_env) {
                        load2(a.hdeltac, i, n, 0.0, d0, d1);
                        load2(a.hfenv, i, n, 0.0, f0, f1);
                    }
                    if (need_shear) load2(a.hshear, i, n, 0.0, s0, s1);
                    need0 = !cent_reject(pw, F, m0, mu0, r0, d0, f0, s0);
                    need1 = need1 && !cent_reject(pw, F, m1, mu1, r1, d1, f1, s1);
                }
                if (need0) q[atomicAdd(&nq, 1)] = (unsigned short)loc;
                if (need1) q[atomicAdd(&nq, 1)] = (unsigned short)(loc + 1);
            }
        }
    } else {
        const bool need_conf = want_ELG && a.pinds != nullptr;
#pragma unroll 2
        for (int k = 0; k < PER_THREAD / 2; k++) {
            const int loc = k * (2 * FBLOCK) + 2 * tid;
            const int64_t i = tile0 + loc;
            if (i < n) {
                bool need0 = true, need1 = i + 1 < n;
                if (F.sat_ok) {
                    double m0, m1, w0, w1, r0, r1;
                    double a0 = 1, a1 = 1, b0 = 1, b1 = 1, c0 = 1, c1 = 1, e0 = 1, e1 = 1;
                    int8_t kc0 = 0, kc1 = 0;
                    load2(a.phmass, i, n, 1.0, m0, m1);
                    load2(a.pweights, i, n, 0.0, w0, w1);
                    load2(a.prandoms, i, n, 2.0, r0, r1);
                    if (enable_ranks) {
                        load2(a.pranks, i, n, 1.0, a0, a1);
                        load2(a.pranksv, i, n, 1.0, b0, b1);
                        load2(a.pranksp, i, n, 1.0, c0, c1);
                        load2(a.pranksr, i, n, 1.0, e0, e1);
                    }
                    if (need_conf) {
                        kc0 = a.keep_c[a.pinds[i]];
                        if (i + 1 < n) kc1 = a.keep_c[a.pinds[i + 1]];
                    }
                    need0 = !sat_reject(pw, F, m0, w0, r0, a0, b0, c0, e0, kc0);
                    need1 = need1 && !sat_reject(pw, F, m1, w1, r1, a1, b1, c1, e1, kc1);
                }
                if (need0) q[atomicAdd(&nq, 1)] = (unsigned short)loc;
                if (need1) q[atomicAdd(&nq, 1)] = (unsigned short)(loc + 1);
            }
        }
    }
    {   // zero this tile's mask: 8 consecutive bytes per thread
        const int64_t o = tile0 + (int64_t)tid * 8;
        if (o + 8 <= n) *reinterpret_cast<unsigned long long *>(keep + o) = 0ull;
        else
            for (int q8 = 0; q8 < 8; q8++)
                if (o + q8 < n) keep[o + q8] = 0;
    }
    __syncthreads();
    // the tile's survivors go to the tile's own slice of the queue: no global atomics anywhere
    const int cnt = nq;
    if (tid == 0) a.q_count[g] = cnt;
    unsigned short *queue = sat ? a.queue_s : a.queue_c;
    for (int j = tid; j < cnt; j += FBLOCK) queue[tile0 + j] = q[j];
}

// ---- float32 shadows of the columns the filter streams ----------------------------------------------------------------
// The filter only ever uses float32 casts of its inputs (every bound carries slack for that rounding), so for a catalogue
// the library owns it streams float32 SHADOW columns built once at staging: 12 B per object instead of 24 (20 instead of
// 40 with assembly bias).  Rounding modes keep every bound an upper bound: masses round UP (the exact test
// `M - kappa M_cut < 0` of the satellite occupations then never fires for a particle the exact path keeps, and n(M) only
// grows with M), randoms round DOWN (`r32 > U` implies `r > U`), the other columns to nearest (covered by the slack).
// Shadows are padded to a multiple of four with values that always reject, so the filter loads whole float4s.
template <int MODE>   // 0: nearest, 1: up, 2: down
__global__ void hod_shadow(const double *__restrict__ src, float *__restrict__ dst, int64_t n, int64_t npad, float fill) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npad; i += (int64_t)gridDim.x * blockDim.x) {
        float f = fill;
        if (i < n) {
            const double x = src[i];
            f = (float)x;
            if (MODE == 1 && (double)f < x) f = nextafterf(f, INFINITY);
            if (MODE == 2 && (double)f > x) f = nextafterf(f, -INFINITY);
        }
        dst[i] = f;
    }
}

struct FiltCols {
    const float *hmass, *hmultis, *hrandoms, *hdeltac, *hfenv, *hshear;
    const float *phmass, *pweights, *prandoms, *pranks, *pranksv, *pranksp, *pranksr;
};

__device__ __forceinline__ void load4f(const float *a, int64_t i, float fill, float (&v)[4]) {
    if (a == nullptr) {
        v[0] = v[1] = v[2] = v[3] = fill;
    } else {
        const float4 t = *reinterpret_cast<const float4 *>(a + i);
        v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
    }
}

// ---- two-stage filter: a table bound in the streaming loop, the arithmetic bound only for its survivors ---------------
// Where the summed occupation of the wanted tracers is non-decreasing in mass (centrals: LRG / QSO erfc forms without
// assembly bias; satellites: every power-law form the filter covers), the bound of a whole mass BIN is the bound at its
// upper edge.  Bins follow the float32 representation (exponent and top three mantissa bits: 8 per octave, 2^33..2^54),
// so the bin of a shadow mass (rounded up) is a shift and a subtraction; the table holds host-evaluated float64
// occupations at the upper edges, rounded up, with a floor of 1e-30 (no bound ever underflows to "reject at r > 0").
// Stage 1 (per object: one table look-up, two multiplies, one compare - no log10 / exp / pow) fills an LDS queue with
// the few per cent that survive; stage 2 evaluates the arithmetic bound of hod_filter for those, all lanes busy, and
// fills the tile's queue.  Conformity (keep_cent[pinds]) is only looked up in stage 2.
constexpr int CH_SHIFT = 20, CH_BASE = (127 + 33) << 3, CH_NLEV = 21 * 8;
struct Cheap {
    int c_ok, s_ok;
    float dec_max;            // >= 1 + sum_q |s_q| |rank_q| for every wanted tracer and every staged rank value
    float pad_;
    float Bc[CH_NLEV], Bs[CH_NLEV];
};

__device__ __forceinline__ float cheap_bound(const float *tab, float mass) {
    const unsigned int lev = (__float_as_uint(mass) >> CH_SHIFT);   // negative / NaN masses land above the table
    const int j = (int)lev - CH_BASE;
    return j < 0 ? tab[0] : (j < CH_NLEV ? tab[j] : INFINITY);
}

// hod_filter on the shadow columns: four consecutive objects per thread and step (one float4 per column).  One
// instantiation per (object kind, one- or two-stage): a launch carries only the code it runs - with all four paths in one
// kernel body the same work took 68 instead of 59 us (instruction fetch).  `first_tile` counts from the first tile of
// the kind.
// KIND: 0 = central tiles, 1 = satellite tiles, 2 = both in one launch (first_tile is then a global tile id)
template <int KIND, bool TWO_STAGE>
__global__ __launch_bounds__(FBLOCK) void hod_filter32(HodPtrs a, FiltCols c, int first_tile, int want_LRG, int want_ELG,
                                                       int want_QSO, int enable_ranks, int need_env, int need_shear,
                                                       Filt F, Cheap ch) {
    __shared__ int nq, nq1;
    __shared__ unsigned short q[TILE], q1[TWO_STAGE ? TILE : 1];
    __shared__ float tab[TWO_STAGE ? CH_NLEV : 1];
    const int tid = threadIdx.x;
    if (tid == 0) nq = 0, nq1 = 0;
    const bool SAT = KIND == 2 ? (int)blockIdx.x + first_tile >= a.ntile_c : KIND == 1;
    if (TWO_STAGE && tid < CH_NLEV) tab[tid] = SAT ? ch.Bs[tid] : ch.Bc[tid];
    __syncthreads();
    const int T = KIND == 2 ? (int)blockIdx.x + first_tile - (SAT ? a.ntile_c : 0) : (int)blockIdx.x + first_tile;
    const int g = SAT ? T + a.ntile_c : T;          // global tile id (index of q_count)
    const int64_t n = SAT ? a.np : a.nh;
    const int64_t tile0 = (int64_t)T * TILE;
    int8_t *keep = SAT ? a.keep_s : a.keep_c;
    abacus_hod_params pw;
    pw.want_LRG = want_LRG, pw.want_ELG = want_ELG, pw.want_QSO = want_QSO, pw.enable_ranks = enable_ranks;
    const bool need_conf = SAT && want_ELG && a.pinds != nullptr;
    if constexpr (TWO_STAGE) {
        // ---- stage 1: envelope table bound - three float4 loads per four objects whatever the HOD weights ----
        const float dec = ch.dec_max;
#pragma unroll 1
        for (int k = 0; k < PER_THREAD / 4; k++) {
            const int loc = k * (4 * FBLOCK) + 4 * tid;
            const int64_t i = tile0 + loc;
            if (i >= n) continue;
            float m[4], w[4], r[4];
            load4f(SAT ? c.phmass : c.hmass, i, 1.f, m);
            load4f(SAT ? c.pweights : c.hmultis, i, 0.f, w);
            load4f(SAT ? c.prandoms : c.hrandoms, i, 2.f, r);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (i + u >= n) continue;
                bool pass = true;                                  // negative / NaN multiplicities are never rejected
                if (w[u] >= 0.f) {
                    float U = cheap_bound(tab, m[u]) * (w[u] * 1.00001f);
                    if (SAT) U *= dec;
                    pass = !(r[u] > U * 1.001f);
                }
                if (pass) q1[atomicAdd(&nq1, 1)] = (unsigned short)(loc + u);
            }
        }
        __syncthreads();
        // ---- stage 2: the arithmetic bound with the object's own environment / ranks, for the survivors ----
        const int n1 = nq1;
        for (int e = tid; e < n1; e += FBLOCK) {
            const int loc = q1[e];
            const int64_t i = tile0 + loc;
            bool rej = false;
            if (!SAT) {
                const float d = need_env && c.hdeltac ? c.hdeltac[i] : 0.f, f = need_env && c.hfenv ? c.hfenv[i] : 0.f,
                            sh = need_shear && c.hshear ? c.hshear[i] : 0.f;
                rej = cent_reject<float>(pw, F, c.hmass[i], c.hmultis[i], c.hrandoms[i], d, f, sh);
            } else if (F.sat_ok) {
                // conformity: the host's exact decision may not exist yet (one filter launch for both kinds) -> -1
                const float r0 = enable_ranks ? c.pranks[i] : 1.f, r1 = enable_ranks ? c.pranksv[i] : 1.f,
                            r2 = enable_ranks ? c.pranksp[i] : 1.f, r3 = enable_ranks ? c.pranksr[i] : 1.f;
                rej = sat_reject<float>(pw, F, c.phmass[i], c.pweights[i], c.prandoms[i], r0, r1, r2, r3,
                                        need_conf ? (int8_t)-1 : (int8_t)0);
            }   // satellites with assembly bias: the envelope's survivors go straight to the exact chain
            if (!rej) q[atomicAdd(&nq, 1)] = (unsigned short)loc;
        }
    } else {
#pragma unroll 1
        for (int k = 0; k < PER_THREAD / 4; k++) {
            const int loc = k * (4 * FBLOCK) + 4 * tid;
            const int64_t i = tile0 + loc;
            if (i >= n) continue;
            bool need[4];
#pragma unroll
            for (int u = 0; u < 4; u++) need[u] = i + u < n;
            if (!SAT && F.cent_ok) {
                float m[4], mu[4], r[4], d[4], f[4], sh[4];
                load4f(c.hmass, i, 1.f, m);
                load4f(c.hmultis, i, 0.f, mu);
                load4f(c.hrandoms, i, 2.f, r);
                load4f(need_env ? c.hdeltac : nullptr, i, 0.f, d);
                load4f(need_env ? c.hfenv : nullptr, i, 0.f, f);
                load4f(need_shear ? c.hshear : nullptr, i, 0.f, sh);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    need[u] = need[u] && !cent_reject<float>(pw, F, m[u], mu[u], r[u], d[u], f[u], sh[u]);
            } else if (SAT && F.sat_ok) {
                float m[4], w[4], r[4], r0[4], r1[4], r2[4], r3[4];
                load4f(c.phmass, i, 1.f, m);
                load4f(c.pweights, i, 0.f, w);
                load4f(c.prandoms, i, 2.f, r);
                load4f(enable_ranks ? c.pranks : nullptr, i, 1.f, r0);
                load4f(enable_ranks ? c.pranksv : nullptr, i, 1.f, r1);
                load4f(enable_ranks ? c.pranksp : nullptr, i, 1.f, r2);
                load4f(enable_ranks ? c.pranksr : nullptr, i, 1.f, r3);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    int8_t kc = 0;
                    if (need_conf && need[u]) kc = a.keep_c[a.pinds[i + u]];
                    need[u] = need[u] && !sat_reject<float>(pw, F, m[u], w[u], r[u], r0[u], r1[u], r2[u], r3[u], kc);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (need[u]) q[atomicAdd(&nq, 1)] = (unsigned short)(loc + u);
        }
    }
    {   // zero this tile's mask: 8 consecutive bytes per thread
        const int64_t o = tile0 + (int64_t)tid * 8;
        if (o + 8 <= n) *reinterpret_cast<unsigned long long *>(keep + o) = 0ull;
        else
            for (int q8 = 0; q8 < 8; q8++)
                if (o + q8 < n) keep[o + q8] = 0;
    }
    __syncthreads();
    const int cnt = nq;
    if (tid == 0) a.q_count[g] = cnt;
    unsigned short *queue = SAT ? a.queue_s : a.queue_c;
    for (int j = tid; j < cnt; j += FBLOCK) queue[tile0 + j] = q[j];
}

// ---- packed filter keys: 2 bytes per object -------------------------------------------------------------------------------
// The filter compares `random > B[bin(mass)] * weight * dec`.  Everything on the object's side of that inequality is fixed
// once the catalogue and its randoms are staged, so it is folded into ONE 16-bit key per object:
//   low 7 bits   the mass bin: the float32-representation bins of cheap_bound (8 per octave), window 2^36 ... 2^51.9 - bin 0
//                also takes every smaller mass (its bound is the largest of the levels it covers), bin 127 everything above
//                and whatever is not a mass (never rejected);
//   high 9 bits  a CODE of q <= random / weight: exponent and three mantissa bits of the float32 lower bound (its bits >> 20,
//                offset so that code 0 is 2^-44 and below), i.e. q rounded down to eight steps per octave over 2^-44 ... 2^20.
//                `code > code(B[bin] * dec)` implies `random > B[bin] * weight * dec`.  Code 0 is never rejected and also
//                stands for "the division says nothing" (weight < 0 or NaN, random <= 0 or NaN): a bin whose bound lies below
//                2^-44 sends the objects with a random / weight below that (float32 randoms: the zeros) to hod_exact.  Above
//                2^20 codes and bounds saturate at 511: never rejected either; +inf (weight = 0 and a positive random: the
//                marker is 0 * n = 0, never kept) too.
// The coarse q costs candidates - objects whose q lies within a step (6 - 12 %) above the bound - and halves what the filter
// streams: 2 B per object (40 MB at 1e7 + 1e7; 4-B keys with a 16-bit mantissa: 80 MB; float32 shadow columns: 240 MB), one
// LDS table look-up and one integer compare per object, eight tiles per workgroup so that eight 16-B loads per thread are in
// flight.  (Four steps per octave over 2^-100 ... 2^28: +10 % candidates; sixteen over 2^-24 ... 2^8: the satellites of
// massive hosts, whose bound exceeds 2^8, all pass - 2.3e6 instead of 1.3e6 particles at LRG + ELG + QSO.)  Keys are
// rebuilt (one pass) when the randoms change (reseed / update); the parameters never enter them.
constexpr int K16_LEV0 = (36 - 33) * 8;            // CH level that is key bin 0: upper edge 2^36 * 9/8
constexpr int K16_QSHIFT = 20, K16_QOFF = (127 - 44) << 3;   // (float bits >> 20) of 2^-44
__host__ __device__ __forceinline__ int k16_code(float v) {   // v >= 0 or NaN / inf: monotone, saturating
    unsigned int u;
    memcpy(&u, &v, 4);
    const int c = (int)((u & 0x7fffffffu) >> K16_QSHIFT) - K16_QOFF;
    return c < 0 ? 0 : (c > 511 ? 511 : c);
}
__global__ __launch_bounds__(256) void hod_build_keys(const double *__restrict__ mass, const double *__restrict__ wgt,
                                                      const double *__restrict__ rnd, int64_t n, int64_t npad,
                                                      unsigned short *__restrict__ keys) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npad; i += (int64_t)gridDim.x * 256) {
        unsigned int key = 127u;   // padding: never rejected - masked by i < n anyway
        if (i < n) {
            const double m = mass[i], w = wgt[i], r = rnd[i];
            float mf = (float)m;
            if ((double)mf < m) mf = nextafterf(mf, INFINITY);     // rounded up, like the shadow masses
            const int j = (int)(__float_as_uint(mf) >> CH_SHIFT) - CH_BASE - K16_LEV0;   // negative / NaN masses: sign bit -> above
            const unsigned int bin = (__float_as_uint(mf) >> 31) || !(mf == mf) ? 127u : (j < 0 ? 0u : (j < 127 ? (unsigned int)j : 127u));
            float q = 0.f;
            if (w > 0.0 && r > 0.0) {
                const double qd = r / w * (1.0 - 1e-6);
                q = (float)qd;
                if ((double)q > qd) q = nextafterf(q, 0.f);
                if (!(q == q)) q = 0.f;
            } else if (w == 0.0 && r > 0.0) {
                q = INFINITY;
            }
            key = ((unsigned int)k16_code(q) << 7) | bin;
        }
        keys[i] = (unsigned short)key;
    }
}

#ifndef ABACUS_KEY_TILES
#define ABACUS_KEY_TILES 8
#endif
constexpr int KEY_TILES = ABACUS_KEY_TILES;   // tiles per workgroup of the key filter

// KIND: 0 = central tile groups, 1 = satellite tile groups, 2 = both (central groups first)
// The table bound is the WHOLE filter here: its survivors go straight to the tiles' queue slices.  An arithmetic bound with
// the object's own environment behind it (a second stage gathering the survivors' records) was measured and dropped: a
// gather costs a 128-B line whatever it reads (scripts/ubench/gather.hip: 21 us per million lines), hod_exact pays the
// same line plus ~35 us of classifier per million, so the second stage only wins where it removes more than 40 % of the
// table's survivors - LRG alone: 2 % (56.3 vs 60.2 us per step without it); LRG + ELG + QSO with assembly bias: 42 % of the
// halos, 53 % of the particles (354 vs 371 us).
struct KeyTab {   // per key bin: the largest q code that is NOT rejected (host-built from the envelope table, make_keytab)
    unsigned short c[128], s[128];
};
template <int KIND>
__global__ __launch_bounds__(FBLOCK) void hod_filter_key(HodPtrs a, const unsigned short *__restrict__ hkeys,
                                                         const unsigned short *__restrict__ pkeys, int ngroup_c, KeyTab kt, int nozero) {
    __shared__ int nq[KEY_TILES];
    __shared__ int tc[128];
    const int tid = threadIdx.x;
    const bool SAT = KIND == 2 ? (int)blockIdx.x >= ngroup_c : KIND == 1;
    const int G = KIND == 2 && SAT ? (int)blockIdx.x - ngroup_c : (int)blockIdx.x;
    if (tid < 128) tc[tid] = SAT ? kt.s[tid] : kt.c[tid];
    if (tid < KEY_TILES) nq[tid] = 0;
    const int ntile = SAT ? a.ntile_s : a.ntile_c;
    const int64_t n = SAT ? a.np : a.nh;
    const unsigned short *keys = SAT ? pkeys : hkeys;
    int8_t *keep = SAT ? a.keep_s : a.keep_c;
    unsigned short *queue = SAT ? a.queue_s : a.queue_c;
    // all key loads of the workgroup's tiles first: 16 B = eight keys per thread and tile (the key array is padded to whole tiles)
    static_assert(TILE == 8 * FBLOCK, "one 16-B load per thread and tile");
    uint4 k[KEY_TILES];
    const int t_first = G * KEY_TILES;
    const int64_t base0 = (int64_t)t_first * TILE;
#pragma unroll
    for (int t = 0; t < KEY_TILES; t++)
        k[t] = (t_first + t < ntile) ? *reinterpret_cast<const uint4 *>(keys + base0 + (int64_t)t * TILE + 8 * tid) : make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();            // the table and the zeroed counters
    // one LDS table look-up and one compare per object
#pragma unroll
    for (int t = 0; t < KEY_TILES; t++) {
        if (t_first + t >= ntile) break;      // uniform
        const int loc = 8 * tid;
        const unsigned int kw[4] = {k[t].x, k[t].y, k[t].z, k[t].w};
        const int lim = (int)min((int64_t)8, n - (base0 + (int64_t)t * TILE + loc));   // objects of this load inside the catalogue
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (u >= lim) continue;
            const unsigned int key = (kw[u >> 1] >> ((u & 1) * 16)) & 0xffffu;
            if ((int)(key >> 7) <= tc[key & 127u]) queue[base0 + (int64_t)t * TILE + atomicAdd(&nq[t], 1)] = (unsigned short)(loc + u);
        }
    }
    if (!nozero) {   // zero the tiles' masks: 8 consecutive bytes per thread and tile (unless hod_exact un-keeps, see there)
#pragma unroll
        for (int t = 0; t < KEY_TILES; t++) {
            const int64_t o = base0 + (int64_t)t * TILE + (int64_t)tid * 8;
            if (t_first + t >= ntile) break;
            if (o + 8 <= n) *reinterpret_cast<unsigned long long *>(keep + o) = 0ull;
            else
                for (int q8 = 0; q8 < 8; q8++)
                    if (o + q8 < n) keep[o + q8] = 0;
        }
    }
    __syncthreads();
    if (tid < KEY_TILES && t_first + tid < ntile) a.q_count[(SAT ? a.ntile_c : 0) + t_first + tid] = nq[tid];   // global tile id
}

// ---- mass-sorted key index (sparse mixes) ---------------------------------------------------------------------------
// For LRG alone 95 % of the halos sit in mass bins whose bound lies below any positive random: streaming their keys only
// to reject them is most of the filter's 17 us.  Once per catalogue (from its second populate with unchanged keys on) the
// objects are sorted by (mass bin, q code); the candidates of a populate are then a PREFIX of every bin's segment - the
// objects with code <= the bin's threshold code - whose length the host reads off a cumulative table without touching the
// device.  hod_deal hands those indices to the tiles' queues (one global atomic per candidate: a hundred thousand at 1e7 +
// 1e7); the candidate set is the key filter's, so everything behind it is unchanged.
__global__ __launch_bounds__(256) void hod_index_keys(const unsigned short *__restrict__ keys, int64_t n,
                                                      unsigned short *__restrict__ sk, unsigned int *__restrict__ idx) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const unsigned int k = keys[i];
        sk[i] = (unsigned short)(((k & 127u) << 9) | (k >> 7));   // bin-major
        idx[i] = (unsigned int)i;
    }
}
// last[v] = 1 + position of the last object with sort key v (0: none), from the sorted keys
__global__ __launch_bounds__(256) void hod_index_last(const unsigned short *__restrict__ sk, int64_t n, unsigned int *__restrict__ last) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        if (i == n - 1 || sk[i] != sk[i + 1]) last[sk[i]] = (unsigned int)(i + 1);
}
struct DealTab {   // candidate segments of the sorted index: centrals first; pre = exclusive prefix of the lengths
    int nseg, nseg_c;
    unsigned int start[256], pre[257];
};
__global__ __launch_bounds__(256) void hod_deal(HodPtrs a, const unsigned int *__restrict__ idx_h, const unsigned int *__restrict__ idx_p,
                                                DealTab tab) {
    __shared__ unsigned int s_pre[257], s_start[256];
    const int tid = threadIdx.x;
    for (int q = tid; q <= tab.nseg; q += 256) s_pre[q] = tab.pre[q];
    for (int q = tid; q < tab.nseg; q += 256) s_start[q] = tab.start[q];
    __syncthreads();
    const unsigned int j = blockIdx.x * 256u + tid;
    if (j >= s_pre[tab.nseg]) return;
    int lo = 0, hi = tab.nseg - 1;   // largest segment with pre <= j
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (s_pre[mid] <= j) lo = mid;
        else hi = mid - 1;
    }
    const bool sat = lo >= tab.nseg_c;
    const unsigned int i = (sat ? idx_p : idx_h)[s_start[lo] + (j - s_pre[lo])];
    const unsigned int tile = i >> 11, loc = i & (TILE - 1);
    static_assert(TILE == 2048, "tile index arithmetic");
    const int slot = atomicAdd(&a.q_count[(sat ? a.ntile_c : 0) + (int)tile], 1);
    if (slot < TILE)   // always, when the counters started at zero (the host guarantees it; never write past a tile's slice)
        (sat ? a.queue_s : a.queue_c)[(int64_t)tile * TILE + slot] = (unsigned short)loc;
}

// The reference's float64 chains as OUT-OF-LINE functions reading the parameters through a pointer (the workgroup's LDS
// copy).  hod_exact settles its candidates with the float32 interval classifier (hod_classify.hpp: a decision is taken
// from float32 enclosures of the markers unless the random lies inside a band) and calls these only for the undecided few
// 1e-4; inlined, their ~190 registers (hoisted float64 constants of three tracers) set the occupancy of the whole kernel.
__device__ __noinline__ int cent_decide_cold(const abacus_hod_params *p, double mass, double multis, double rnd, double dc,
                                             double fe, double sh) {
    return cent_decide(*p, mass, multis, rnd, dc, fe, sh);
}
__device__ __noinline__ int sat_decide_cold(const abacus_hod_params *p, const SatPre *pre, double mass, double w, double rnd,
                                            double dc, double fe, double sh, double r0, double r1, double r2, double r3,
                                            int kc) {
    return sat_decide(*p, *pre, mass, w, rnd, dc, fe, sh, r0, r1, r2, r3, (int8_t)kc);
}

// Exact kernel: one workgroup per superblock (16 tiles = 32768 objects).  The tiles' queue lengths are prefix-summed
// in LDS so the few hundred survivors of the superblock are processed as one dense list.  Kept objects set a bit in
// one of three LDS bitmaps (one per tracer); a popcount scan of the bitmaps then yields every kept object's rank in
// index order, and the workgroup writes the superblock's kept list (uint16 in-superblock indices, tracer-major,
// ascending) plus its three counts.  hod_emit needs nothing else: no mask re-read, no per-tile counters.
template <int SBT>
struct ExactLds {
    int pre[SBT + 1];
    unsigned int bm[3][SBT * TILE / 32];
    unsigned long long wave_tot[8];
};

template <int SBT>
__device__ __forceinline__ int exact_find_tile(const ExactLds<SBT> &L, int j) {   // largest q with pre[q] <= j
    int lo = 0, hi = SBT - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (L.pre[mid] <= j) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// One launch for the central superblocks (global superblock id < nsb_c) and the satellite ones; `first_sb` splits
// it when the satellites depend on the exact central decisions (ELG conformity).
// PIPE (dense tracer mixes: thousands of candidates per superblock, a dozen per thread): the record of the thread's next
// candidate and the queue entry of the one after it are requested before the current candidate is classified, so the
// memory round trips (queue entry -> record line -> conformity byte) run under the classifier's ~1000 instructions instead
// of in front of them.  It costs ~40 registers, which a sparse mix (LRG alone: less than one candidate per thread, twice
// as many resident workgroups) would pay for nothing.
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void aload16(v4u &dst, const void *p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void aload_u16(unsigned int &dst, const void *p) {
    asm volatile("global_load_ushort %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void aload_i8(int &dst, const void *p) {
    asm volatile("global_load_sbyte %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int K>
__device__ __forceinline__ void await_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory");
}
// tie registers to the wait above them: their uses cannot be scheduled before it
__device__ __forceinline__ void touch4(v4u &a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void touch1(unsigned int &a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void touch1i(int &a) { asm volatile("" : "+v"(a)); }

struct ExactCand {
    double mass, w, rnd, dc, fe, sh, r0, r1, r2, r3;
    long long pinds;
    int q, loc;
};

template <int XB, bool PIPE, int SBT>
__device__ __forceinline__ void hod_exact_body(const HodPtrs &a, int first_sb, const abacus_hod_params &p, const SatPre &pre,
                                               const abacus_cls::ClsConst &cc, int use_cls, int clear_prev) {
    constexpr int SB_TILES = SBT, SB_OBJ = SBT * TILE, SB_WORDS = SB_OBJ / 32;
    constexpr int WORDS_PER_THREAD = SB_WORDS / XB;
    static_assert(SB_WORDS % XB == 0 && XB <= 512 && SB_OBJ <= 65536, "bitmap words must divide over the workgroup");
    __shared__ ExactLds<SBT> L;
    __shared__ abacus_hod_params s_p;     // read by the out-of-line float64 chains
    __shared__ SatPre s_pre;
    const int tid = threadIdx.x;
    static_assert(sizeof(abacus_hod_params) % 8 == 0 && sizeof(SatPre) % 8 == 0, "copied as 8-byte words");
    if (tid < (int)(sizeof(abacus_hod_params) / 8))
        reinterpret_cast<unsigned long long *>(&s_p)[tid] = reinterpret_cast<const unsigned long long *>(&p)[tid];
    if (tid < (int)(sizeof(SatPre) / 8))
        reinterpret_cast<unsigned long long *>(&s_pre)[tid] = reinterpret_cast<const unsigned long long *>(&pre)[tid];
    const int g = (int)blockIdx.x + first_sb;
    const bool sat = g >= a.nsb_c;
    const int S = sat ? g - a.nsb_c : g;
    const int ntile = sat ? a.ntile_s : a.ntile_c, nsb = sat ? a.nsb_s : a.nsb_c;
    const int tile_first = sb_first_tile(S, ntile, nsb), ntl = sb_first_tile(S + 1, ntile, nsb) - tile_first;   // ntl <= SB_TILES
    const int64_t obj_first = (int64_t)tile_first * TILE;
    const int *q_count = a.q_count + (sat ? a.ntile_c : 0);
    if (tid < SB_TILES) {
        const int t = tile_first + tid;
        L.pre[tid + 1] = tid < ntl ? q_count[t] : 0;
        if ((clear_prev & 2) && tid < ntl) a.q_count[(sat ? a.ntile_c : 0) + t] = 0;   // the index path counts into zeroed counters
    }
    if (tid == 0) L.pre[0] = 0;
#pragma unroll
    for (int w = 0; w < 3 * WORDS_PER_THREAD; w++) (&L.bm[0][0])[w * XB + tid] = 0u;
    __syncthreads();
    if (tid == 0)
        for (int q = 1; q <= SB_TILES; q++) L.pre[q] += L.pre[q - 1];
    __syncthreads();
    const int total = L.pre[SB_TILES];
    if (clear_prev & 1) {
        // Lazy keep masks (sparse mixes): the filter did not zero the 1 B per object - 20 MB of the 100 MB it moves at 1e7 +
        // 1e7 (23.8 -> 21.3 us; 83 -> 67 us at 4e7 + 4e7).  The only non-zero bytes are the objects the PREVIOUS populate
        // kept, and this superblock's share of them is still listed in its kept slice (counts in sb_counts, overwritten at
        // the end of this workgroup): un-keep those, then decide.  The host falls back to the zeroing filter whenever the
        // lists do not describe the masks (first populate, another superblock size, the NFW path in between).
        const int prev = a.sb_counts[(int64_t)g * 4] + a.sb_counts[(int64_t)g * 4 + 1] + a.sb_counts[(int64_t)g * 4 + 2];
        const unsigned short *pk = (sat ? a.kept_s : a.kept_c) + obj_first;
        int8_t *keep = (sat ? a.keep_s : a.keep_c) + obj_first;
        for (int e = tid; e < prev; e += XB) keep[pk[e]] = 0;
        __syncthreads();   // an object kept again is written again below, by whichever thread classifies it
    }
    const bool need_conf = sat && p.want_ELG && a.pinds != nullptr;
    const bool need_ranks = p.enable_ranks != 0;
    const unsigned short *queue = sat ? a.queue_s : a.queue_c;
    // tile and tile-local index of the superblock's j-th candidate
    auto locate = [&](int j, int &q, int &loc) {
        q = exact_find_tile(L, j);
        loc = queue[(int64_t)(tile_first + q) * TILE + (j - L.pre[q])];
    };
    // the candidate's scalars: one record line, or the staged columns (caller-owned catalogues)
    auto fetch = [&](int q, int loc, ExactCand &c) {
        c.q = q, c.loc = loc;
        const int64_t i = (int64_t)(tile_first + q) * TILE + loc;
        c.r0 = c.r1 = c.r2 = c.r3 = 1.0;
        c.pinds = 0;
        if (!sat) {
            if (a.hrec) {
                const HaloRec &r = a.hrec[i];
                c.mass = r.mass, c.w = r.multis, c.rnd = r.rnd, c.dc = r.deltac, c.fe = r.fenv, c.sh = p.want_ELG ? r.shear : 0.0;
            } else {
                c.rnd = a.hrandoms[i];
                c.mass = a.hmass[i], c.w = a.hmultis[i], c.dc = load1(a.hdeltac, i, 0.0), c.fe = load1(a.hfenv, i, 0.0),
                c.sh = p.want_ELG ? load1(a.hshear, i, 0.0) : 0.0;
            }
        } else {
            if (a.prec) {
                const PartRec &r = a.prec[i];   // line 0
                c.mass = r.mass, c.w = r.weights, c.rnd = r.rnd, c.dc = r.deltac, c.fe = r.fenv, c.sh = p.want_ELG ? r.shear : 0.0;
                if (need_ranks) c.r0 = r.rank0, c.r1 = r.rank1, c.r2 = r.rank2, c.r3 = r.rank3;
                if (need_conf) c.pinds = r.pinds;
            } else {
                if (need_conf) c.pinds = a.pinds[i];
                c.rnd = a.prandoms[i];
                c.mass = a.phmass[i], c.w = a.pweights[i], c.dc = load1(a.pdeltac, i, 0.0), c.fe = load1(a.pfenv, i, 0.0),
                c.sh = p.want_ELG ? load1(a.pshear, i, 0.0) : 0.0;
                if (need_ranks) c.r0 = a.pranks[i], c.r1 = a.pranksv[i], c.r2 = a.pranksp[i], c.r3 = a.pranksr[i];
            }
        }
    };
    auto classify = [&](const ExactCand &c, int kc) {   // -1: the random lies inside a marker's band (or the classifier is off)
        if (!use_cls) return -1;
        return sat ? abacus_cls::sat_classify(cc, c.mass, c.w, c.rnd, c.dc, c.fe, c.sh, c.r0, c.r1, c.r2, c.r3, kc)
                   : abacus_cls::cent_classify(cc, c.mass, c.w, c.rnd, c.dc, c.fe, c.sh);
    };
    auto exact_chain = [&](const ExactCand &c, int kc) {   // the reference's float64 chain (out of line)
        return sat ? sat_decide_cold(&s_p, &s_pre, c.mass, c.w, c.rnd, c.dc, c.fe, c.sh, c.r0, c.r1, c.r2, c.r3, kc)
                   : cent_decide_cold(&s_p, c.mass, c.w, c.rnd, c.dc, c.fe, c.sh);
    };
    auto record = [&](const ExactCand &c, int kk) {
        if (kk) {
            (sat ? a.keep_s : a.keep_c)[(int64_t)(tile_first + c.q) * TILE + c.loc] = (int8_t)kk;
            const int ls = c.q * TILE + c.loc;
            atomicOr(&L.bm[kk - 1][ls >> 5], 1u << (ls & 31));
        }
    };
    if constexpr (!PIPE) {
        for (int j = tid; j < total; j += XB) {
            int q, loc;
            locate(j, q, loc);
            ExactCand c;
            fetch(q, loc, c);
            const int kc = need_conf ? a.keep_c[c.pinds] : 0;   // keep_cent[pinds[i]] (GRAND_HOD.py:1562)
            int kk = classify(c, kc);
            if (kk < 0) kk = exact_chain(c, kc);
            record(c, kk);
        }
    } else {
        // Records only (the host launches this variant for owned catalogues).  The loads of the pipeline are inline-asm loads
        // the compiler does not track, waited for by hand: left to the compiler, the address arithmetic of the candidate after
        // next was hoisted above the classification and with it a vmcnt(0) that drained the prefetch before it could overlap.
        // Lanes past their last candidate keep loading the superblock's last candidate (valid addresses, no extra branches).
        const char *rec = sat ? reinterpret_cast<const char *>(a.prec) : reinterpret_cast<const char *>(a.hrec);
        const int64_t rb = sat ? (int64_t)sizeof(PartRec) : (int64_t)sizeof(HaloRec);
        auto queue_addr = [&](int jj, int &q) {
            jj = min(jj, total - 1);
            q = exact_find_tile(L, jj);
            return queue + ((int64_t)(tile_first + q) * TILE + (jj - L.pre[q]));
        };
        auto rec_addr = [&](int q, int loc) { return rec + ((int64_t)(tile_first + q) * TILE + loc) * rb; };
        auto decode = [&](const v4u (&r)[6], int q, int loc, ExactCand &c) {
            auto dbl = [](unsigned int lo, unsigned int hi) { return __hiloint2double((int)hi, (int)lo); };
            c.q = q, c.loc = loc;
            c.mass = dbl(r[0].x, r[0].y), c.w = dbl(r[0].z, r[0].w), c.rnd = dbl(r[1].x, r[1].y), c.dc = dbl(r[1].z, r[1].w);
            c.fe = dbl(r[2].x, r[2].y), c.sh = p.want_ELG ? dbl(r[2].z, r[2].w) : 0.0;
            c.r0 = c.r1 = c.r2 = c.r3 = 1.0;
            c.pinds = 0;
            if (sat) {
                if (need_ranks) c.r0 = dbl(r[3].x, r[3].y), c.r1 = dbl(r[3].z, r[3].w), c.r2 = dbl(r[4].x, r[4].y), c.r3 = dbl(r[4].z, r[4].w);
                c.pinds = (long long)(((unsigned long long)r[5].y << 32) | r[5].x);
            }
        };
        if (total > 0) {   // uniform
            int j = tid;
            int q0, q1, q2;
            unsigned int loc0, loc1, loc2 = 0;
            v4u raw[6];
            aload_u16(loc0, queue_addr(j, q0));
            aload_u16(loc1, queue_addr(j + XB, q1));
            await_vm<0>();
            touch1(loc0), touch1(loc1);
            {
                const char *r0 = rec_addr(q0, (int)loc0);
#pragma unroll
                for (int l = 0; l < 6; l++)
                    if (l < 3 || sat) aload16(raw[l], r0 + 16 * l);
            }
            await_vm<0>();
#pragma unroll
            for (int l = 0; l < 6; l++)
                if (l < 3 || sat) touch4(raw[l]);
            ExactCand cur;
            decode(raw, q0, (int)loc0, cur);
            int kc = 0;
            if (need_conf) {
                aload_i8(kc, a.keep_c + cur.pinds);
                await_vm<0>();
                touch1i(kc);
            }
#pragma unroll 1
            for (; j < total; j += XB) {
                {   // the next candidate's record and the queue entry of the one after it: in flight during decide()
                    const char *r1 = rec_addr(q1, (int)loc1);
#pragma unroll
                    for (int l = 0; l < 6; l++)
                        if (l < 3 || sat) aload16(raw[l], r1 + 16 * l);
                    aload_u16(loc2, queue_addr(j + 2 * XB, q2));
                }
                int kk = classify(cur, kc);
                if (kk < 0) {
                    // a function call saves and restores live registers around it: the prefetch must have landed first
                    await_vm<0>();
#pragma unroll
                    for (int l = 0; l < 6; l++)
                        if (l < 3 || sat) touch4(raw[l]);
                    touch1(loc2);
                    kk = exact_chain(cur, kc);
                }
                record(cur, kk);
                await_vm<0>();
#pragma unroll
                for (int l = 0; l < 6; l++)
                    if (l < 3 || sat) touch4(raw[l]);
                touch1(loc2);
                decode(raw, q1, (int)loc1, cur);
                if (need_conf) {   // keep_cent[pinds]: 10 MB, cache-resident
                    aload_i8(kc, a.keep_c + cur.pinds);
                    await_vm<0>();
                    touch1i(kc);
                }
                q1 = q2, loc1 = loc2;
            }
        }
    }
    __syncthreads();
    // ranks: thread t owns words [4t, 4t+4) of each bitmap; packed 3 x 21-bit exclusive scan over the workgroup
    unsigned int wbits[3][WORDS_PER_THREAD];
    unsigned long long mine = 0;
#pragma unroll
    for (int t = 0; t < 3; t++) {
        unsigned int c = 0;
#pragma unroll
        for (int w = 0; w < WORDS_PER_THREAD; w++) {
            wbits[t][w] = L.bm[t][tid * WORDS_PER_THREAD + w];
            c += __popc(wbits[t][w]);
        }
        mine |= (unsigned long long)c << (21 * t);
    }
    unsigned long long incl = mine;
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
    }
    if (lane == 63) L.wave_tot[wv] = incl;
    __syncthreads();
    unsigned long long before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < XB / 64; w++) {
        if (w < wv) before += L.wave_tot[w];
        all += L.wave_tot[w];
    }
    const unsigned long long excl = before + incl - mine;
    const int T0 = (int)(all & 0x1fffff), T1 = (int)((all >> 21) & 0x1fffff), T2 = (int)((all >> 42) & 0x1fffff);
    unsigned short *kept = (sat ? a.kept_s : a.kept_c) + obj_first;
    const int base[3] = {0, T0, T0 + T1};
#pragma unroll
    for (int t = 0; t < 3; t++) {
        int r = base[t] + (int)((excl >> (21 * t)) & 0x1fffff);
#pragma unroll
        for (int w = 0; w < WORDS_PER_THREAD; w++) {
            unsigned int bits = wbits[t][w];
            while (bits) {
                const int bpos = __ffs((int)bits) - 1;
                bits &= bits - 1;
                kept[r++] = (unsigned short)((tid * WORDS_PER_THREAD + w) * 32 + bpos);
            }
        }
    }
    if (tid < 4) a.sb_counts[(int64_t)g * 4 + tid] = tid == 0 ? T0 : (tid == 1 ? T1 : (tid == 2 ? T2 : 0));
}

// The pipelined form (dense mixes) runs at its ~150 registers, three workgroups per CU (set_superblocks counts on that); the
// plain form (sparse mixes: LRG alone) is held to 128 registers - two or three values spilled - for a fourth wave per SIMD
// under its dependent gathers: 38.4 -> 35.0 us at 4e7 + 4e7, no change at 1e7 + 1e7.  (The pipelined form held to 128 spills
// twenty: 94 us instead of 81.)
template <int XB, bool PIPE, int SBT>
__global__ __launch_bounds__(XB) void hod_exact(HodPtrs a, int first_sb, abacus_hod_params p, SatPre pre,
                                                abacus_cls::ClsConst cc, int use_cls, int clear_prev) {
    static_assert(PIPE, "the plain form is hod_exact_plain");
    hod_exact_body<XB, true, SBT>(a, first_sb, p, pre, cc, use_cls, clear_prev);
}
template <int XB, int SBT>
__global__ __launch_bounds__(XB) __attribute__((amdgpu_waves_per_eu(4))) void hod_exact_plain(HodPtrs a, int first_sb,
                                                                                               abacus_hod_params p, SatPre pre,
                                                                                               abacus_cls::ClsConst cc, int use_cls,
                                                                                               int clear_prev) {
    hod_exact_body<XB, false, SBT>(a, first_sb, p, pre, cc, use_cls, clear_prev);
}

struct OutCols {
    double *c[3][7];  // [tracer][x,y,z,vx,vy,vz,mass]
    int64_t *id[3];
    int64_t cap[3];
};

// galaxy emission shared by centrals (hod/GRAND_HOD.py:298-325) and satellites (:1134-1165)
typedef __attribute__((address_space(1))) double *gdouble_p;   // the output columns are device allocations
typedef __attribute__((address_space(1))) long long *gint64_p;

__device__ __forceinline__ void emit_one(const abacus_hod_params &p, const OutCols &o, int t, int64_t j, double x,
                                         double y, double z, double vx, double vy, double vz, double mass,
                                         int64_t id) {
    if (j >= o.cap[t]) return;  // buffers too small: the host grows them and re-runs the emission
    if (p.rsd && p.has_origin) {
        double nx = x - p.origin[0], ny = y - p.origin[1], nz = z - p.origin[2];
        double inv_norm = 1.0 / sqrt(nx * nx + ny * ny + nz * nz);
        nx *= inv_norm;
        ny *= inv_norm;
        nz *= inv_norm;
        double proj = p.inv_velz2kms * (vx * nx + vy * ny + vz * nz);
        x = x + proj * nx;
        y = y + proj * ny;
        z = z + proj * nz;
    } else if (p.rsd) {
        z = wrap_box(z + vz * p.inv_velz2kms, p.lbox);
    }
    __builtin_nontemporal_store(x, (gdouble_p)o.c[t][0] + j);
    __builtin_nontemporal_store(y, (gdouble_p)o.c[t][1] + j);
    __builtin_nontemporal_store(z, (gdouble_p)o.c[t][2] + j);
    __builtin_nontemporal_store(vx, (gdouble_p)o.c[t][3] + j);
    __builtin_nontemporal_store(vy, (gdouble_p)o.c[t][4] + j);
    __builtin_nontemporal_store(vz, (gdouble_p)o.c[t][5] + j);
    __builtin_nontemporal_store(mass, (gdouble_p)o.c[t][6] + j);
    __builtin_nontemporal_store((long long)id, (gint64_p)o.id[t] + j);
}

// Ordered emission, one workgroup per superblock.  The output offset of a superblock is a workgroup reduction over
// the counts of the superblocks in front of it (satellites start after all centrals of their tracer, so
// centrals||satellites land concatenated and fast_concatenate never runs); the rank inside the superblock is the
// position in the kept list hod_exact wrote.  All lanes gather and emit: the only inputs are the counters, the kept
// list and the kept rows.  Workgroups [0, nsb_c) handle centrals, the rest satellites.

__device__ __forceinline__ int64_t wave_sum(int64_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

struct EmitPtrs {
    const double *hpos, *hvel, *hvdev, *hmass, *ppos, *pvel, *phvel, *phmass;
    const int64_t *hid, *phid;
    const HaloRec *hrec;   // nullptr: gather from the columns
    const PartRec *prec;
};

template <int EBLOCK, int SBT>
__global__ __launch_bounds__(EBLOCK) void hod_emit(int nsb_c, int nsb_s, int ntile_c, int ntile_s,
                                                   const unsigned short *__restrict__ kept_c,
                                                   const unsigned short *__restrict__ kept_s,
                                                   const int *__restrict__ sb_counts, int64_t *__restrict__ totals,
                                                   EmitPtrs in, abacus_hod_params p, OutCols o_arg, int dbg) {
    __shared__ int64_t red[EBLOCK / 64][6];
    // the column pointers and capacities are indexed by the galaxy's tracer: from a copy in LDS (an LDS read per use, counted
    // by lgkmcnt) - indexing the kernel arguments at run time made every galaxy load them from the argument segment with
    // vector memory loads, whose waits sat in front of the column stores
    __shared__ OutCols o;
    static_assert(sizeof(OutCols) % 8 == 0, "copied as 8-byte words");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < (int)(sizeof(OutCols) / 8))
        reinterpret_cast<unsigned long long *>(&o)[tid] = reinterpret_cast<const unsigned long long *>(&o_arg)[tid];
    const int g = blockIdx.x;
    const bool sat = g >= nsb_c;
    const int S = sat ? g - nsb_c : g;
    const int64_t obj_first = (int64_t)sb_first_tile(S, sat ? ntile_s : ntile_c, sat ? nsb_s : nsb_c) * TILE;
    const int *sb_c = sb_counts, *sb_s = sb_counts + (int64_t)nsb_c * 4;
    const int *sb_mine = sat ? sb_s : sb_c;
    // v[0..2]: counts of the superblocks of my kind in front of me; v[3..5]: all central counts (satellite offset,
    // and block 0 reports the totals)
    int64_t v[6] = {0, 0, 0, 0, 0, 0};
    for (int s = tid; s < S; s += EBLOCK)
#pragma unroll
        for (int t = 0; t < 3; t++) v[t] += sb_mine[(int64_t)s * 4 + t];
    if (sat || g == 0)
        for (int s = tid; s < nsb_c; s += EBLOCK)
#pragma unroll
            for (int t = 0; t < 3; t++) v[3 + t] += sb_c[(int64_t)s * 4 + t];
#pragma unroll
    for (int t = 0; t < 6; t++) v[t] = wave_sum(v[t]);
    if (lane == 0)
#pragma unroll
        for (int t = 0; t < 6; t++) red[wv][t] = v[t];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 6; t++) {
        v[t] = 0;
#pragma unroll
        for (int w = 0; w < EBLOCK / 64; w++) v[t] += red[w][t];
    }
    if (g == 0) {   // totals for the host: Ncent[3], Nsat[3]
        int64_t s3[3] = {0, 0, 0};
        for (int s = tid; s < nsb_s; s += EBLOCK)
#pragma unroll
            for (int t = 0; t < 3; t++) s3[t] += sb_s[(int64_t)s * 4 + t];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 3; t++) s3[t] = wave_sum(s3[t]);
        if (lane == 0)
#pragma unroll
            for (int t = 0; t < 3; t++) red[wv][t] = s3[t];
        __syncthreads();
        if (tid < 3) {
            int64_t tot = 0;
            for (int w = 0; w < EBLOCK / 64; w++) tot += red[w][tid];
            totals[3 + tid] = tot;
            totals[tid] = tid == 0 ? v[3] : (tid == 1 ? v[4] : v[5]);
        }
    }
    const int m0 = sb_counts[(int64_t)g * 4], m1 = sb_counts[(int64_t)g * 4 + 1], m2 = sb_counts[(int64_t)g * 4 + 2];
    const int total = m0 + m1 + m2;
    if (total == 0) return;
    const int64_t off0 = v[0] + (sat ? v[3] : 0), off1 = v[1] + (sat ? v[4] : 0), off2 = v[2] + (sat ? v[5] : 0);
    const unsigned short *kept = (sat ? kept_s : kept_c) + obj_first;
    const double a0 = sat ? p.L_alpha_s : p.L_alpha_c, a1 = sat ? p.E_alpha_s : p.E_alpha_c,
                 a2 = sat ? p.Q_alpha_s : p.Q_alpha_c;
    if (in.hrec && in.prec && !(dbg & 3)) {
        // Records: software pipeline with untracked loads (see hod_exact).  Written plainly, an iteration was three dependent
        // waits - the kept index, the record line, and (vmcnt counts in order) the eight column stores of the iteration
        // before, which the wait for the kept index drained.  Here the stores of galaxy k are issued, then the record of
        // k + 1 and the kept index of k + 2 are requested, and one wait covers all of them: 103 -> 95 us at LRG + ELG + QSO
        // (1.8e6 galaxies).  What remains is bandwidth: gather alone 58 us, stores alone 38 us - 234 MB of record lines in
        // and 117 MB of columns out share the memory system at ~4 TB/s.
        const char *rec = sat ? reinterpret_cast<const char *>(in.prec) + offsetof(PartRec, id) : reinterpret_cast<const char *>(in.hrec);
        const int64_t rb = sat ? (int64_t)sizeof(PartRec) : (int64_t)sizeof(HaloRec);
        // 16-B pieces of the line: halo [mass, multis] [id, pos0] [pos1, pos2] [vel0, vel1] [vel2, vdev0] [vdev1, vdev2];
        // particle (line 1) [id, mass] [pos0, pos1] [pos2, vel0] [vel1, vel2] [hvel0, hvel1] [hvel2, -]
        const int o1 = sat ? 16 : 48, o2 = sat ? 32 : 64, o3 = sat ? 48 : 80, o4 = sat ? 64 : 96, o5 = sat ? 80 : 112;
        auto issue = [&](unsigned int k, v4u (&r)[6]) {
            const char *q = rec + (obj_first + k) * rb;
            aload16(r[0], q), aload16(r[1], q + o1), aload16(r[2], q + o2), aload16(r[3], q + o3), aload16(r[4], q + o4),
                aload16(r[5], q + o5);
        };
        auto dbl = [](unsigned int lo, unsigned int hi) { return __hiloint2double((int)hi, (int)lo); };
        int e = tid;
        if (e >= total) return;
        unsigned int k0, k1, k2 = 0;
        v4u r[6];
        aload_u16(k0, kept + e);
        aload_u16(k1, kept + min(e + EBLOCK, total - 1));
        await_vm<0>();
        touch1(k0), touch1(k1);
        issue(k0, r);
        await_vm<0>();
#pragma unroll 1
        for (;;) {
#pragma unroll
            for (int l = 0; l < 6; l++) touch4(r[l]);
            const int t = e < m0 ? 0 : (e < m0 + m1 ? 1 : 2);
            const int64_t j = t == 0 ? off0 + e : (t == 1 ? off1 + (e - m0) : off2 + (e - m0 - m1));
            const double al = t == 0 ? a0 : (t == 1 ? a1 : a2);
            double x, y, z, vx, vy, vz, m;
            int64_t id;
            if (!sat) {
                m = dbl(r[0].x, r[0].y), id = (int64_t)(((unsigned long long)r[1].y << 32) | r[1].x);
                x = dbl(r[1].z, r[1].w), y = dbl(r[2].x, r[2].y), z = dbl(r[2].z, r[2].w);
                vx = dbl(r[3].x, r[3].y) + al * dbl(r[4].z, r[4].w);   // velocity bias (:301-305)
                vy = dbl(r[3].z, r[3].w) + al * dbl(r[5].x, r[5].y);
                vz = dbl(r[4].x, r[4].y) + al * dbl(r[5].z, r[5].w);
            } else {
                id = (int64_t)(((unsigned long long)r[0].y << 32) | r[0].x), m = dbl(r[0].z, r[0].w);
                x = dbl(r[1].x, r[1].y), y = dbl(r[1].z, r[1].w), z = dbl(r[2].x, r[2].y);
                const double h0 = dbl(r[4].x, r[4].y), h1 = dbl(r[4].z, r[4].w), h2 = dbl(r[5].x, r[5].y);
                vx = h0 + al * (dbl(r[2].z, r[2].w) - h0);   // (:1136-1146)
                vy = h1 + al * (dbl(r[3].x, r[3].y) - h1);
                vz = h2 + al * (dbl(r[3].z, r[3].w) - h2);
            }
            // the record of galaxy k + 1 and the kept index of k + 2 are requested BEFORE the eight column stores of galaxy k:
            // vmcnt counts in order, so the wait below can leave exactly those stores outstanding (they drain under the next
            // iteration) - unless a lane skipped its stores (catalogue buffers too small), then everything is waited for
            const bool more = e + EBLOCK < total;
            asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(vx), "+v"(vy), "+v"(vz), "+v"(m), "+v"(id));   // decoded: r[] is free
            if (more) {
                issue(k1, r);
                aload_u16(k2, kept + min(e + 2 * EBLOCK, total - 1));
            }
            const bool stored = j < o.cap[t];
            emit_one(p, o, t, j, x, y, z, vx, vy, vz, m, id);
            e += EBLOCK;
            if (!more) break;
            if (__all(stored)) await_vm<8>();
            else await_vm<0>();
            touch1(k2);
            k1 = k2;
        }
        return;
    }
    for (int e = tid; e < total; e += EBLOCK) {
        const int t = e < m0 ? 0 : (e < m0 + m1 ? 1 : 2);
        const int64_t j = t == 0 ? off0 + e : (t == 1 ? off1 + (e - m0) : off2 + (e - m0 - m1));
        const int64_t i = obj_first + kept[e];
        const double al = t == 0 ? a0 : (t == 1 ? a1 : a2);
        double x, y, z, vx, vy, vz, m;
        int64_t id;
        if (dbg & 2) {   // ablation: no gather
            x = y = z = vx = vy = vz = m = (double)i;
            id = i;
        } else if (!sat && in.hrec) {
            const HaloRec &r = in.hrec[i];
            x = r.pos0, y = r.pos1, z = r.pos2;
            vx = r.vel[0] + al * r.vdev[0];
            vy = r.vel[1] + al * r.vdev[1];
            vz = r.vel[2] + al * r.vdev[2];
            m = r.mass, id = r.id;
        } else if (sat && in.prec) {
            const PartRec &r = in.prec[i];   // line 1
            x = r.pos[0], y = r.pos[1], z = r.pos[2];
            vx = r.hvel[0] + al * (r.vel0 - r.hvel[0]);
            vy = r.hvel[1] + al * (r.vel1 - r.hvel[1]);
            vz = r.hvel[2] + al * (r.vel2 - r.hvel[2]);
            m = r.mass2, id = r.id;
        } else if (!sat) {
            x = in.hpos[3 * i], y = in.hpos[3 * i + 1], z = in.hpos[3 * i + 2];
            vx = in.hvel[3 * i] + al * in.hvdev[3 * i];  // velocity bias (:301-305)
            vy = in.hvel[3 * i + 1] + al * in.hvdev[3 * i + 1];
            vz = in.hvel[3 * i + 2] + al * in.hvdev[3 * i + 2];
            m = in.hmass[i];
            id = in.hid[i];
        } else {
            x = in.ppos[3 * i], y = in.ppos[3 * i + 1], z = in.ppos[3 * i + 2];
            vx = in.phvel[3 * i] + al * (in.pvel[3 * i] - in.phvel[3 * i]);  // (:1136-1146)
            vy = in.phvel[3 * i + 1] + al * (in.pvel[3 * i + 1] - in.phvel[3 * i + 1]);
            vz = in.phvel[3 * i + 2] + al * (in.pvel[3 * i + 2] - in.phvel[3 * i + 2]);
            m = in.phmass[i];
            id = in.phid[i];
        }
        if (dbg & 1) {   // ablation: no column stores (one conditional store keeps the gather alive)
            if (x + y + z + vx + vy + vz + m == 1.2345e300 && id == 77) o.c[t][0][j] = x;
            continue;
        }
        emit_one(p, o, t, j, x, y, z, vx, vy, vz, m, id);
    }
}

// ---- device RNG for `run_hod(reseed=...)` (hod/abacus_hod.py:775-839) ------------------------------------------------
// Philox4x32-10, counter-based: the value of an object depends only on (seed, stream, object index), never on the
// launch geometry, so a catalogue sharded over GPUs draws the same numbers as the unsharded one when the caller
// passes its global index offset.  The reference draws float32 uniforms / normals from parallel_numpy_rng (third party,
// absent here: the REFERENCE's stream cannot be pinned); distributions and dtypes are the same: hrandoms, prandoms =
// float32 U[0,1); hveldev = float32 N(0,1) (or the two-sided exponential of `want_expvel`) * hsigma3d / sqrt(3) in
// float64.  THIS stream is pinned: Philox by its published known-answer vectors, the transforms by the oracle's restatement.
#include "rng_device.hpp"
__device__ __forceinline__ float laplace_f32(float rt) {   // (:799-801): -log(2 (1 - rt)) for rt >= 0.5, log(2 rt) below
    return rt >= 0.5f ? (float)(-det_log(2.0 * (1.0 - (double)rt))) : (float)det_log(2.0 * (double)rt);
}

// Stream layout (the oracle's oracle_reseed_* restate it): halo with GLOBAL index g draws a = Philox(ctr = (g lo, g hi, 0, 0))
// and b = Philox(ctr = (g lo, g hi, 1, 0)), key = (seed lo, seed hi); hrandoms = u01(a.x); Gaussian deviates by Box-Muller:
// (r0, r1) from (a.y, a.z), r2 from (b.x, b.y) - magnitude sqrt(-2 log(1 - u)), angle 2 pi u', each rounded once to float32;
// want_expvel: r_d = laplace(max(u01(a.y|a.z|a.w), 1e-30)).  hveldev = float32 deviate * hsigma3d / sqrt(3) in float64
// (:826-833).  Particle 4 q + d takes word d of Philox(ctr = (q lo, q hi, 2, 0)).
__global__ __launch_bounds__(256) void hod_reseed_halos(int64_t n, int64_t index0, unsigned long long seed,
                                                        const double *__restrict__ sigma3d, int expvel,
                                                        double *__restrict__ hrandoms, double *__restrict__ hveldev) {
    const uint2 key = make_uint2((unsigned int)seed, (unsigned int)(seed >> 32));
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const unsigned long long gi = (unsigned long long)(index0 + i);
        const uint4 a = philox4x32_10(make_uint4((unsigned int)gi, (unsigned int)(gi >> 32), 0u, 0u), key);   // stream 0
        const uint4 b = philox4x32_10(make_uint4((unsigned int)gi, (unsigned int)(gi >> 32), 1u, 0u), key);   // stream 1
        hrandoms[i] = (double)u01(a.x);
        float r[3];
        if (expvel) {
            r[0] = laplace_f32(fmaxf(u01(a.y), 1e-30f)), r[1] = laplace_f32(fmaxf(u01(a.z), 1e-30f));
            r[2] = laplace_f32(fmaxf(u01(a.w), 1e-30f));
        } else {   // Box-Muller on (0, 1] x [0, 1)
            const double m0 = sqrt(-2.0 * det_log(1.0 - (double)u01(a.y))), m1 = sqrt(-2.0 * det_log(1.0 - (double)u01(b.x)));
            double s0, c0, s1, c1;
            det_sincos2pi((double)u01(a.z), &s0, &c0);
            det_sincos2pi((double)u01(b.y), &s1, &c1);
            r[0] = (float)(m0 * c0), r[1] = (float)(m0 * s0), r[2] = (float)(m1 * c1);
            (void)s1;
        }
        const double sg = sigma3d ? sigma3d[i] : 0.0;
#pragma unroll
        for (int d = 0; d < 3; d++) hveldev[3 * i + d] = (double)r[d] * sg / 1.7320508075688772;   // r2 * hsigma3d / sqrt(3)
    }
}

__global__ __launch_bounds__(256) void hod_reseed_particles(int64_t n, int64_t index0, unsigned long long seed,
                                                            double *__restrict__ prandoms) {
    const uint2 key = make_uint2((unsigned int)seed, (unsigned int)(seed >> 32));
    // one Philox call serves the four particles 4q .. 4q+3 (q is a GLOBAL index: sharding-invariant)
    const int64_t q0 = index0 >> 2, q1 = (index0 + n + 3) >> 2;
    for (int64_t q = q0 + (int64_t)blockIdx.x * 256 + threadIdx.x; q < q1; q += (int64_t)gridDim.x * 256) {
        const uint4 a = philox4x32_10(make_uint4((unsigned int)q, (unsigned int)((unsigned long long)q >> 32), 2u, 0u), key);
        const unsigned int w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int64_t i = 4 * q + d - index0;
            if (i >= 0 && i < n) prandoms[i] = (double)u01(w[d]);
        }
    }
}

// ---- NFW satellites (gen_sats_nfw / compute_fast_NFW / getPointsOnSphere, hod/GRAND_HOD.py:417-822) ------------------
// The reference draws from NumPy's unseeded per-thread generators, so only the DISTRIBUTIONS can be matched:
// N_sat ~ Poisson(n_sat(M) * ic) per halo and tracer (no particle weights on this path, :634-706); isotropic
// direction (:433-441); radius r = eta * Rvir with eta = NFW_draw[k] / c for a random table entry k with
// NFW_draw[k] <= c (rejection, :503-508; times nfw_rescale), or with probability exp_frac an exponential of scale
// exp_scale over c (:499-501); velocity ~ N(v_halo, (0.577 f_sigv vrms)^2) per component (:513-516); RSD
// z = (z + vz / velz2kms) mod L (:787-789: [0, L), unlike the particle path).  Counter-based Philox streams keyed by
// (seed, global halo index, tracer, satellite rank) make a run reproducible and sharding-invariant.
struct NfwArgs {
    unsigned long long seed;
    double f_sigv[3];
    double exp_frac, exp_scale, nfw_rescale;
    int64_t halo_index0;
    int64_t n_draw;
};

struct PhiloxStream {   // 4 words per block; blocks (c0, c1, stream, block#)
    uint4 ctr, buf;
    uint2 key;
    int have;
    __device__ PhiloxStream(unsigned long long seed, unsigned long long index, unsigned int stream) {
        ctr = make_uint4((unsigned int)index, (unsigned int)(index >> 32), stream, 0u);
        key = make_uint2((unsigned int)seed, (unsigned int)(seed >> 32));
        have = 0;
        buf = make_uint4(0u, 0u, 0u, 0u);
    }
    __device__ unsigned int next() {
        if (have == 0) {
            buf = philox4x32_10(ctr, key);
            ctr.w++;
            have = 4;
        }
        const unsigned int v = have == 4 ? buf.x : (have == 3 ? buf.y : (have == 2 ? buf.z : buf.w));
        have--;
        return v;
    }
    __device__ double uniform() {   // (0, 1): 53 bits would need two words; 32 bits are plenty for these draws
        return ((double)next() + 0.5) * 2.3283064365386963e-10;
    }
};

__device__ int poisson_draw(PhiloxStream &g, double lam) {
    if (!(lam > 0.0)) return 0;
    if (lam < 10.0) {   // multiplication method
        const double L = exp(-lam);
        int k = 0;
        double pr = g.uniform();
        while (pr > L && k < 1000) {
            k++;
            pr *= g.uniform();
        }
        return k;
    }
    // transformed rejection (Hoermann 1993, PTRS)
    const double slam = sqrt(lam), loglam = log(lam), b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b;
    const double invalpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2.0);
    for (int it = 0; it < 1000; it++) {
        const double U = g.uniform() - 0.5, V = g.uniform();
        const double us = 0.5 - fabs(U);
        const double kf = floor((2.0 * a / us + b) * U + lam + 0.43);
        if (us >= 0.07 && V <= vr) return (int)kf;
        if (kf < 0.0 || (us < 0.013 && V > us)) continue;
        if (log(V) + log(invalpha) - log(a / (us * us) + b) <= -lam + kf * loglam - lgamma(kf + 1.0)) return (int)kf;
    }
    return (int)lam;
}

// expected satellites per halo of the NFW path (:634-706) and their Poisson draws
__global__ __launch_bounds__(256) void hod_nfw_count(HodPtrs a, abacus_hod_params p, NfwArgs nf,
                                                     unsigned int *__restrict__ nL, unsigned int *__restrict__ nE,
                                                     unsigned int *__restrict__ nQ) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.nh; i += (int64_t)gridDim.x * 256) {
        const double m = a.hmass[i], dc = load1(a.hdeltac, i, 0.0), fe = load1(a.hfenv, i, 0.0),
                     sh = load1(a.hshear, i, 0.0);
        const unsigned long long gi = (unsigned long long)(nf.halo_index0 + i);
        if (p.want_LRG) {
            const double M1 = pow(10.0, p.L_logM1 + p.L_Asat * dc + p.L_Bsat * fe);
            const double lc = p.L_logM_cut + p.L_Acent * dc + p.L_Bcent * fe;
            const double base = n_sat_LRG_modified(m, lc, pow(10.0, lc), M1, p.L_sigma, p.L_alpha, p.L_kappa) * p.L_ic;
            PhiloxStream g(nf.seed, gi, 10u);
            nL[i] = (unsigned int)poisson_draw(g, base);
        }
        if (p.want_ELG) {
            double M1 = pow(10.0, p.E_logM1 + p.E_Asat * dc + p.E_Bsat * fe + p.E_Csat * sh);
            const double lc = p.E_logM_cut + p.E_Acent * dc + p.E_Bcent * fe + p.E_Ccent * sh;
            double alpha = p.E_alpha;
            const int8_t kc = a.keep_c[i];   // ELG conformity (:664-693)
            if (kc == 1) M1 = pow(10.0, p.E_logM1_EL + p.E_Asat * dc + p.E_Bsat * fe), alpha = p.E_alpha_EL;
            else if (kc == 2) M1 = pow(10.0, p.E_logM1_EE + p.E_Asat * dc + p.E_Bsat * fe), alpha = p.E_alpha_EE;
            const double base = N_sat_generic(m, pow(10.0, lc), p.E_kappa, M1, alpha, p.E_A_s) * p.E_ic;
            PhiloxStream g(nf.seed, gi, 11u);
            nE[i] = (unsigned int)poisson_draw(g, base);
        }
        if (p.want_QSO) {
            const double M1 = pow(10.0, p.Q_logM1 + p.Q_Asat * dc + p.Q_Bsat * fe);
            const double lc = p.Q_logM_cut + p.Q_Acent * dc + p.Q_Bcent * fe;
            const double base = N_sat_generic(m, pow(10.0, lc), p.Q_kappa, M1, p.Q_alpha, 1.0) * p.Q_ic;   // A_s = 1 (:697-703)
            PhiloxStream g(nf.seed, gi, 12u);
            nQ[i] = (unsigned int)poisson_draw(g, base);
        }
    }
}

// one thread per satellite of tracer `t`: host halo by binary search in the exclusive offsets
__global__ __launch_bounds__(256) void hod_nfw_emit(int t, int64_t nsat, int64_t nh, const int64_t *__restrict__ off,
                                                    int64_t out0, const double *__restrict__ hpos,
                                                    const double *__restrict__ hvel, const double *__restrict__ hmass,
                                                    const int64_t *__restrict__ hid, const double *__restrict__ hvrms,
                                                    const double *__restrict__ hc, const double *__restrict__ hrvir,
                                                    const double *__restrict__ draw, abacus_hod_params p, NfwArgs nf,
                                                    OutCols o) {
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < nsat; j += (int64_t)gridDim.x * 256) {
        int64_t lo = 0, hi = nh;   // largest h with off[h] <= j
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (off[mid] <= j) lo = mid;
            else hi = mid;
        }
        const int64_t h = lo;
        const unsigned long long rank = (unsigned long long)(j - off[h]);
        PhiloxStream g(nf.seed ^ (rank * 0x9E3779B97F4A7C15ull), (unsigned long long)(nf.halo_index0 + h), 20u + (unsigned int)t);
        // direction (:433-441)
        const double u1 = g.uniform(), u2 = g.uniform();
        const double ra = u1 * 6.283185307179586, dec = 3.141592653589793 - acos(-1.0 + 2.0 * u2);
        const double sd = sin(dec), ux = sd * cos(ra), uy = sd * sin(ra), uz = cos(dec);
        // radius (:497-510)
        const double c = hc[h];
        double eta;
        if (g.uniform() < nf.exp_frac) {
            eta = -nf.exp_scale * log(g.uniform()) / c;
        } else {
            double d = 0.0;
            bool ok = false;
            for (int it = 0; it < 256 && !ok; it++) {
                const int64_t k = (int64_t)(g.uniform() * (double)nf.n_draw);
                d = draw[k < nf.n_draw ? k : nf.n_draw - 1];
                ok = !(d > c);
            }
            if (!ok) d = c * g.uniform();   // a table without entries below this concentration
            eta = d / c * nf.nfw_rescale;
        }
        const double r = eta * hrvir[h];
        double x = hpos[3 * h] + ux * r, y = hpos[3 * h + 1] + uy * r, z = hpos[3 * h + 2] + uz * r;
        // velocity (:511-516): Box-Muller pairs
        const double sig = hvrms[h] * 0.577 * nf.f_sigv[t];
        const double m0 = sqrt(-2.0 * log(g.uniform())), m1 = sqrt(-2.0 * log(g.uniform()));
        const double a0 = 6.283185307179586 * g.uniform(), a1 = 6.283185307179586 * g.uniform();
        const double vx = hvel[3 * h] + sig * m0 * cos(a0), vy = hvel[3 * h + 1] + sig * m0 * sin(a0),
                     vz = hvel[3 * h + 2] + sig * m1 * cos(a1);
        if (p.rsd) {   // (z + vz * inv_velz2kms) % lbox, Python modulo (:787-789)
            z = z + vz * p.inv_velz2kms;
            z = z - floor(z / p.lbox) * p.lbox;
        }
        const int64_t q = out0 + j;
        if (q < o.cap[t]) {
            o.c[t][0][q] = x, o.c[t][1][q] = y, o.c[t][2][q] = z;
            o.c[t][3][q] = vx, o.c[t][4][q] = vy, o.c[t][5][q] = vz;
            o.c[t][6][q] = hmass[h];
            o.id[t][q] = hid[h];
        }
    }
}

// ---- compute_ngal (hod/abacus_hod.py:861-1179) -----------------------------------------------------------------------
// The reference sums  hist[cell] * n(centre of cell)  over a 100^3 (LRG, QSO) or 100^4 (ELG) weighted halo histogram.
// The same sum runs over the HALOS here: sum_i multis[i] * n(centre of the cell of halo i) - identical terms, no
// 800-MB histogram, one streaming pass over 12 bytes per halo.  The cell indices come from the host once
// (np.histogramdd's own edge rule), 255 = outside the histogram range.
struct NgalTables {
    const double *Mh, *deltac, *fenv, *shear;   // 10**logM centre, deltac / fenv / shear centres
};

__global__ __launch_bounds__(256) void hod_ngal(int64_t n, const uchar4 *__restrict__ bins,
                                                const double *__restrict__ multis, NgalTables tb, abacus_hod_params p,
                                                double *__restrict__ out) {
    double acc[6] = {0, 0, 0, 0, 0, 0};   // cent L, E, Q; sat L, E, Q
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uchar4 b = bins[i];
        if (b.x == 255 || b.y == 255 || b.z == 255) continue;
        const double w = multis[i], Mh = tb.Mh[b.x], dc = tb.deltac[b.y], fe = tb.fenv[b.z];
        if (p.want_LRG) {   // _compute_ngal_lrg (:980-1033)
            const double lc = p.L_logM_cut + p.L_Acent * dc + p.L_Bcent * fe;
            const double M1 = pow(10.0, p.L_logM1 + p.L_Asat * dc + p.L_Bsat * fe);
            acc[0] += w * n_cen_LRG(Mh, lc, p.L_sigma) * p.L_ic;
            acc[3] += w * n_sat_LRG_modified(Mh, lc, pow(10.0, lc), M1, p.L_sigma, p.L_alpha, p.L_kappa) * p.L_ic;
        }
        if (p.want_QSO) {   // _compute_ngal_qso (:1135-1179)
            const double lc = p.Q_logM_cut + p.Q_Acent * dc + p.Q_Bcent * fe;
            const double M1 = pow(10.0, p.Q_logM1 + p.Q_Asat * dc + p.Q_Bsat * fe);
            acc[2] += w * N_cen_QSO(Mh, lc, p.Q_sigma) * p.Q_ic;
            acc[5] += w * N_sat_generic(Mh, pow(10.0, lc), p.Q_kappa, M1, p.Q_alpha, 1.0) * p.Q_ic;
        }
        if (p.want_ELG && b.w != 255) {   // _compute_ngal_elg (:1035-1132)
            const double sh = tb.shear[b.w];
            const double lc = p.E_logM_cut + p.E_Acent * dc + p.E_Bcent * fe + p.E_Ccent * sh;
            const double M1 = pow(10.0, p.E_logM1 + p.E_Asat * dc + p.E_Bsat * fe + p.E_Csat * sh);
            const double ncent = N_cen_ELG_v1(Mh, p.E_p_max, p.E_Q, lc, p.E_sigma, p.E_gamma) * p.E_ic;
            const double Mc = pow(10.0, lc);
            const double nsat = N_sat_generic(Mh, Mc, p.E_kappa, M1, p.E_alpha, p.E_A_s) * p.E_ic;
            const double M1c = pow(10.0, p.E_logM1_EE + p.E_Asat * dc + p.E_Bsat * fe + p.E_Csat * sh);
            const double nconf = N_sat_generic(Mh, Mc, p.E_kappa, M1c, p.E_alpha_EE, p.E_A_s) * p.E_ic;
            acc[1] += w * ncent;
            acc[4] += w * (nsat * (1 - ncent) + nconf * ncent);
        }
    }
    __shared__ double red[4][6];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 6; q++) {
        double v = acc[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[wv][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6) atomicAdd(&out[threadIdx.x], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

struct ColRange {
    double lo = 0, hi = 0;
};
struct HodRanges {
    ColRange hdeltac, hfenv, hshear, pdeltac, pfenv, pshear, pranks[4];
};

// per-block minimum / maximum of a column, NaNs ignored (fmin / fmax); the host folds the partials
__global__ __launch_bounds__(256) void hod_minmax(const double *__restrict__ src, int64_t n, double *__restrict__ part) {
    double lo = INFINITY, hi = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = src[i];
        lo = fmin(lo, v), hi = fmax(hi, v);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lo = fmin(lo, __shfl_xor(lo, off, 64)), hi = fmax(hi, __shfl_xor(hi, off, 64));
    __shared__ double red[4][2];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][0] = lo, red[threadIdx.x >> 6][1] = hi;
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = fmin(fmin(red[0][0], red[1][0]), fmin(red[2][0], red[3][0]));
        part[2 * blockIdx.x + 1] = fmax(fmax(red[0][1], red[1][1]), fmax(red[2][1], red[3][1]));
    }
}

}  // namespace

// ---- handle ---------------------------------------------------------------------------------------------
struct abacus_hod_state {
    int64_t nh = 0, np = 0;
    bool owns = true;
    // staged inputs (device)
    double *hpos = nullptr, *hvel = nullptr, *hmass = nullptr, *hmultis = nullptr, *hrandoms = nullptr,
           *hveldev = nullptr, *hdeltac = nullptr, *hfenv = nullptr, *hshear = nullptr;
    double *hsigma3d = nullptr;   // optional (abacus_hod_set_sigma3d): device reseed and NFW satellites
    bool owns_sigma = false;
    double *hc = nullptr, *hrvir = nullptr;   // optional (abacus_hod_set_profile): NFW satellites
    DevBuf nfw_counts, nfw_offsets, nfw_draw, nfw_scan;
    DevBuf ngal_bins, ngal_tables, ngal_out;   // compute_ngal: cell indices, 4 x nbin centres, 6 sums
    int ngal_nbin = 0;
    int64_t *hid = nullptr;
    double *ppos = nullptr, *pvel = nullptr, *phvel = nullptr, *phmass = nullptr, *pweights = nullptr,
           *prandoms = nullptr, *pdeltac = nullptr, *pfenv = nullptr, *pshear = nullptr, *pranks = nullptr,
           *pranksv = nullptr, *pranksp = nullptr, *pranksr = nullptr;
    int64_t *phid = nullptr, *pinds = nullptr;
    // work arrays
    int ntile_c = 0, ntile_s = 0;
    int8_t *keep_c = nullptr, *keep_s = nullptr;
    int *sb_counts = nullptr;   // [(nsb_c + nsb_s)][4]
    int *q_count = nullptr;                                  // [ntile_c + ntile_s] survivors of the float32 filter
    unsigned short *queue_c = nullptr, *queue_s = nullptr;   // tile-local indices, one TILE-sized slice per tile
    unsigned short *kept_c = nullptr, *kept_s = nullptr;     // kept lists, one SB_OBJ-sized slice per superblock
    int nsb_c = 0, nsb_s = 0;   // superblocks of the current populate (sb_tiles tiles each)
    int sb_tiles = SB_TILES_DENSE;
    // mass-sorted key index (sparse mixes; hod_deal)
    DevBuf index_idx, index_scratch, index_tmp, index_last;
    std::vector<unsigned int> upper_h, upper_p;   // host: number of objects with (bin << 9 | code) <= v
    bool index_ok = false;
    int key_uses = 0;           // populates since the keys were (re)built
    bool q_zero = false;        // the per-tile candidate counters are known to be zero
    int64_t last_cand[2] = {-1, -1};
    bool kept_valid = false;    // the kept lists (superblocks of kept_sb_tiles tiles) name exactly the non-zero mask bytes
    int kept_sb_tiles = 0;
    int64_t *d_totals = nullptr;  // 6
    int64_t *h_totals = nullptr;  // pinned, 6
    // outputs
    DevBuf out[3];
    int64_t cap[3] = {0, 0, 0};
    int64_t counts[6] = {0, 0, 0, 0, 0, 0};
    abacus_hod_params params;
    bool have_run = false, counts_valid = false;
    // float32 shadows of the columns the filter streams (owned catalogues only; see hod_shadow)
    DevBuf shadow;
    FiltCols fc = {};
    bool shadow_ok = false, shadow_rand_ok = false;
    DevBuf hrec, prec;          // packed records (owned catalogues)
    bool rec_ok = false, rec_rand_ok = false;   // rec_rand_ok: the records hold the current randoms / hveldev
    HodRanges ranges;           // value ranges of the environment / rank columns (envelope table of the two-stage filter)
    bool ranges_ok = false;
    DevBuf keys;                // packed filter keys (hod_build_keys): [ntile_c * TILE][ntile_s * TILE] uint32
    bool keys_ok = false;
};

namespace {


// host side of the float32 rejection filter: constants rounded so that every bound stays an upper bound
Filt make_filter(const abacus_hod_params &p, const SatPre &pre) {
    Filt F;
    memset(&F, 0, sizeof F);
    auto up = [](double v) { return (float)(v * (v >= 0 ? 1.000001 : 0.999999)); };   // >= v after rounding
    auto finite = [](double v) { return std::isfinite(v); };
    bool ok = true;
    auto tracer = [&](bool want, double lc, double Ac, double Bc, double Cc, double sigma, double ic) {
        if (!want) return;
        ok = ok && finite(lc) && finite(Ac) && finite(Bc) && finite(Cc) && finite(sigma) && sigma > 1e-3 && finite(ic) &&
             ic >= 0;
    };
    tracer(p.want_LRG, p.L_logM_cut, p.L_Acent, p.L_Bcent, 0, p.L_sigma, p.L_ic);
    tracer(p.want_ELG, p.E_logM_cut, p.E_Acent, p.E_Bcent, p.E_Ccent, p.E_sigma, p.E_ic);
    tracer(p.want_QSO, p.Q_logM_cut, p.Q_Acent, p.Q_Bcent, 0, p.Q_sigma, p.Q_ic);
    if (p.want_ELG) ok = ok && finite(p.E_p_max) && finite(p.E_Q) && p.E_Q != 0 && finite(p.E_gamma);
    F.cent_ok = ok;
    F.L_lc = (float)p.L_logM_cut, F.L_Ac = (float)p.L_Acent, F.L_Bc = (float)p.L_Bcent;
    F.L_inv_s = (float)(1.0 / (1.41421356 * p.L_sigma)), F.L_ic = up(p.L_ic);
    F.E_lc = (float)p.E_logM_cut, F.E_Ac = (float)p.E_Acent, F.E_Bc = (float)p.E_Bcent, F.E_Cc = (float)p.E_Ccent;
    F.E_c_phi = up(std::max(2.0 * (p.E_p_max - 1.0 / p.E_Q), 0.0) * 0.3989422804014327 / p.E_sigma);
    F.E_half_inv_s2 = (float)(0.5 / (p.E_sigma * p.E_sigma)), F.E_ic = up(p.E_ic);
    F.E_gs = (float)(p.E_gamma / p.E_sigma / 1.4142135623730951);
    F.Q_lc = (float)p.Q_logM_cut, F.Q_Ac = (float)p.Q_Acent, F.Q_Bc = (float)p.Q_Bcent;
    F.Q_inv_s = (float)(1.0 / (1.41421356 * p.Q_sigma)), F.Q_ic = up(p.Q_ic);
    // satellites: the arithmetic bound only when every wanted tracer has particle-independent M1 / M_cut (`sok`); the
    // envelope table of the two-stage filter needs just finite parameters, positive masses and alpha >= 0 (`sbasic`)
    bool sok = ok, sbasic = ok;
    auto sat = [&](bool want, int is_const, double M1, double alpha, double kappa, double Mcut) {
        if (!want) return;
        const bool b = finite(M1) && M1 > 0 && finite(alpha) && alpha >= 0 && finite(kappa) && finite(Mcut);
        sbasic = sbasic && b;
        sok = sok && is_const && b;
    };
    sat(p.want_LRG, pre.L_const, pre.L_M1, p.L_alpha, p.L_kappa, pre.L_Mcut);
    sat(p.want_ELG, pre.E_const, pre.E_M1, p.E_alpha, p.E_kappa, pre.E_Mcut);
    sat(p.want_ELG, pre.E_const, pre.E_M1_EL, p.E_alpha_EL, p.E_kappa, pre.E_Mcut);
    sat(p.want_ELG, pre.E_const, pre.E_M1_EE, p.E_alpha_EE, p.E_kappa, pre.E_Mcut);
    sat(p.want_QSO, pre.Q_const, pre.Q_M1, p.Q_alpha, p.Q_kappa, pre.Q_Mcut);
    if (p.want_ELG) sok = sok && finite(p.E_A_s) && p.E_A_s >= 0, sbasic = sbasic && finite(p.E_A_s) && p.E_A_s >= 0;
    F.sat_ok = sok;
    F.sat_basic = sbasic;
    F.L_invM1 = up(1.0 / pre.L_M1), F.L_alpha = (float)p.L_alpha;
    F.E_invM1[0] = up(1.0 / pre.E_M1), F.E_invM1[1] = up(1.0 / pre.E_M1_EL), F.E_invM1[2] = up(1.0 / pre.E_M1_EE);
    F.E_alpha[0] = (float)p.E_alpha, F.E_alpha[1] = (float)p.E_alpha_EL, F.E_alpha[2] = (float)p.E_alpha_EE;
    F.E_As = up(p.E_A_s);
    F.Q_invM1 = up(1.0 / pre.Q_M1), F.Q_alpha = (float)p.Q_alpha;
    const double Ls[4] = {p.L_s, p.L_s_v, p.L_s_p, p.L_s_r}, Es[4] = {p.E_s, p.E_s_v, p.E_s_p, p.E_s_r},
                 Qs[4] = {p.Q_s, p.Q_s_v, p.Q_s_p, p.Q_s_r};
    for (int q = 0; q < 4; q++) {
        F.L_s[q] = (float)Ls[q], F.E_s[q] = (float)Es[q], F.Q_s[q] = (float)Qs[q];
        if (p.enable_ranks) {
            const bool b = finite(Ls[q]) && finite(Es[q]) && finite(Qs[q]);
            F.sat_ok = F.sat_ok && b, F.sat_basic = F.sat_basic && b;
        }
    }
    // kappa*M_cut exactly as n_sat_* forms it (FP64 product)
    F.L_kMcut = p.L_kappa * pre.L_Mcut, F.E_kMcut = p.E_kappa * pre.E_Mcut, F.Q_kMcut = p.Q_kappa * pre.Q_Mcut;
    return F;
}

// Value ranges of the staged environment / rank columns (NaNs ignored), measured once per catalogue: the envelope
// table bounds every object's occupation by the occupation at the most favourable environment in these ranges.
inline void prod_range(double c, const ColRange &r, double &lo, double &hi) {   // range of c * x, x in r
    const double a = c * r.lo, b = c * r.hi;
    if (c == 0) return;   // the reference forms 0 * x: exactly 0 for finite x
    lo += std::min(a, b), hi += std::max(a, b);
}

// host side of the two-stage filter: an ENVELOPE table - for every float32 mass bin an upper bound of the summed
// occupation of the wanted tracers over the masses of the bin AND over the staged ranges of deltac / fenv / shear, so the
// streaming loop reads mass, multiplicity / weight and random only (12 B per object) for any HOD:
//   erfc forms (LRG / QSO centrals, the LRG satellites' n_cen factor): largest at the bin's upper edge and the smallest
//   logM_cut + A d + B f of the range;  ELG centrals 2 (p_max - 1/Q) phi(x) Phi(gamma x), x = (logM - logM_cut') / sigma
//   (not monotone): Gaussian at the smallest |logM - logM_cut'| the bin and the range admit, Phi at the largest gamma x;
//   power laws ((M - kappa M_cut') / M1')^alpha: upper edge, smallest kappa M_cut' and smallest M1' of the range, the
//   largest of the three conformity variants for ELG.  Rank modulation: 1 + sum |s_q| max |rank_q|.
Cheap make_cheap(const abacus_hod_params &p, const Filt &F, const HodRanges &R) {
    Cheap c;
    memset(&c, 0, sizeof c);
    const bool one_stage = option("hod_one_stage") != 0;
    c.c_ok = F.cent_ok && (p.want_LRG || p.want_ELG || p.want_QSO) && !one_stage;
    c.s_ok = F.sat_basic && (p.want_LRG || p.want_ELG || p.want_QSO) && !one_stage;
    auto up = [](double v) { return std::max((float)(v * 1.00001), 1e-30f); };
    // ranges of logM_cut' (centrals: halo columns; satellites: particle columns) and of logM1' per tracer / variant
    struct LR {
        double lo, hi;
    };
    auto lin = [&](double base, double A, const ColRange &d, double B, const ColRange &f, double Cc, const ColRange &sh) {
        LR r{base, base};
        prod_range(A, d, r.lo, r.hi), prod_range(B, f, r.lo, r.hi), prod_range(Cc, sh, r.lo, r.hi);
        return r;
    };
    const LR Lc_h = lin(p.L_logM_cut, p.L_Acent, R.hdeltac, p.L_Bcent, R.hfenv, 0, R.hshear);
    const LR Ec_h = lin(p.E_logM_cut, p.E_Acent, R.hdeltac, p.E_Bcent, R.hfenv, p.E_Ccent, R.hshear);
    const LR Qc_h = lin(p.Q_logM_cut, p.Q_Acent, R.hdeltac, p.Q_Bcent, R.hfenv, 0, R.hshear);
    const LR Lc_p = lin(p.L_logM_cut, p.L_Acent, R.pdeltac, p.L_Bcent, R.pfenv, 0, R.pshear);
    const LR Ec_p = lin(p.E_logM_cut, p.E_Acent, R.pdeltac, p.E_Bcent, R.pfenv, p.E_Ccent, R.pshear);
    const LR Qc_p = lin(p.Q_logM_cut, p.Q_Acent, R.pdeltac, p.Q_Bcent, R.pfenv, 0, R.pshear);
    const LR L1 = lin(p.L_logM1, p.L_Asat, R.pdeltac, p.L_Bsat, R.pfenv, 0, R.pshear);
    const LR E1 = lin(p.E_logM1, p.E_Asat, R.pdeltac, p.E_Bsat, R.pfenv, p.E_Csat, R.pshear);
    const LR E1L = lin(p.E_logM1_EL, p.E_Asat, R.pdeltac, p.E_Bsat, R.pfenv, 0, R.pshear);   // no Csat term (:1006-1035)
    const LR E1E = lin(p.E_logM1_EE, p.E_Asat, R.pdeltac, p.E_Bsat, R.pfenv, 0, R.pshear);
    const LR Q1 = lin(p.Q_logM1, p.Q_Asat, R.pdeltac, p.Q_Bsat, R.pfenv, 0, R.pshear);
    double dec = 1.0;
    if (p.enable_ranks) {
        const double Ls[4] = {p.L_s, p.L_s_v, p.L_s_p, p.L_s_r}, Es[4] = {p.E_s, p.E_s_v, p.E_s_p, p.E_s_r},
                     Qs[4] = {p.Q_s, p.Q_s_v, p.Q_s_p, p.Q_s_r};
        for (int q = 0; q < 4; q++) {
            double m = 0;
            if (p.want_LRG) m = std::max(m, std::fabs(Ls[q]));
            if (p.want_ELG) m = std::max(m, std::fabs(Es[q]));
            if (p.want_QSO) m = std::max(m, std::fabs(Qs[q]));
            dec += m * std::max(std::fabs(R.pranks[q].lo), std::fabs(R.pranks[q].hi));
        }
    }
    if (!std::isfinite(dec)) c.s_ok = 0;
    c.dec_max = (float)(dec * 1.0001);
    auto half_erfc = [](double lM, double lc, double sigma) { return 0.5 * std::erfc((lc - lM) / (1.41421356 * sigma)); };
    auto powa = [](double x, double a) { return a == 1.0 ? x : std::pow(x, a); };
    // ((M - kappa M_cut') / M1')^alpha at its largest: smallest kappa * 10^lc and smallest 10^l1 of the ranges
    auto plaw = [&](double M, double kappa, const LR &lc, const LR &l1, double alpha) {
        const double kM = std::min(kappa * std::pow(10.0, lc.lo), kappa * std::pow(10.0, lc.hi));
        const double x = M - kM;
        return x < 0 ? 0.0 : powa(x / std::pow(10.0, l1.lo), alpha);
    };
    double lM_prev = -INFINITY;   // bin 0 also takes every smaller mass
    for (int j = 0; j < CH_NLEV; j++) {
        const uint32_t bits = (uint32_t)(CH_BASE + j + 1) << CH_SHIFT;   // upper edge of bin j (exclusive)
        float Tf;
        memcpy(&Tf, &bits, 4);
        const double T = (double)Tf, lM = std::log10(T);
        double bc = 0, bs = 0;
        if (c.c_ok) {
            if (p.want_LRG) bc += half_erfc(lM, Lc_h.lo, p.L_sigma) * p.L_ic;
            if (p.want_QSO) bc += half_erfc(lM, Qc_h.lo, p.Q_sigma) * p.Q_ic;   // 0.5 (1 + erf(u)) = 0.5 erfc(-u)
            if (p.want_ELG) {
                // d = logM - logM_cut' over the bin (its lower edge widened by the round-up of the shadow mass) and the range
                const double dl = (lM_prev - 1e-6) - Ec_h.hi, dh = lM - Ec_h.lo;
                const double dmin = (dl <= 0 && dh >= 0) ? 0.0 : std::min(std::fabs(dl), std::fabs(dh));
                const double phi = 0.3989422804014327 / p.E_sigma * std::exp(-(dmin * dmin) / 2 / (p.E_sigma * p.E_sigma));
                const double xmax = std::max(p.E_gamma * dl / p.E_sigma, p.E_gamma * dh / p.E_sigma);
                const double Phi = std::isfinite(xmax) ? 0.5 * (1 + std::erf(xmax / 1.4142135623730951)) : 1.0;
                bc += std::max(2.0 * (p.E_p_max - 1.0 / p.E_Q), 0.0) * phi * Phi * p.E_ic;
            }
        }
        if (c.s_ok) {
            if (p.want_LRG) bs += plaw(T, p.L_kappa, Lc_p, L1, p.L_alpha) * half_erfc(lM, Lc_p.lo, p.L_sigma) * p.L_ic;
            if (p.want_ELG) {
                const double v = std::max(plaw(T, p.E_kappa, Ec_p, E1, p.E_alpha),
                                          std::max(plaw(T, p.E_kappa, Ec_p, E1L, p.E_alpha_EL), plaw(T, p.E_kappa, Ec_p, E1E, p.E_alpha_EE)));
                bs += p.E_A_s * v * p.E_ic;
            }
            if (p.want_QSO) bs += plaw(T, p.Q_kappa, Qc_p, Q1, p.Q_alpha) * p.Q_ic;
        }
        if (!std::isfinite(bc)) c.c_ok = 0;
        if (!std::isfinite(bs)) c.s_ok = 0;
        c.Bc[j] = up(bc), c.Bs[j] = up(bs);
        lM_prev = lM;
    }
    return c;
}

// threshold codes of the 16-bit key filter from the envelope table (see hod_build_keys): key bin b = table level
// K16_LEV0 + b; bin 0 also covers every level below it, bin 127 is never rejected
KeyTab make_keytab(const Cheap &ch) {
    KeyTab kt;
    for (int sat = 0; sat < 2; sat++) {
        const float *B = sat ? ch.Bs : ch.Bc;
        const float dec = sat ? ch.dec_max : 1.0f;   // folded into the table
        for (int b = 0; b < 128; b++) {
            float v = INFINITY;
            if (b == 0) {
                v = 0.f;
                for (int l = 0; l <= K16_LEV0; l++) v = std::fmax(v, B[l]);
            } else if (b < 127 && K16_LEV0 + b < CH_NLEV) {
                v = B[K16_LEV0 + b];
            }
            (sat ? kt.s : kt.c)[b] = (unsigned short)k16_code(v * dec * 1.0001f);   // NaN / inf saturate at 511: never rejected
        }
    }
    return kt;
}

template <class T>
int upload(T *&dst, const T *src, int64_t n, bool on_device) {
    if (src == nullptr) {
        dst = nullptr;
        return 0;
    }
    if (on_device) {
        dst = const_cast<T *>(src);
        return 0;
    }
    HIP_TRY(hipMalloc((void **)&dst, (n > 0 ? n : 1) * sizeof(T)));
    HIP_TRY(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyHostToDevice, stream()));
    return 0;
}

int set_capacity(abacus_hod_state *st, int t, int64_t cap) {
    if (cap < 1024) cap = 1024;
    cap = (cap + 255) & ~int64_t(255);  // columns stay 2 KiB aligned
    ABACUS_TRY(st->out[t].reserve((size_t)cap * 8 * 8));
    st->cap[t] = cap;
    return 0;
}

OutCols out_cols(abacus_hod_state *st) {
    OutCols o;
    for (int t = 0; t < 3; t++) {
        double *base = st->out[t].as<double>();
        for (int c = 0; c < 7; c++) o.c[t][c] = base + (int64_t)c * st->cap[t];
        o.id[t] = reinterpret_cast<int64_t *>(base + (int64_t)7 * st->cap[t]);
        o.cap[t] = st->cap[t];
    }
    return o;
}

// packed records of an owned catalogue: built once; the random / hveldev fields rewritten after a reseed or an update
int build_records(abacus_hod_state *st) {
    const bool norec = option("hod_norec") != 0;
    if (!st->owns || norec || (st->rec_ok && st->rec_rand_ok)) return 0;
    ABACUS_TRY(st->hrec.reserve((size_t)std::max<int64_t>(st->nh, 1) * sizeof(HaloRec)));
    ABACUS_TRY(st->prec.reserve((size_t)std::max<int64_t>(st->np, 1) * sizeof(PartRec)));
    RecSrc c;
    c.hpos = st->hpos, c.hvel = st->hvel, c.hvdev = st->hveldev, c.hmass = st->hmass, c.hmultis = st->hmultis,
    c.hrandoms = st->hrandoms, c.hdeltac = st->hdeltac, c.hfenv = st->hfenv, c.hshear = st->hshear, c.hid = st->hid;
    c.ppos = st->ppos, c.pvel = st->pvel, c.phvel = st->phvel, c.phmass = st->phmass, c.pweights = st->pweights,
    c.prandoms = st->prandoms, c.pdeltac = st->pdeltac, c.pfenv = st->pfenv, c.pshear = st->pshear, c.phid = st->phid;
    c.pinds = st->pinds;
    c.pranks[0] = st->pranks, c.pranks[1] = st->pranksv, c.pranks[2] = st->pranksp, c.pranks[3] = st->pranksr;
    const int grid = (int)std::min<int64_t>(std::max<int64_t>(ceil_div(std::max(st->nh, st->np), 256), 1), 8192);
    if (!st->rec_ok)
        ABACUS_LAUNCH("hod_build_recs", hod_build_recs<false>, dim3(grid), dim3(256), 0, st->nh, st->np, c, st->hrec.as<HaloRec>(),
                      st->prec.as<PartRec>());
    else
        ABACUS_LAUNCH("hod_refresh_recs", hod_build_recs<true>, dim3(grid), dim3(256), 0, st->nh, st->np, c, st->hrec.as<HaloRec>(),
                      st->prec.as<PartRec>());
    st->rec_ok = st->rec_rand_ok = true;
    return 0;
}

SatPre make_pre(const abacus_hod_params *p) {
    // particle-independent 10**x values, with libm's pow (the function the CPU path uses for every particle)
    SatPre pre;
    memset(&pre, 0, sizeof pre);
    pre.L_const = p->L_Acent == 0 && p->L_Asat == 0 && p->L_Bcent == 0 && p->L_Bsat == 0;
    pre.E_const = p->E_Acent == 0 && p->E_Asat == 0 && p->E_Bcent == 0 && p->E_Bsat == 0 && p->E_Ccent == 0 &&
                  p->E_Csat == 0;
    pre.Q_const = p->Q_Acent == 0 && p->Q_Asat == 0 && p->Q_Bcent == 0 && p->Q_Bsat == 0;
    pre.L_M1 = pow(10.0, p->L_logM1), pre.L_Mcut = pow(10.0, p->L_logM_cut);
    pre.E_M1 = pow(10.0, p->E_logM1), pre.E_Mcut = pow(10.0, p->E_logM_cut);
    pre.E_M1_EL = pow(10.0, p->E_logM1_EL), pre.E_M1_EE = pow(10.0, p->E_logM1_EE);
    pre.Q_M1 = pow(10.0, p->Q_logM1), pre.Q_Mcut = pow(10.0, p->Q_logM_cut);
    return pre;
}

HodPtrs make_ptrs(const abacus_hod_state *st) {
    HodPtrs a;
    a.nh = st->nh, a.np = st->np, a.ntile_c = st->ntile_c, a.ntile_s = st->ntile_s, a.nsb_c = st->nsb_c, a.nsb_s = st->nsb_s;
    a.hmass = st->hmass, a.hmultis = st->hmultis, a.hrandoms = st->hrandoms, a.hdeltac = st->hdeltac,
    a.hfenv = st->hfenv, a.hshear = st->hshear;
    a.phmass = st->phmass, a.pweights = st->pweights, a.prandoms = st->prandoms, a.pdeltac = st->pdeltac,
    a.pfenv = st->pfenv, a.pshear = st->pshear, a.pranks = st->pranks, a.pranksv = st->pranksv,
    a.pranksp = st->pranksp, a.pranksr = st->pranksr, a.pinds = st->pinds;
    a.keep_c = st->keep_c, a.keep_s = st->keep_s, a.q_count = st->q_count, a.queue_c = st->queue_c,
    a.queue_s = st->queue_s, a.kept_c = st->kept_c, a.kept_s = st->kept_s, a.sb_counts = st->sb_counts;
    const bool rec = st->rec_ok && st->rec_rand_ok;   // stale randoms / hveldev in the records: gather from the columns
    a.hrec = rec ? st->hrec.as<HaloRec>() : nullptr;
    a.prec = rec ? st->prec.as<PartRec>() : nullptr;
    return a;
}

// superblock size of a populate: 16 tiles for LRG alone, 8 for the dense mixes (see SB_TILES_*)
void set_superblocks(abacus_hod_state *st, const abacus_hod_params *p) {
    int sbt = option("hod_sbtiles");
    if (sbt != SB_TILES_DENSE && sbt != SB_TILES_SPARSE) sbt = (p->want_ELG || p->want_QSO) ? SB_TILES_DENSE : SB_TILES_SPARSE;
    st->sb_tiles = sbt;
    // Dense mixes: hod_exact is one round of workgroups (three fit a CU at its ~150 registers), and a CU that holds three of
    // them is done a third later than one that holds two - 611 superblocks of 8 tiles at 1e7 objects left the 256 CUs with
    // 2.4 on average and 3 at most.  A whole number of workgroups per CU (768: 6 or 7 tiles each) evens that out.
    const int per_round = 3 * 256;
    auto count = [&](int ntile) {
        int n = (int)ceil_div(ntile, sbt);
        if (sbt == SB_TILES_DENSE && n > 256 && !option("hod_nobalance")) n = (int)ceil_div(n, per_round) * per_round;
        return std::min(n, std::max(ntile, 0));
    };
    st->nsb_c = count(st->ntile_c);
    st->nsb_s = count(st->ntile_s);
}

int launch_emit(abacus_hod_state *st) {
    const int nemit = st->nsb_c + st->nsb_s;
    if (nemit == 0) {
        HIP_TRY(hipMemsetAsync(st->d_totals, 0, 6 * sizeof(int64_t), stream()));
        return 0;
    }
    EmitPtrs in;
    in.hpos = st->hpos, in.hvel = st->hvel, in.hvdev = st->hveldev, in.hmass = st->hmass, in.hid = st->hid;
    in.ppos = st->ppos, in.pvel = st->pvel, in.phvel = st->phvel, in.phmass = st->phmass, in.phid = st->phid;
    const bool rec = st->rec_ok && st->rec_rand_ok;
    in.hrec = rec ? st->hrec.as<HaloRec>() : nullptr;
    in.prec = rec ? st->prec.as<PartRec>() : nullptr;
    // workgroup size: 384 threads per superblock for the dense mixes (ELG / QSO: a few thousand galaxies per superblock; at
    // its 86 registers three such workgroups fit a CU - all 768 central superblocks of 1e7 halos at once - where only two of
    // 512 threads do: 92 vs 101 us at LRG + ELG + QSO on 1e7 + 1e7, 95 with 256), 256 for LRG alone (12 vs 20 us: the larger
    // workgroups only cost launch time)
    int eb = option("hod_eblock");
    if (eb != 256 && eb != 384 && eb != 512) eb = (st->params.want_ELG || st->params.want_QSO) ? 384 : 256;
#define EMIT(EB, SBT)                                                                                                       \
    ABACUS_LAUNCH("hod_emit", (hod_emit<EB, SBT>), dim3(nemit), dim3(EB), 0, st->nsb_c, st->nsb_s, st->ntile_c, st->ntile_s, st->kept_c, st->kept_s, \
                  st->sb_counts, st->d_totals, in, st->params, out_cols(st), option("dbg"))
    const bool sparse = st->sb_tiles == SB_TILES_SPARSE;
    if (eb == 384 && sparse) EMIT(384, SB_TILES_SPARSE);
    else if (eb == 384) EMIT(384, SB_TILES_DENSE);
    else if (eb == 256 && sparse) EMIT(256, SB_TILES_SPARSE);
    else if (eb == 256) EMIT(256, SB_TILES_DENSE);
    else if (sparse) EMIT(512, SB_TILES_SPARSE);
    else EMIT(512, SB_TILES_DENSE);
#undef EMIT
    return 0;
}

// (re)build the packed filter keys of an owned catalogue (after staging, a reseed or an update of the randoms)
int build_keys(abacus_hod_state *st) {
    if (st->keys_ok) return 0;
    const int64_t ph = (int64_t)std::max(st->ntile_c, 1) * TILE, pp = (int64_t)std::max(st->ntile_s, 1) * TILE;
    ABACUS_TRY(st->keys.reserve((size_t)(ph + pp) * sizeof(unsigned short)));
    unsigned short *hk = st->keys.as<unsigned short>(), *pk = hk + ph;
    ABACUS_LAUNCH("hod_build_keys", hod_build_keys, dim3((unsigned)std::min<int64_t>(ceil_div(ph, 256), 8192)), dim3(256), 0,
                  (const double *)st->hmass, (const double *)st->hmultis, (const double *)st->hrandoms, st->nh, ph, hk);
    ABACUS_LAUNCH("hod_build_keys", hod_build_keys, dim3((unsigned)std::min<int64_t>(ceil_div(pp, 256), 8192)), dim3(256), 0,
                  (const double *)st->phmass, (const double *)st->pweights, (const double *)st->prandoms, st->np, pp, pk);
    st->keys_ok = true;
    st->index_ok = false;   // the index sorts these keys
    st->key_uses = 0;
    return 0;
}

// sort the objects of both kinds by (mass bin, q code) and tabulate, on the host, how many lie at or below every sort key
int build_index(abacus_hod_state *st) {
    if (st->index_ok) return 0;
    const int64_t nh = st->nh, np = st->np, nmax = std::max<int64_t>(std::max(nh, np), 1);
    if (nh > 0x7fffffffll || np > 0x7fffffffll) return 0;   // 32-bit indices / sort sizes: larger catalogues keep streaming the keys
    ABACUS_TRY(st->index_idx.reserve((size_t)std::max<int64_t>(nh + np, 1) * sizeof(unsigned int)));
    ABACUS_TRY(st->index_scratch.reserve((size_t)nmax * (2 + 2 + 4)));
    ABACUS_TRY(st->index_last.reserve(65536 * sizeof(unsigned int)));
    unsigned short *sk_in = st->index_scratch.as<unsigned short>(), *sk_out = sk_in + nmax;
    unsigned int *idx_in = reinterpret_cast<unsigned int *>(sk_out + nmax);
    unsigned int *last = st->index_last.as<unsigned int>();
    const unsigned short *hk = st->keys.as<unsigned short>(), *pk = hk + (int64_t)std::max(st->ntile_c, 1) * TILE;
    std::vector<unsigned int> host(65536);
    for (int kind = 0; kind < 2; kind++) {
        const int64_t n = kind ? np : nh;
        std::vector<unsigned int> &up = kind ? st->upper_p : st->upper_h;
        up.assign(65536, 0u);
        if (n == 0) continue;
        unsigned int *idx_out = st->index_idx.as<unsigned int>() + (kind ? nh : 0);
        const int grid = (int)std::min<int64_t>(ceil_div(n, 256), 8192);
        ABACUS_LAUNCH("hod_index_keys", hod_index_keys, dim3(grid), dim3(256), 0, kind ? pk : hk, n, sk_in, idx_in);
        ABACUS_TRY(sort_pairs_u16(sk_in, sk_out, idx_in, idx_out, n, st->index_tmp));
        HIP_TRY(hipMemsetAsync(last, 0, 65536 * sizeof(unsigned int), stream()));
        ABACUS_LAUNCH("hod_index_last", hod_index_last, dim3(grid), dim3(256), 0, (const unsigned short *)sk_out, n, last);
        HIP_TRY(hipMemcpyAsync(host.data(), last, 65536 * sizeof(unsigned int), hipMemcpyDeviceToHost, stream()));
        HIP_TRY(hipStreamSynchronize(stream()));
        unsigned int run = 0;
        for (int v = 0; v < 65536; v++) {
            if (host[v]) run = host[v];
            up[v] = run;
        }
    }
    st->index_ok = true;
    return 0;
}

// value ranges of the environment / rank columns (once per catalogue; absent columns keep the value the exact chain
// substitutes for them: 0, ranks 1)
int compute_ranges(abacus_hod_state *st) {
    if (st->ranges_ok) return 0;
    constexpr int NB = 1024;
    DevBuf part;
    ABACUS_TRY(part.reserve((size_t)NB * 2 * sizeof(double)));
    std::vector<double> hostv((size_t)NB * 2);
    double *host = hostv.data();
    auto col = [&](const double *src, int64_t n, double absent, ColRange &r) -> int {
        r.lo = r.hi = absent;
        if (!src || n <= 0) return 0;
        const int grid = (int)std::min<int64_t>(ceil_div(n, 256), NB);
        ABACUS_LAUNCH("hod_minmax", hod_minmax, dim3(grid), dim3(256), 0, src, n, part.as<double>());
        HIP_TRY(hipMemcpyAsync(host, part.p, (size_t)grid * 2 * sizeof(double), hipMemcpyDeviceToHost, stream()));
        HIP_TRY(hipStreamSynchronize(stream()));
        double lo = INFINITY, hi = -INFINITY;
        for (int b = 0; b < grid; b++) lo = std::fmin(lo, host[2 * b]), hi = std::fmax(hi, host[2 * b + 1]);
        if (lo <= hi) r.lo = lo, r.hi = hi;   // all-NaN column: the exact chain keeps nothing it weights; any range will do
        return 0;
    };
    HodRanges &R = st->ranges;
    int rc = 0;
    rc = rc ? rc : col(st->hdeltac, st->nh, 0.0, R.hdeltac);
    rc = rc ? rc : col(st->hfenv, st->nh, 0.0, R.hfenv);
    rc = rc ? rc : col(st->hshear, st->nh, 0.0, R.hshear);
    rc = rc ? rc : col(st->pdeltac, st->np, 0.0, R.pdeltac);
    rc = rc ? rc : col(st->pfenv, st->np, 0.0, R.pfenv);
    rc = rc ? rc : col(st->pshear, st->np, 0.0, R.pshear);
    rc = rc ? rc : col(st->pranks, st->np, 1.0, R.pranks[0]);
    rc = rc ? rc : col(st->pranksv, st->np, 1.0, R.pranks[1]);
    rc = rc ? rc : col(st->pranksp, st->np, 1.0, R.pranks[2]);
    rc = rc ? rc : col(st->pranksr, st->np, 1.0, R.pranks[3]);
    (void)part.release();
    if (rc) return rc;
    st->ranges_ok = true;
    return 0;
}

// (re)build the float32 shadow columns of an owned catalogue; `rand_only`: just the two random columns (after a reseed
// or an update of the randoms)
int build_shadows(abacus_hod_state *st, bool rand_only) {
    const int64_t nh = st->nh, np = st->np;
    const int64_t ph = (nh + 3) / 4 * 4 + 4, pp = (np + 3) / 4 * 4 + 4;
    if (!rand_only) {
        const int ncol_h = 3 + (st->hdeltac ? 1 : 0) + (st->hfenv ? 1 : 0) + (st->hshear ? 1 : 0);
        const int ncol_p = 3 + (st->pranks ? 1 : 0) + (st->pranksv ? 1 : 0) + (st->pranksp ? 1 : 0) + (st->pranksr ? 1 : 0);
        ABACUS_TRY(st->shadow.reserve(((size_t)ncol_h * ph + (size_t)ncol_p * pp) * sizeof(float)));
    }
    float *cur = st->shadow.as<float>();
    auto col = [&](const double *src, int64_t n, int64_t npad, int mode, float fill, const float **slot, bool is_rand) -> int {
        if (!src) {
            *slot = nullptr;
            return 0;
        }
        float *dst = cur;
        cur += npad;
        *slot = dst;
        if (rand_only && !is_rand) return 0;
        const int grid = (int)std::min<int64_t>(std::max<int64_t>(ceil_div(npad, 256), 1), 8192);
        if (mode == 0) ABACUS_LAUNCH("hod_shadow", hod_shadow<0>, dim3(grid), dim3(256), 0, src, dst, n, npad, fill);
        else if (mode == 1) ABACUS_LAUNCH("hod_shadow", hod_shadow<1>, dim3(grid), dim3(256), 0, src, dst, n, npad, fill);
        else ABACUS_LAUNCH("hod_shadow", hod_shadow<2>, dim3(grid), dim3(256), 0, src, dst, n, npad, fill);
        return 0;
    };
    FiltCols &c = st->fc;
    // padding values: randoms 2 (> any bound) reject; the objects behind n are masked in the kernel anyway
    ABACUS_TRY(col(st->hmass, nh, ph, 1, 1.f, &c.hmass, false));
    ABACUS_TRY(col(st->hmultis, nh, ph, 0, 0.f, &c.hmultis, false));
    ABACUS_TRY(col(st->hrandoms, nh, ph, 2, 2.f, &c.hrandoms, true));
    ABACUS_TRY(col(st->hdeltac, nh, ph, 0, 0.f, &c.hdeltac, false));
    ABACUS_TRY(col(st->hfenv, nh, ph, 0, 0.f, &c.hfenv, false));
    ABACUS_TRY(col(st->hshear, nh, ph, 0, 0.f, &c.hshear, false));
    ABACUS_TRY(col(st->phmass, np, pp, 1, 1.f, &c.phmass, false));
    ABACUS_TRY(col(st->pweights, np, pp, 0, 0.f, &c.pweights, false));
    ABACUS_TRY(col(st->prandoms, np, pp, 2, 2.f, &c.prandoms, true));
    ABACUS_TRY(col(st->pranks, np, pp, 0, 1.f, &c.pranks, false));
    ABACUS_TRY(col(st->pranksv, np, pp, 0, 1.f, &c.pranksv, false));
    ABACUS_TRY(col(st->pranksp, np, pp, 0, 1.f, &c.pranksp, false));
    ABACUS_TRY(col(st->pranksr, np, pp, 0, 1.f, &c.pranksr, false));
    st->shadow_ok = st->shadow_rand_ok = true;
    return 0;
}

}  // namespace

__global__ void hod_check_pinds(const int64_t *__restrict__ pinds, int64_t np, int64_t nh, int *__restrict__ flag) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < np; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t h = pinds[i];
        bad = bad || h < 0 || h >= nh;
    }
    if (bad) *flag = 1;
}

// uploads / adopts the arrays and allocates the work buffers of a freshly constructed state; on failure the caller
// frees whatever was allocated so far (abacus_hod_free is null-safe for every member)
static int stage_fill(abacus_hod_state *st, const abacus_hod_arrays *a, int on_device) {
    const bool d = on_device != 0;
    const int64_t nh = st->nh, np = st->np;
#define UP(field, n) ABACUS_TRY(upload(st->field, a->field, n, d))
    UP(hpos, 3 * nh); UP(hvel, 3 * nh); UP(hmass, nh); UP(hid, nh); UP(hmultis, nh); UP(hrandoms, nh);
    UP(hveldev, 3 * nh); UP(hdeltac, nh); UP(hfenv, nh); UP(hshear, nh);
    UP(ppos, 3 * np); UP(pvel, 3 * np); UP(phvel, 3 * np); UP(phmass, np); UP(phid, np); UP(pweights, np);
    UP(prandoms, np); UP(pdeltac, np); UP(pfenv, np); UP(pshear, np); UP(pranks, np); UP(pranksv, np);
    UP(pranksp, np); UP(pranksr, np); UP(pinds, np);
#undef UP
    st->ntile_c = (int)ceil_div(nh, TILE);
    st->ntile_s = (int)ceil_div(np, TILE);
    const int64_t ntiles = (int64_t)st->ntile_c + st->ntile_s;
    HIP_TRY(hipMalloc((void **)&st->keep_c, nh > 0 ? nh + 64 : 64));
    HIP_TRY(hipMalloc((void **)&st->keep_s, np > 0 ? np + 64 : 64));
    // counters and kept lists sized for either superblock size (the populate picks one, set_superblocks)
    st->nsb_c = (int)ceil_div(st->ntile_c, SB_TILES_MIN);
    st->nsb_s = (int)ceil_div(st->ntile_s, SB_TILES_MIN);
    // (set_superblocks rounds the counts of a populate up to a whole number of workgroups per CU: at most 768 more each)
    HIP_TRY(hipMalloc((void **)&st->sb_counts, (size_t)(st->nsb_c + st->nsb_s + 2 * 768 + 1) * 4 * sizeof(int)));
    HIP_TRY(hipMalloc((void **)&st->d_totals, 8 * sizeof(int64_t)));
    HIP_TRY(hipMalloc((void **)&st->q_count, (size_t)(ntiles > 0 ? ntiles : 1) * sizeof(int)));
    HIP_TRY(hipMalloc((void **)&st->queue_c, (size_t)(st->ntile_c > 0 ? st->ntile_c : 1) * TILE * sizeof(unsigned short)));
    HIP_TRY(hipMalloc((void **)&st->queue_s, (size_t)(st->ntile_s > 0 ? st->ntile_s : 1) * TILE * sizeof(unsigned short)));
    HIP_TRY(hipMalloc((void **)&st->kept_c, (size_t)(st->ntile_c + SB_TILES_MAX) * TILE * sizeof(unsigned short)));
    HIP_TRY(hipMalloc((void **)&st->kept_s, (size_t)(st->ntile_s + SB_TILES_MAX) * TILE * sizeof(unsigned short)));
    HIP_TRY(hipHostMalloc((void **)&st->h_totals, 8 * sizeof(int64_t), hipHostMallocDefault));
    // first guess for the catalog buffers; grown on demand by abacus_hod_counts
    for (int t = 0; t < 3; t++) ABACUS_TRY(set_capacity(st, t, (nh + np) / 64));
    if (st->pinds && st->np > 0) {   // keep_c[pinds[i]] is read unchecked by the kernels: reject a stale / out-of-range index here
        int *flag = (int *)st->d_totals;   // scratch, rewritten by every populate
        HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int), stream()));
        ABACUS_LAUNCH("hod_check_pinds", hod_check_pinds, dim3((unsigned)std::min<int64_t>(ceil_div(st->np, 256), 4096)), dim3(256), 0,
                      st->pinds, st->np, st->nh, flag);
        int bad = 0;
        HIP_TRY(hipMemcpyAsync(&bad, flag, sizeof(int), hipMemcpyDeviceToHost, stream()));
        HIP_TRY(hipStreamSynchronize(stream()));
        if (bad) return fail("abacus_hod_stage: pinds holds a host index outside [0, %lld) (stale after sub-selecting the halos?)", (long long)st->nh);
    }
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

extern "C" {

int abacus_hod_stage(const abacus_hod_arrays *a, int on_device, abacus_hod_state **out) {
    ABACUS_ENTER();
    if (!a || !out) return fail("abacus_hod_stage: null argument");
    if (a->n_halo < 0 || a->n_part < 0) return fail("abacus_hod_stage: negative length");
    if (a->n_halo > 0 && (!a->hpos || !a->hvel || !a->hmass || !a->hid || !a->hmultis || !a->hrandoms || !a->hveldev))
        return fail("abacus_hod_stage: a required halo array is NULL");
    if (a->n_part > 0 && (!a->ppos || !a->pvel || !a->phvel || !a->phmass || !a->phid || !a->pweights || !a->prandoms))
        return fail("abacus_hod_stage: a required particle array is NULL");
    if (a->n_halo >= ((int64_t)1 << 32) || a->n_part >= ((int64_t)1 << 32))
        return fail("abacus_hod_stage: too many objects for one device");
    auto *st = new abacus_hod_state();
    st->nh = a->n_halo;
    st->np = a->n_part;
    st->owns = !on_device;
    const int rc = stage_fill(st, a, on_device);
    if (rc != 0) {   // e.g. out of HBM half-way through: nothing of the partial state may leak (a retry with a smaller chunk must fit)
        const std::string msg = abacus_last_error();
        (void)abacus_hod_free(st);
        return fail("%s", msg.c_str());
    }
    *out = st;
    return 0;
}

int abacus_hod_update(abacus_hod_state *st, const char *field, const double *host) {
    ABACUS_ENTER();
    if (!st || !field || !host) return fail("abacus_hod_update: null argument");
    double *dst = nullptr;
    int64_t n = 0;
    if (!strcmp(field, "hrandoms")) dst = st->hrandoms, n = st->nh;
    else if (!strcmp(field, "hveldev")) dst = st->hveldev, n = 3 * st->nh;
    else if (!strcmp(field, "prandoms")) dst = st->prandoms, n = st->np;
    else return fail("abacus_hod_update: unknown field '%s'", field);
    if (n == 0) return 0;
    HIP_TRY(hipMemcpyAsync(dst, host, n * sizeof(double), hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    st->shadow_rand_ok = false;   // the float32 shadows of the randoms are rebuilt by the next populate
    st->keys_ok = false;          // ... and the packed filter keys
    st->rec_rand_ok = false;      // ... and the random / hveldev fields of the packed records
    return 0;
}

int abacus_hod_set_sigma3d(abacus_hod_state *st, const double *sigma3d, int on_device) {
    ABACUS_ENTER();
    if (!st || !sigma3d) return fail("abacus_hod_set_sigma3d: null argument");
    if (st->owns_sigma && st->hsigma3d) HIP_TRY(hipFree(st->hsigma3d));
    st->hsigma3d = nullptr;
    st->owns_sigma = false;
    if (on_device) {
        st->hsigma3d = const_cast<double *>(sigma3d);
        return 0;
    }
    HIP_TRY(hipMalloc((void **)&st->hsigma3d, (size_t)(st->nh > 0 ? st->nh : 1) * sizeof(double)));
    st->owns_sigma = true;
    HIP_TRY(hipMemcpyAsync(st->hsigma3d, sigma3d, (size_t)st->nh * sizeof(double), hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_hod_reseed(abacus_hod_state *st, uint64_t seed, int want_expvel, int64_t halo_index0, int64_t part_index0) {
    ABACUS_ENTER();
    if (!st) return fail("abacus_hod_reseed: null handle");
    if (!st->owns) return fail("abacus_hod_reseed: the catalogue was staged from caller-owned device arrays (read-only)");
    if (st->nh > 0 && !st->hsigma3d) return fail("abacus_hod_reseed: hsigma3d has not been set (abacus_hod_set_sigma3d)");
    if (halo_index0 < 0 || part_index0 < 0) return fail("abacus_hod_reseed: negative index offset");
    if (st->nh > 0) {
        const int grid = (int)std::min<int64_t>(ceil_div(st->nh, 256), 256 * 32);
        ABACUS_LAUNCH("hod_reseed_halos", hod_reseed_halos, dim3(grid), dim3(256), 0, st->nh, halo_index0,
                      (unsigned long long)seed, st->hsigma3d, want_expvel, st->hrandoms, st->hveldev);
    }
    if (st->np > 0) {
        const int grid = (int)std::min<int64_t>(ceil_div(ceil_div(st->np, 4) + 1, 256), 256 * 32);
        ABACUS_LAUNCH("hod_reseed_particles", hod_reseed_particles, dim3(grid), dim3(256), 0, st->np, part_index0,
                      (unsigned long long)seed, st->prandoms);
    }
    st->shadow_rand_ok = false;
    st->keys_ok = false;
    st->rec_rand_ok = false;
    return 0;
}

int abacus_hod_fetch_field(abacus_hod_state *st, const char *field, double *host) {
    ABACUS_ENTER();
    if (!st || !field || !host) return fail("abacus_hod_fetch_field: null argument");
    const double *src = nullptr;
    int64_t n = 0;
    if (!strcmp(field, "hrandoms")) src = st->hrandoms, n = st->nh;
    else if (!strcmp(field, "hveldev")) src = st->hveldev, n = 3 * st->nh;
    else if (!strcmp(field, "prandoms")) src = st->prandoms, n = st->np;
    else return fail("abacus_hod_fetch_field: unknown field '%s'", field);
    if (n == 0) return 0;
    HIP_TRY(hipMemcpyAsync(host, src, n * sizeof(double), hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_hod_set_profile(abacus_hod_state *st, const double *hc, const double *hrvir) {
    ABACUS_ENTER();
    if (!st || !hc || !hrvir) return fail("abacus_hod_set_profile: null argument");
    const size_t bytes = (size_t)(st->nh > 0 ? st->nh : 1) * sizeof(double);
    if (!st->hc) HIP_TRY(hipMalloc((void **)&st->hc, bytes));
    if (!st->hrvir) HIP_TRY(hipMalloc((void **)&st->hrvir, bytes));
    HIP_TRY(hipMemcpyAsync(st->hc, hc, (size_t)st->nh * sizeof(double), hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipMemcpyAsync(st->hrvir, hrvir, (size_t)st->nh * sizeof(double), hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_hod_populate_nfw(abacus_hod_state *st, const abacus_hod_params *p, const abacus_nfw_params *nfw,
                            const double *NFW_draw, int64_t n_draw, int64_t counts[6]) {
    ABACUS_ENTER();
    if (!st || !p || !nfw) return fail("abacus_hod_populate_nfw: null argument");
    if (p->has_origin) return fail("abacus_hod_populate_nfw: the NFW path does not support light cones (hod/GRAND_HOD.py:551)");
    if (st->nh > 0 && (!st->hsigma3d || !st->hc || !st->hrvir))
        return fail("abacus_hod_populate_nfw: hsigma3d / hc / hrvir have not been staged (abacus_hod_set_sigma3d, _set_profile)");
    if (!NFW_draw || n_draw < 1) return fail("abacus_hod_populate_nfw: NFW_draw is empty");
    st->params = *p;
    SatPre pre;
    memset(&pre, 0, sizeof pre);
    Filt F = make_filter(*p, pre);
    HodPtrs a;
    memset(&a, 0, sizeof a);
    set_superblocks(st, p);
    st->kept_valid = false;   // the particles' masks and kept lists are left as they are: no lazy masks after this path
    st->q_zero = false, st->last_cand[0] = st->last_cand[1] = -1;
    a.nh = st->nh, a.np = 0, a.ntile_c = st->ntile_c, a.ntile_s = 0, a.nsb_c = st->nsb_c, a.nsb_s = 0;
    a.hmass = st->hmass, a.hmultis = st->hmultis, a.hrandoms = st->hrandoms, a.hdeltac = st->hdeltac,
    a.hfenv = st->hfenv, a.hshear = st->hshear;
    a.keep_c = st->keep_c, a.keep_s = st->keep_s, a.q_count = st->q_count, a.queue_c = st->queue_c,
    a.queue_s = st->queue_s, a.kept_c = st->kept_c, a.kept_s = st->kept_s, a.sb_counts = st->sb_counts;
    const int need_env = (p->want_LRG && (p->L_Acent != 0 || p->L_Bcent != 0)) ||
                         (p->want_ELG && (p->E_Acent != 0 || p->E_Bcent != 0)) ||
                         (p->want_QSO && (p->Q_Acent != 0 || p->Q_Bcent != 0));
    const int need_shear = p->want_ELG && p->E_Ccent != 0 && st->hshear != nullptr;
    // centrals exactly as the particle path decides them; no particle satellites
    if (st->nsb_s) HIP_TRY(hipMemsetAsync(st->sb_counts + (int64_t)st->nsb_c * 4, 0, (size_t)st->nsb_s * 4 * sizeof(int), stream()));
    if (st->ntile_c) {
        ABACUS_LAUNCH("hod_filter", hod_filter, dim3(st->ntile_c), dim3(FBLOCK), 0, a, 0, p->want_LRG, p->want_ELG,
                      p->want_QSO, p->enable_ranks, need_env, need_shear, F);
        abacus_cls::ClsConst cc;
        abacus_cls::make_cls_const(*p, pre, cc);
        if (st->sb_tiles == SB_TILES_SPARSE)
            ABACUS_LAUNCH("hod_exact", (hod_exact_plain<256, SB_TILES_SPARSE>), dim3(st->nsb_c), dim3(256), 0, a, 0, *p, pre, cc, 1, 0);
        else
            ABACUS_LAUNCH("hod_exact", (hod_exact_plain<256, SB_TILES_DENSE>), dim3(st->nsb_c), dim3(256), 0, a, 0, *p, pre, cc, 1, 0);
    }
    ABACUS_TRY(launch_emit(st));
    HIP_TRY(hipMemcpyAsync(st->h_totals, st->d_totals, 6 * sizeof(int64_t), hipMemcpyDeviceToHost, stream()));
    // Poisson satellite numbers per halo and tracer, their exclusive offsets
    const int64_t nh = st->nh;
    ABACUS_TRY(st->nfw_counts.reserve((size_t)(3 * (nh + 1)) * sizeof(unsigned int)));
    ABACUS_TRY(st->nfw_offsets.reserve((size_t)(3 * (nh + 1)) * sizeof(int64_t)));
    ABACUS_TRY(st->nfw_draw.reserve((size_t)n_draw * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(st->nfw_draw.p, NFW_draw, (size_t)n_draw * sizeof(double), hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipMemsetAsync(st->nfw_counts.p, 0, (size_t)(3 * (nh + 1)) * sizeof(unsigned int), stream()));
    unsigned int *cnt[3];
    int64_t *off[3];
    for (int t = 0; t < 3; t++) {
        cnt[t] = st->nfw_counts.as<unsigned int>() + (int64_t)t * (nh + 1);
        off[t] = st->nfw_offsets.as<int64_t>() + (int64_t)t * (nh + 1);
    }
    NfwArgs nf;
    nf.seed = nfw->seed;
    for (int t = 0; t < 3; t++) nf.f_sigv[t] = nfw->f_sigv[t];
    nf.exp_frac = nfw->exp_frac, nf.exp_scale = nfw->exp_scale, nf.nfw_rescale = nfw->nfw_rescale;
    nf.halo_index0 = nfw->halo_index0;
    nf.n_draw = n_draw;
    int64_t nsat[3] = {0, 0, 0};
    if (nh > 0) {
        const int grid = (int)std::min<int64_t>(ceil_div(nh, 256), 256 * 32);
        ABACUS_LAUNCH("hod_nfw_count", hod_nfw_count, dim3(grid), dim3(256), 0, a, *p, nf, cnt[0], cnt[1], cnt[2]);
        for (int t = 0; t < 3; t++) {
            ABACUS_TRY(exclusive_scan_u32(cnt[t], nh, off[t], st->nfw_scan, 0));
            HIP_TRY(hipMemcpyAsync(&nsat[t], off[t] + nh, sizeof(int64_t), hipMemcpyDeviceToHost, stream()));
        }
    }
    HIP_TRY(hipStreamSynchronize(stream()));
    bool grew = false;
    for (int t = 0; t < 3; t++) {
        st->counts[t] = st->h_totals[t];
        st->counts[3 + t] = nsat[t];
        const int64_t need = st->counts[t] + st->counts[3 + t];
        if (need > st->cap[t]) {
            ABACUS_TRY(set_capacity(st, t, need + need / 8));
            grew = true;
        }
    }
    if (grew) ABACUS_TRY(launch_emit(st));   // the centrals again, into the new buffers
    for (int t = 0; t < 3; t++) {
        if (nsat[t] == 0) continue;
        const int grid = (int)std::min<int64_t>(ceil_div(nsat[t], 256), 256 * 32);
        ABACUS_LAUNCH("hod_nfw_emit", hod_nfw_emit, dim3(grid), dim3(256), 0, t, nsat[t], nh, (const int64_t *)off[t],
                      st->counts[t], st->hpos, st->hvel, st->hmass, st->hid, st->hsigma3d, st->hc, st->hrvir,
                      st->nfw_draw.as<double>(), *p, nf, out_cols(st));
    }
    HIP_TRY(hipStreamSynchronize(stream()));
    st->have_run = true;
    st->counts_valid = true;
    if (counts) memcpy(counts, st->counts, sizeof st->counts);
    return 0;
}

int abacus_hod_set_ngal_bins(abacus_hod_state *st, const uint8_t *bins4, const double *centres, int nbin) {
    ABACUS_ENTER();
    if (!st || !bins4 || !centres) return fail("abacus_hod_set_ngal_bins: null argument");
    if (nbin < 1 || nbin > 254) return fail("abacus_hod_set_ngal_bins: nbin must be in [1, 254]");
    ABACUS_TRY(st->ngal_bins.reserve((size_t)(st->nh > 0 ? st->nh : 1) * 4));
    ABACUS_TRY(st->ngal_tables.reserve((size_t)4 * nbin * sizeof(double)));
    ABACUS_TRY(st->ngal_out.reserve(8 * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(st->ngal_bins.p, bins4, (size_t)st->nh * 4, hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipMemcpyAsync(st->ngal_tables.p, centres, (size_t)4 * nbin * sizeof(double), hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    st->ngal_nbin = nbin;
    return 0;
}

int abacus_hod_ngal(abacus_hod_state *st, const abacus_hod_params *p, double out[6]) {
    ABACUS_ENTER();
    if (!st || !p || !out) return fail("abacus_hod_ngal: null argument");
    if (!st->ngal_nbin) return fail("abacus_hod_ngal: the histogram cells have not been set (abacus_hod_set_ngal_bins)");
    HIP_TRY(hipMemsetAsync(st->ngal_out.p, 0, 6 * sizeof(double), stream()));
    if (st->nh > 0) {
        NgalTables tb;
        const double *c = st->ngal_tables.as<double>();
        tb.Mh = c, tb.deltac = c + st->ngal_nbin, tb.fenv = c + 2 * st->ngal_nbin, tb.shear = c + 3 * st->ngal_nbin;
        const int grid = (int)std::min<int64_t>(ceil_div(st->nh, 256), 256 * 8);
        ABACUS_LAUNCH("hod_ngal", hod_ngal, dim3(grid), dim3(256), 0, st->nh, (const uchar4 *)st->ngal_bins.p,
                      (const double *)st->hmultis, tb, *p, st->ngal_out.as<double>());
    }
    HIP_TRY(hipMemcpyAsync(out, st->ngal_out.p, 6 * sizeof(double), hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_hod_populate_async(abacus_hod_state *st, const abacus_hod_params *p) {
    ABACUS_ENTER();
    if (!st || !p) return fail("abacus_hod_populate: null argument");
    if (p->enable_ranks && st->np > 0 && (!st->pranks || !st->pranksv || !st->pranksp || !st->pranksr))
        return fail("abacus_hod_populate: enable_ranks set but the rank arrays were not staged");
    if (p->want_ELG && st->np > 0 && !st->pinds)
        return fail("abacus_hod_populate: ELG conformity needs pinds to be staged");
    st->params = *p;
    const SatPre pre = make_pre(p);
    Filt F = make_filter(*p, pre);
    // deltac / fenv / shear are streamed by the central filter only when some wanted tracer weights them
    const int need_env = (p->want_LRG && (p->L_Acent != 0 || p->L_Bcent != 0)) ||
                         (p->want_ELG && (p->E_Acent != 0 || p->E_Bcent != 0)) ||
                         (p->want_QSO && (p->Q_Acent != 0 || p->Q_Bcent != 0));
    const int need_shear = p->want_ELG && p->E_Ccent != 0 && st->hshear != nullptr;
    ABACUS_TRY(build_records(st));
    set_superblocks(st, p);
    const HodPtrs a = make_ptrs(st);
    const bool conf = p->want_ELG && st->pinds != nullptr && st->ntile_s > 0;   // satellites read keep_cent[pinds]
    const int ntile = st->ntile_c + st->ntile_s, nsb = st->nsb_c + st->nsb_s;
    // owned catalogues: the filter streams the float32 shadow columns (half the bytes); caller-owned device arrays can
    // change behind the library's back, so they are streamed as they are
    const bool force64 = option("hod_f64filter") != 0;
    const bool use32 = st->owns && !force64;
    if (use32 && !st->shadow_ok) ABACUS_TRY(build_shadows(st, false));
    else if (use32 && !st->shadow_rand_ok) ABACUS_TRY(build_shadows(st, true));
    const FiltCols fc = st->fc;
    if (use32) ABACUS_TRY(compute_ranges(st));
    if (use32) ABACUS_TRY(build_keys(st));
    const Cheap cheap = use32 ? make_cheap(*p, F, st->ranges) : Cheap{};
    const KeyTab keytab = use32 ? make_keytab(cheap) : KeyTab{};
    // lazy keep masks (see hod_exact): only when one key-filter launch covers both kinds, the mix is sparse, and the kept
    // lists of the previous populate describe the masks
    const bool filter_first_ = conf && use32 && cheap.s_ok;
    const bool lazy_masks = use32 && !option("hod_nokeys") && !option("hod_nolazy") && cheap.c_ok && cheap.s_ok && st->ntile_c > 0 &&
                            st->ntile_s > 0 && (!conf || filter_first_) && st->sb_tiles == SB_TILES_SPARSE && st->kept_valid &&
                            st->kept_sb_tiles == st->sb_tiles;
    st->kept_valid = false;   // until this populate's launches are all enqueued
    // mass-sorted key index (see hod_deal): from the second populate on the same keys, for the mixes that run lazy masks
    bool index_mode = false;
    DealTab deal;
    if (use32 && st->sb_tiles == SB_TILES_SPARSE && !option("hod_noindex") && !option("hod_nokeys")) {
        st->key_uses++;
        if (!st->index_ok && st->key_uses >= 2) ABACUS_TRY(build_index(st));
        if (st->index_ok && lazy_masks) {
            unsigned int pre = 0;
            deal.nseg = 0;
            int64_t cand[2] = {0, 0};
            for (int kind = 0; kind < 2; kind++) {
                const std::vector<unsigned int> &up = kind ? st->upper_p : st->upper_h;
                const unsigned short *tcv = kind ? keytab.s : keytab.c;
                for (int b = 0; b < 128; b++) {
                    const unsigned int lo = b == 0 ? 0u : up[(b << 9) - 1], hi = up[(b << 9) | std::min<int>(tcv[b], 511)];
                    if (hi > lo) {
                        deal.start[deal.nseg] = lo, deal.pre[deal.nseg] = pre;
                        pre += hi - lo, cand[kind] += hi - lo;
                        deal.nseg++;
                    }
                }
                if (kind == 0) deal.nseg_c = deal.nseg;
            }
            deal.pre[deal.nseg] = pre;
            if ((int64_t)pre <= (st->nh + st->np) / 8) {   // a dense threshold set: stream the keys instead
                index_mode = true;
                st->last_cand[0] = cand[0], st->last_cand[1] = cand[1];
            }
        }
    }
    if (!index_mode) st->last_cand[0] = st->last_cand[1] = -1;
    const int exact_flags = (lazy_masks ? 1 : 0) | (index_mode ? 2 : 0);
    // `first`, `count` in global tile ids (centrals first): the shadow path launches the two kinds separately
    auto filter32 = [&](int first, int count) -> int {
        const int c0 = std::min(first, st->ntile_c), c1 = std::min(first + count, st->ntile_c);
        const int s0 = std::max(first, st->ntile_c) - st->ntile_c, s1 = std::max(first + count, st->ntile_c) - st->ntile_c;
#define F32(KIND, TWO, first_, count_)                                                                                \
    ABACUS_LAUNCH("hod_filter", (hod_filter32<KIND, TWO>), dim3(count_), dim3(FBLOCK), 0, a, fc, first_, p->want_LRG, \
                  p->want_ELG, p->want_QSO, p->enable_ranks, need_env, need_shear, F, cheap)
        const bool c2 = cheap.c_ok != 0, s2 = cheap.s_ok != 0;
        // two-stage kinds stream the packed keys (4 B per object), whole tile groups: only full-kind ranges take this path
        const bool keyed = !option("hod_nokeys");
        const unsigned short *hk = st->keys.as<unsigned short>(), *pk = hk + (int64_t)std::max(st->ntile_c, 1) * TILE;
        const bool kc = keyed && c2 && c0 == 0 && c1 == st->ntile_c && c1 > c0, ks = keyed && s2 && s0 == 0 && s1 == st->ntile_s && s1 > s0;
        const int gc = (int)ceil_div(st->ntile_c, KEY_TILES), gs = (int)ceil_div(st->ntile_s, KEY_TILES);
#define FKEY(KIND, grid_) ABACUS_LAUNCH("hod_filter", (hod_filter_key<KIND>), dim3(grid_), dim3(FBLOCK), 0, a, hk, pk, gc, keytab, lazy_masks ? 1 : 0)
        if (kc && ks) {
            FKEY(2, gc + gs);
            return 0;
        }
        if (kc) FKEY(0, gc);
        if (ks) FKEY(1, gs);
#undef FKEY
        if (kc) {
            if (s1 > s0 && !ks) {
                if (s2) F32(1, true, s0, s1 - s0);
                else F32(1, false, s0, s1 - s0);
            }
            return 0;
        }
        if (ks) {
            if (c1 > c0) {
                if (c2) F32(0, true, c0, c1 - c0);
                else F32(0, false, c0, c1 - c0);
            }
            return 0;
        }
        if (c1 > c0 && s1 > s0 && c2 == s2) {   // both kinds, same path: one launch
            if (c2) F32(2, true, first, count);
            else F32(2, false, first, count);
            return 0;
        }
        if (c1 > c0) {
            if (c2) F32(0, true, c0, c1 - c0);
            else F32(0, false, c0, c1 - c0);
        }
        if (s1 > s0) {
            if (s2) F32(1, true, s0, s1 - s0);
            else F32(1, false, s0, s1 - s0);
        }
#undef F32
        return 0;
    };
#define FILTER(first, count)                                                                                         \
    if ((count) > 0) {                                                                                               \
        if (index_mode) {                                                                                            \
            if (!st->q_zero) HIP_TRY(hipMemsetAsync(st->q_count, 0, (size_t)ntile * sizeof(int), stream()));         \
            st->q_zero = false;   /* until hod_exact (which zeroes them again) is enqueued as well */               \
            if (deal.pre[deal.nseg] > 0)                                                                             \
                ABACUS_LAUNCH("hod_deal", hod_deal, dim3((deal.pre[deal.nseg] + 255u) / 256u), dim3(256), 0, a,       \
                              (const unsigned int *)st->index_idx.as<unsigned int>(),                                \
                              (const unsigned int *)(st->index_idx.as<unsigned int>() + st->nh), deal);             \
        } else if (use32) ABACUS_TRY(filter32(first, count));                                                               \
        else                                                                                                         \
            ABACUS_LAUNCH("hod_filter", hod_filter, dim3(count), dim3(FBLOCK), 0, a, first, p->want_LRG, p->want_ELG, \
                          p->want_QSO, p->enable_ranks, need_env, need_shear, F);                                    \
    }
    const bool nocls = option("hod_nocls") != 0;   // A/B: every candidate through the float64 chain
    abacus_cls::ClsConst cc;
    abacus_cls::make_cls_const(*p, pre, cc);
    // software-pipelined candidate loop for the dense mixes (see hod_exact); `hod_pipe` = 1 / 2 forces it off / on (A/B)
    const int pipe_opt = option("hod_pipe");
    const bool pipe = a.hrec && a.prec && (pipe_opt == 2 || (pipe_opt != 1 && (p->want_ELG || p->want_QSO)));
    const bool sparse_sb = st->sb_tiles == SB_TILES_SPARSE;
#define EXACT_(PIPE, SBT, first, count)                                                                                      \
    do {                                                                                                                     \
        if (PIPE)                                                                                                            \
            ABACUS_LAUNCH("hod_exact", (hod_exact<256, true, SBT>), dim3(count), dim3(256), 0, a, first, *p, pre, cc,        \
                          nocls ? 0 : 1, exact_flags);                                                                       \
        else                                                                                                                 \
            ABACUS_LAUNCH("hod_exact", (hod_exact_plain<256, SBT>), dim3(count), dim3(256), 0, a, first, *p, pre, cc,        \
                          nocls ? 0 : 1, exact_flags);                                                                       \
    } while (0)
#define EXACT(first, count)                                                    \
    if ((count) > 0) {                                                         \
        if (pipe && sparse_sb) EXACT_(true, SB_TILES_SPARSE, first, count);    \
        else if (pipe) EXACT_(true, SB_TILES_DENSE, first, count);             \
        else if (sparse_sb) EXACT_(false, SB_TILES_SPARSE, first, count);      \
        else EXACT_(false, SB_TILES_DENSE, first, count);                      \
    }
    // the two-stage satellite filter bounds the conformity variants by their largest, so it does not wait for the exact
    // central decisions: one filter launch for both kinds, then the exact passes in order
    const bool filter_first = conf && use32 && cheap.s_ok;
    if (!conf) {
        FILTER(0, ntile)
        EXACT(0, nsb);
    } else if (filter_first) {
        FILTER(0, ntile)
        EXACT(0, st->nsb_c);
        EXACT(st->nsb_c, st->nsb_s);
    } else {
        FILTER(0, st->ntile_c)
        EXACT(0, st->nsb_c);
        FILTER(st->ntile_c, st->ntile_s)
        EXACT(st->nsb_c, st->nsb_s);
    }
#undef FILTER
#undef EXACT
#undef EXACT_
    // speculative emission into the current buffers (writes past capacity are suppressed on the device)
    ABACUS_TRY(launch_emit(st));
    HIP_TRY(hipMemcpyAsync(st->h_totals, st->d_totals, 6 * sizeof(int64_t), hipMemcpyDeviceToHost, stream()));
    st->have_run = true;
    st->counts_valid = false;
    st->kept_valid = true, st->kept_sb_tiles = st->sb_tiles;   // every mask byte that is set is in a kept list
    st->q_zero = index_mode;   // index path: hod_exact zeroed the counters it read; streaming filter: its counts stay in q_count
    return 0;
}

int abacus_hod_counts(abacus_hod_state *st, int64_t counts[6]) {
    ABACUS_ENTER();
    if (!st || !st->have_run) return fail("abacus_hod_counts: populate has not been called");
    if (!st->counts_valid) {
        HIP_TRY(hipStreamSynchronize(stream()));
        bool grew = false;
        for (int t = 0; t < 3; t++) {
            st->counts[t] = st->h_totals[t];
            st->counts[3 + t] = st->h_totals[3 + t];
            int64_t need = st->counts[t] + st->counts[3 + t];
            if (need > st->cap[t]) {
                ABACUS_TRY(set_capacity(st, t, need + need / 8));
                grew = true;
            }
        }
        if (grew) {  // keep masks and offsets are still valid: only the emission is repeated
            ABACUS_TRY(launch_emit(st));
            HIP_TRY(hipStreamSynchronize(stream()));
        }
        st->counts_valid = true;
    }
    if (counts) memcpy(counts, st->counts, sizeof st->counts);
    return 0;
}

int abacus_hod_candidates(abacus_hod_state *st, int64_t out[2]) {
    ABACUS_ENTER();
    if (!st || !st->have_run || !out) return fail("abacus_hod_candidates: populate has not been called");
    if (st->last_cand[0] >= 0) {   // index path: the host dealt them out (the counters are zero again)
        out[0] = st->last_cand[0], out[1] = st->last_cand[1];
        return 0;
    }
    const int nt = st->ntile_c + st->ntile_s;
    std::vector<int> q((size_t)std::max(nt, 1));
    HIP_TRY(hipMemcpyAsync(q.data(), st->q_count, (size_t)nt * sizeof(int), hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    out[0] = out[1] = 0;
    for (int t = 0; t < nt; t++) out[t >= st->ntile_c] += q[t];
    return 0;
}

int abacus_hod_populate(abacus_hod_state *st, const abacus_hod_params *p, int64_t counts[6]) {
    ABACUS_ENTER();
    ABACUS_TRY(abacus_hod_populate_async(st, p));
    return abacus_hod_counts(st, counts);
}

int abacus_hod_fetch(abacus_hod_state *st, int tracer, double *x, double *y, double *z, double *vx, double *vy,
                     double *vz, double *mass, int64_t *id) {
    ABACUS_ENTER();
    if (tracer < 0 || tracer > 2) return fail("abacus_hod_fetch: tracer %d out of range", tracer);
    ABACUS_TRY(abacus_hod_counts(st, nullptr));
    const int64_t n = st->counts[tracer] + st->counts[3 + tracer];
    if (n == 0) return 0;
    OutCols o = out_cols(st);
    double *dst[7] = {x, y, z, vx, vy, vz, mass};
    for (int c = 0; c < 7; c++)
        if (dst[c]) HIP_TRY(hipMemcpyAsync(dst[c], o.c[tracer][c], n * sizeof(double), hipMemcpyDeviceToHost, stream()));
    if (id) HIP_TRY(hipMemcpyAsync(id, o.id[tracer], n * sizeof(int64_t), hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_hod_fetch_block(abacus_hod_state *st, int tracer, void *out8, int64_t n_expected) {
    ABACUS_ENTER();
    if (tracer < 0 || tracer > 2) return fail("abacus_hod_fetch_block: tracer %d out of range", tracer);
    if (!out8) return fail("abacus_hod_fetch_block: null output");
    ABACUS_TRY(abacus_hod_counts(st, nullptr));
    const int64_t n = st->counts[tracer] + st->counts[3 + tracer];
    if (n != n_expected) return fail("abacus_hod_fetch_block: the catalogue has %lld rows, the buffer %lld", (long long)n, (long long)n_expected);
    if (n == 0) return 0;
    // the 8 columns of a tracer are rows of one device allocation, `cap` elements apart: one 2-D copy
    HIP_TRY(hipMemcpy2DAsync(out8, (size_t)n * 8, st->out[tracer].p, (size_t)st->cap[tracer] * 8, (size_t)n * 8, 8,
                             hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_hod_device_columns(abacus_hod_state *st, int tracer, void *cols[8]) {
    ABACUS_ENTER();
    if (tracer < 0 || tracer > 2) return fail("abacus_hod_device_columns: tracer %d out of range", tracer);
    ABACUS_TRY(abacus_hod_counts(st, nullptr));
    OutCols o = out_cols(st);
    for (int c = 0; c < 7; c++) cols[c] = o.c[tracer][c];
    cols[7] = o.id[tracer];
    return 0;
}

int abacus_hod_fetch_keep(abacus_hod_state *st, int8_t *keep_cent, int8_t *keep_sat) {
    ABACUS_ENTER();
    if (!st || !st->have_run) return fail("abacus_hod_fetch_keep: populate has not been called");
    if (keep_cent && st->nh) HIP_TRY(hipMemcpyAsync(keep_cent, st->keep_c, st->nh, hipMemcpyDeviceToHost, stream()));
    if (keep_sat && st->np) HIP_TRY(hipMemcpyAsync(keep_sat, st->keep_s, st->np, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_hod_free(abacus_hod_state *st) {
    if (!st) return 0;
    std::lock_guard<std::recursive_mutex> guard(::abacus::api_mutex());
    (void)hipStreamSynchronize(stream());
    if (st->owns) {
        void *ptrs[] = {st->hpos, st->hvel, st->hmass, st->hid, st->hmultis, st->hrandoms, st->hveldev, st->hdeltac,
                        st->hfenv, st->hshear, st->ppos, st->pvel, st->phvel, st->phmass, st->phid, st->pweights,
                        st->prandoms, st->pdeltac, st->pfenv, st->pshear, st->pranks, st->pranksv, st->pranksp,
                        st->pranksr, st->pinds};
        for (void *q : ptrs)
            if (q) (void)hipFree(q);
    }
    void *work[] = {st->keep_c, st->keep_s, st->kept_c, st->kept_s, st->sb_counts, st->d_totals, st->q_count, st->queue_c, st->queue_s};
    for (void *q : work)
        if (q) (void)hipFree(q);
    if (st->owns_sigma && st->hsigma3d) (void)hipFree(st->hsigma3d);
    if (st->hc) (void)hipFree(st->hc);
    if (st->hrvir) (void)hipFree(st->hrvir);
    (void)st->ngal_bins.release(), (void)st->ngal_tables.release(), (void)st->ngal_out.release();
    (void)st->nfw_counts.release(), (void)st->nfw_offsets.release(), (void)st->nfw_draw.release(), (void)st->nfw_scan.release();
    if (st->h_totals) (void)hipHostFree(st->h_totals);
    for (int t = 0; t < 3; t++) (void)st->out[t].release();
    (void)st->shadow.release();
    (void)st->keys.release();
    (void)st->index_idx.release(), (void)st->index_scratch.release(), (void)st->index_tmp.release(), (void)st->index_last.release();
    (void)st->hrec.release(), (void)st->prec.release();
    delete st;
    return 0;
}

}  // extern "C"
