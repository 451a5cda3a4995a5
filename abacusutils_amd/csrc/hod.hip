// AbacusHOD population on MI355X (gfx950): replaces gen_cent / gen_sats / fast_concatenate of the reference
// (abacusnbody/hod/GRAND_HOD.py:139-414, 825-1262, 1265-1299), called as in gen_gals (:1477-1589).
//
// Data layout in HBM: the staged arrays keep the reference's layout (float64 SoA scalars, (N,3) C-order
// pos/vel, int64 ids) and stay resident across populate calls (the MCMC use case of staging()).
//
// Kernels (all HBM-bound streaming, no MFMA - there is no contraction on this path):
//   hod_decide_cent / hod_decide_sat   one pass over the per-halo / per-particle scalars (16-B coalesced loads,
//        FP64 occupation math), writes the int8 keep mask and per-tile tracer counts (wave ballots).
//        Tiles of 2048 objects -> thousands of workgroups for the 256 CUs.
//   hod_scan_tiles                     exclusive scan of the tile counts -> per-tile output offsets; satellites
//        start at Ncent so centrals||satellites land concatenated (no fast_concatenate pass).
//   hod_emit                           re-reads the 1-byte mask in blocked order, block-scans per-thread counts
//        and writes the galaxies in input order (stable compaction = the reference's order for any Nthread).
// All FP64 arithmetic that reaches an OUTPUT (velocity bias, RSD) is + - * / sqrt in the reference's order,
// compiled with -ffp-contract=off, so outputs are bit-identical to the CPU; the keep decision compares
// randoms against erfc/log10/pow-based markers whose last-ulp differences (ocml vs libm) only matter for a
// random within ~1e-16 of its marker.
#include <cmath>
#include <cstring>

#include "../../include/abacus_hip.h"
#include "common.hpp"

using namespace abacus;

namespace {

constexpr int TILE = 2048;    // objects per workgroup
constexpr int BLOCK = 256;    // threads per workgroup (4 waves)
constexpr int PER_THREAD = TILE / BLOCK;

// ---- occupation functions (hod/GRAND_HOD.py:23-136) ------------------------------------------------------
__device__ __forceinline__ double n_cen_LRG(double M_h, double logM_cut, double sigma) {
    return 0.5 * erfc((logM_cut - log10(M_h)) / (1.41421356 * sigma));
}
__device__ __forceinline__ double n_sat_LRG_modified(double M_h, double logM_cut, double M_cut, double M_1,
                                                     double sigma, double alpha, double kappa) {
    if (M_h - kappa * M_cut < 0) return 0;
    return pow((M_h - kappa * M_cut) / M_1, alpha) * 0.5 * erfc((logM_cut - log10(M_h)) / (1.41421356 * sigma));
}
__device__ __forceinline__ double N_sat_generic(double M_h, double M_cut, double kappa, double M_1, double alpha,
                                                double A_s) {
    if (M_h - kappa * M_cut < 0) return 0;
    return A_s * pow((M_h - kappa * M_cut) / M_1, alpha);
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

// marker chain of gen_sats pass 1 (hod/GRAND_HOD.py:957-1088)
__device__ __forceinline__ int8_t sat_decide(const abacus_hod_params &p, double hmass, double weights,
                                             double randoms, double d, double f, double sh, double r, double rv,
                                             double rp, double rr, int8_t keep_cent) {
    double LRG_marker = 0;
    if (p.want_LRG) {
        double M1 = pow(10.0, p.L_logM1 + p.L_Asat * d + p.L_Bsat * f);
        double lc = p.L_logM_cut + p.L_Acent * d + p.L_Bcent * f;
        double base =
            n_sat_LRG_modified(hmass, lc, pow(10.0, lc), M1, p.L_sigma, p.L_alpha, p.L_kappa) * weights * p.L_ic;
        double exp_sat = base;
        if (p.enable_ranks) {
            double dec = 1 + p.L_s * r + p.L_s_v * rv + p.L_s_p * rp + p.L_s_r * rr;
            exp_sat = base * dec;
        }
        LRG_marker += exp_sat;
    }
    double ELG_marker = LRG_marker;
    if (p.want_ELG) {
        double M1 = pow(10.0, p.E_logM1 + p.E_Asat * d + p.E_Bsat * f + p.E_Csat * sh);
        double lc = p.E_logM_cut + p.E_Acent * d + p.E_Bcent * f + p.E_Ccent * sh;
        double alpha = p.E_alpha;
        if (keep_cent == 1) {  // ELG conformity (:1006-1035); these branches carry no Csat term
            M1 = pow(10.0, p.E_logM1_EL + p.E_Asat * d + p.E_Bsat * f);
            alpha = p.E_alpha_EL;
        } else if (keep_cent == 2) {
            M1 = pow(10.0, p.E_logM1_EE + p.E_Asat * d + p.E_Bsat * f);
            alpha = p.E_alpha_EE;
        }
        double base = N_sat_generic(hmass, pow(10.0, lc), p.E_kappa, M1, alpha, p.E_A_s) * weights * p.E_ic;
        if (p.enable_ranks) {
            double dec = 1 + p.E_s * r + p.E_s_v * rv + p.E_s_p * rp + p.E_s_r * rr;
            base = base * dec;
        }
        ELG_marker += base;
    }
    double QSO_marker = ELG_marker;
    if (p.want_QSO) {
        double M1 = pow(10.0, p.Q_logM1 + p.Q_Asat * d + p.Q_Bsat * f);
        double lc = p.Q_logM_cut + p.Q_Acent * d + p.Q_Bcent * f;
        double base = N_sat_generic(hmass, pow(10.0, lc), p.Q_kappa, M1, p.Q_alpha, 1.0) * weights * p.Q_ic;
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

// per-tile tracer counts from the 64-lane ballots of each wave
__device__ __forceinline__ void tile_count(int8_t k0, int8_t k1, int *lds_counts) {
#pragma unroll
    for (int t = 1; t <= 3; t++) {
        unsigned long long b0 = __ballot(k0 == t), b1 = __ballot(k1 == t);
        if ((threadIdx.x & 63) == 0) {
            int c = __popcll(b0) + __popcll(b1);
            if (c) atomicAdd(&lds_counts[t - 1], c);
        }
    }
}

__global__ __launch_bounds__(BLOCK) void hod_decide_cent(int64_t n, const double *__restrict__ mass,
                                                         const double *__restrict__ multis,
                                                         const double *__restrict__ randoms,
                                                         const double *__restrict__ deltac,
                                                         const double *__restrict__ fenv,
                                                         const double *__restrict__ shear, abacus_hod_params p,
                                                         int8_t *__restrict__ keep, int *__restrict__ tile_counts) {
    __shared__ int cnt[4];
    if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t tile0 = (int64_t)blockIdx.x * TILE;
    const bool need_shear = p.want_ELG && shear != nullptr;
#pragma unroll
    for (int k = 0; k < PER_THREAD / 2; k++) {
        int64_t i = tile0 + (int64_t)k * (2 * BLOCK) + 2 * threadIdx.x;
        int8_t k0 = 0, k1 = 0;
        if (i < n) {
            double m0, m1, mu0, mu1, r0, r1, d0, d1, f0, f1, s0 = 0, s1 = 0;
            load2(mass, i, n, 1.0, m0, m1);
            load2(multis, i, n, 0.0, mu0, mu1);
            load2(randoms, i, n, 2.0, r0, r1);
            load2(deltac, i, n, 0.0, d0, d1);
            load2(fenv, i, n, 0.0, f0, f1);
            if (need_shear) load2(shear, i, n, 0.0, s0, s1);
            k0 = cent_decide(p, m0, mu0, r0, d0, f0, s0);
            if (i + 1 < n) {
                k1 = cent_decide(p, m1, mu1, r1, d1, f1, s1);
                *reinterpret_cast<char2 *>(keep + i) = make_char2(k0, k1);
            } else {
                keep[i] = k0;
            }
        }
        tile_count(k0, k1, cnt);
    }
    __syncthreads();
    if (threadIdx.x < 4) tile_counts[(int64_t)blockIdx.x * 4 + threadIdx.x] = cnt[threadIdx.x];
}

__global__ __launch_bounds__(BLOCK) void hod_decide_sat(
    int64_t n, const double *__restrict__ hmass, const double *__restrict__ weights,
    const double *__restrict__ randoms, const double *__restrict__ deltac, const double *__restrict__ fenv,
    const double *__restrict__ shear, const double *__restrict__ ranks, const double *__restrict__ ranksv,
    const double *__restrict__ ranksp, const double *__restrict__ ranksr, const int64_t *__restrict__ pinds,
    const int8_t *__restrict__ keep_cent, abacus_hod_params p, int8_t *__restrict__ keep,
    int *__restrict__ tile_counts) {
    __shared__ int cnt[4];
    if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t tile0 = (int64_t)blockIdx.x * TILE;
    const bool need_shear = p.want_ELG && shear != nullptr;
    const bool need_conf = p.want_ELG && pinds != nullptr;
    const bool need_ranks = p.enable_ranks != 0;
#pragma unroll
    for (int k = 0; k < PER_THREAD / 2; k++) {
        int64_t i = tile0 + (int64_t)k * (2 * BLOCK) + 2 * threadIdx.x;
        int8_t k0 = 0, k1 = 0;
        if (i < n) {
            double m0, m1, w0, w1, r0, r1, d0, d1, f0, f1, s0 = 0, s1 = 0;
            double a0 = 1, a1 = 1, b0 = 1, b1 = 1, c0 = 1, c1 = 1, e0 = 1, e1 = 1;
            int8_t kc0 = 0, kc1 = 0;
            load2(hmass, i, n, 1.0, m0, m1);
            load2(weights, i, n, 0.0, w0, w1);
            load2(randoms, i, n, 2.0, r0, r1);
            load2(deltac, i, n, 0.0, d0, d1);
            load2(fenv, i, n, 0.0, f0, f1);
            if (need_shear) load2(shear, i, n, 0.0, s0, s1);
            if (need_ranks) {
                load2(ranks, i, n, 1.0, a0, a1);
                load2(ranksv, i, n, 1.0, b0, b1);
                load2(ranksp, i, n, 1.0, c0, c1);
                load2(ranksr, i, n, 1.0, e0, e1);
            }
            if (need_conf) {  // keep_cent[pinds[i]] (hod/GRAND_HOD.py:1562), gathered here instead of on the host
                kc0 = keep_cent[pinds[i]];
                if (i + 1 < n) kc1 = keep_cent[pinds[i + 1]];
            }
            k0 = sat_decide(p, m0, w0, r0, d0, f0, s0, a0, b0, c0, e0, kc0);
            if (i + 1 < n) {
                k1 = sat_decide(p, m1, w1, r1, d1, f1, s1, a1, b1, c1, e1, kc1);
                *reinterpret_cast<char2 *>(keep + i) = make_char2(k0, k1);
            } else {
                keep[i] = k0;
            }
        }
        tile_count(k0, k1, cnt);
    }
    __syncthreads();
    if (threadIdx.x < 4) tile_counts[(int64_t)blockIdx.x * 4 + threadIdx.x] = cnt[threadIdx.x];
}

// Exclusive scan over tile counts.  One workgroup; thread t owns a contiguous run of tiles.
// tile_counts / tile_offsets: [(ntile_c + ntile_s)][4]; totals: [0..2] Ncent, [3..5] Nsat.
constexpr int SCAN_BLOCK = 1024;
__global__ __launch_bounds__(SCAN_BLOCK) void hod_scan_tiles(const int *__restrict__ tile_counts, int ntile_c,
                                                             int ntile_s, int64_t *__restrict__ tile_offsets,
                                                             int64_t *__restrict__ totals) {
    __shared__ int64_t part[SCAN_BLOCK][3];
    __shared__ int64_t ncent[3];
    for (int phase = 0; phase < 2; phase++) {
        const int base = phase == 0 ? 0 : ntile_c;
        const int nt = phase == 0 ? ntile_c : ntile_s;
        const int per = (nt + SCAN_BLOCK - 1) / SCAN_BLOCK;
        const int lo = min(nt, (int)threadIdx.x * per), hi = min(nt, lo + per);
        int64_t s[3] = {0, 0, 0};
        for (int b = lo; b < hi; b++)
#pragma unroll
            for (int t = 0; t < 3; t++) s[t] += tile_counts[(int64_t)(base + b) * 4 + t];
#pragma unroll
        for (int t = 0; t < 3; t++) part[threadIdx.x][t] = s[t];
        __syncthreads();
        // Hillis-Steele inclusive scan over the 1024 partial sums
        for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
            int64_t v[3] = {0, 0, 0};
            if ((int)threadIdx.x >= off)
#pragma unroll
                for (int t = 0; t < 3; t++) v[t] = part[threadIdx.x - off][t];
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 3; t++) part[threadIdx.x][t] += v[t];
            __syncthreads();
        }
        int64_t run[3];
#pragma unroll
        for (int t = 0; t < 3; t++) {
            run[t] = part[threadIdx.x][t] - s[t];       // exclusive prefix of this thread's run
            if (phase == 1) run[t] += ncent[t];          // satellites follow the centrals of their tracer
        }
        for (int b = lo; b < hi; b++)
#pragma unroll
            for (int t = 0; t < 3; t++) {
                tile_offsets[(int64_t)(base + b) * 4 + t] = run[t];
                run[t] += tile_counts[(int64_t)(base + b) * 4 + t];
            }
        if (threadIdx.x == SCAN_BLOCK - 1) {
#pragma unroll
            for (int t = 0; t < 3; t++) {
                totals[phase * 3 + t] = part[threadIdx.x][t];
                if (phase == 0) ncent[t] = part[threadIdx.x][t];
            }
        }
        __syncthreads();
    }
}

struct OutCols {
    double *c[3][7];  // [tracer][x,y,z,vx,vy,vz,mass]
    int64_t *id[3];
    int64_t cap[3];
};

// galaxy emission shared by centrals (hod/GRAND_HOD.py:298-325) and satellites (:1134-1165)
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
    o.c[t][0][j] = x;
    o.c[t][1][j] = y;
    o.c[t][2][j] = z;
    o.c[t][3][j] = vx;
    o.c[t][4][j] = vy;
    o.c[t][5][j] = vz;
    o.c[t][6][j] = mass;
    o.id[t][j] = id;
}

// Ordered emission.  Workgroups [0, ntile_c) handle central tiles, the rest satellite tiles.
// Thread t owns objects [tile0 + 8t, tile0 + 8t + 8): one 8-byte load of the mask, a packed 3x20-bit block scan.
__global__ __launch_bounds__(BLOCK) void hod_emit(int64_t nh, int64_t np, int ntile_c, const int8_t *__restrict__ keep_c,
                                                  const int8_t *__restrict__ keep_s,
                                                  const int64_t *__restrict__ tile_offsets,
                                                  const double *__restrict__ hpos, const double *__restrict__ hvel,
                                                  const double *__restrict__ hvdev, const double *__restrict__ hmass,
                                                  const int64_t *__restrict__ hid, const double *__restrict__ ppos,
                                                  const double *__restrict__ pvel, const double *__restrict__ phvel,
                                                  const double *__restrict__ phmass,
                                                  const int64_t *__restrict__ phid, abacus_hod_params p, OutCols o) {
    const bool sat = (int)blockIdx.x >= ntile_c;
    const int tile = sat ? blockIdx.x - ntile_c : blockIdx.x;
    const int64_t n = sat ? np : nh;
    const int8_t *keep = sat ? keep_s : keep_c;
    const int64_t i0 = (int64_t)tile * TILE + (int64_t)threadIdx.x * PER_THREAD;
    int8_t k[PER_THREAD];
    if (i0 + PER_THREAD <= n) {
        unsigned long long bits = *reinterpret_cast<const unsigned long long *>(keep + i0);
#pragma unroll
        for (int q = 0; q < PER_THREAD; q++) k[q] = (int8_t)((bits >> (8 * q)) & 0xff);
    } else {
#pragma unroll
        for (int q = 0; q < PER_THREAD; q++) k[q] = (i0 + q < n) ? keep[i0 + q] : (int8_t)0;
    }
    unsigned long long mine = 0;  // counts of tracer 1,2,3 in 20-bit fields
#pragma unroll
    for (int q = 0; q < PER_THREAD; q++)
        if (k[q]) mine += 1ull << (20 * (k[q] - 1));
    // block-wide exclusive scan of `mine`: wave scan by shuffles, then the 4 wave totals through LDS
    __shared__ unsigned long long wave_tot[BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        unsigned long long v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned long long excl = incl - mine;
    for (int w = 0; w < wave; w++) excl += wave_tot[w];
    if (!__syncthreads_or(mine != 0)) return;
    int64_t j[3];
#pragma unroll
    for (int t = 0; t < 3; t++)
        j[t] = tile_offsets[(int64_t)blockIdx.x * 4 + t] + (int64_t)((excl >> (20 * t)) & 0xfffff);
    if (mine == 0) return;
    const double alpha[3] = {sat ? p.L_alpha_s : p.L_alpha_c, sat ? p.E_alpha_s : p.E_alpha_c,
                             sat ? p.Q_alpha_s : p.Q_alpha_c};
#pragma unroll
    for (int q = 0; q < PER_THREAD; q++) {
        const int t = k[q] - 1;
        if (t < 0) continue;
        const int64_t i = i0 + q;
        double x, y, z, vx, vy, vz, m;
        int64_t id;
        if (!sat) {
            x = hpos[3 * i], y = hpos[3 * i + 1], z = hpos[3 * i + 2];
            vx = hvel[3 * i] + alpha[t] * hvdev[3 * i];  // velocity bias (:301-305)
            vy = hvel[3 * i + 1] + alpha[t] * hvdev[3 * i + 1];
            vz = hvel[3 * i + 2] + alpha[t] * hvdev[3 * i + 2];
            m = hmass[i];
            id = hid[i];
        } else {
            x = ppos[3 * i], y = ppos[3 * i + 1], z = ppos[3 * i + 2];
            vx = phvel[3 * i] + alpha[t] * (pvel[3 * i] - phvel[3 * i]);  // (:1136-1146)
            vy = phvel[3 * i + 1] + alpha[t] * (pvel[3 * i + 1] - phvel[3 * i + 1]);
            vz = phvel[3 * i + 2] + alpha[t] * (pvel[3 * i + 2] - phvel[3 * i + 2]);
            m = phmass[i];
            id = phid[i];
        }
        emit_one(p, o, t, j[t], x, y, z, vx, vy, vz, m, id);
        j[t]++;
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
    int64_t *hid = nullptr;
    double *ppos = nullptr, *pvel = nullptr, *phvel = nullptr, *phmass = nullptr, *pweights = nullptr,
           *prandoms = nullptr, *pdeltac = nullptr, *pfenv = nullptr, *pshear = nullptr, *pranks = nullptr,
           *pranksv = nullptr, *pranksp = nullptr, *pranksr = nullptr;
    int64_t *phid = nullptr, *pinds = nullptr;
    // work arrays
    int ntile_c = 0, ntile_s = 0;
    int8_t *keep_c = nullptr, *keep_s = nullptr;
    int *tile_counts = nullptr;
    int64_t *tile_offsets = nullptr;
    int64_t *d_totals = nullptr;  // 6
    int64_t *h_totals = nullptr;  // pinned, 6
    // outputs
    DevBuf out[3];
    int64_t cap[3] = {0, 0, 0};
    int64_t counts[6] = {0, 0, 0, 0, 0, 0};
    abacus_hod_params params;
    bool have_run = false, counts_valid = false;
};

namespace {

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

int launch_emit(abacus_hod_state *st) {
    const int ntiles = st->ntile_c + st->ntile_s;
    if (ntiles == 0) return 0;
    ABACUS_LAUNCH("hod_emit", hod_emit, dim3(ntiles), dim3(BLOCK), 0, st->nh, st->np, st->ntile_c, st->keep_c,
                  st->keep_s, st->tile_offsets, st->hpos, st->hvel, st->hveldev, st->hmass, st->hid, st->ppos,
                  st->pvel, st->phvel, st->phmass, st->phid, st->params, out_cols(st));
    return 0;
}

}  // namespace

extern "C" {

int abacus_hod_stage(const abacus_hod_arrays *a, int on_device, abacus_hod_state **out) {
    ABACUS_TRY(ensure_init());
    if (!a || !out) return fail("abacus_hod_stage: null argument");
    if (a->n_halo < 0 || a->n_part < 0) return fail("abacus_hod_stage: negative length");
    if (a->n_halo > 0 && (!a->hpos || !a->hvel || !a->hmass || !a->hid || !a->hmultis || !a->hrandoms || !a->hveldev))
        return fail("abacus_hod_stage: a required halo array is NULL");
    if (a->n_part > 0 && (!a->ppos || !a->pvel || !a->phvel || !a->phmass || !a->phid || !a->pweights || !a->prandoms))
        return fail("abacus_hod_stage: a required particle array is NULL");
    if (a->n_halo >= (int64_t)TILE * 0x7fffffff || a->n_part >= (int64_t)TILE * 0x7fffffff)
        return fail("abacus_hod_stage: too many objects for one device");
    auto *st = new abacus_hod_state();
    st->nh = a->n_halo;
    st->np = a->n_part;
    st->owns = !on_device;
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
    HIP_TRY(hipMalloc((void **)&st->keep_c, nh > 0 ? nh + 16 : 16));
    HIP_TRY(hipMalloc((void **)&st->keep_s, np > 0 ? np + 16 : 16));
    HIP_TRY(hipMalloc((void **)&st->tile_counts, (ntiles > 0 ? ntiles : 1) * 4 * sizeof(int)));
    HIP_TRY(hipMalloc((void **)&st->tile_offsets, (ntiles > 0 ? ntiles : 1) * 4 * sizeof(int64_t)));
    HIP_TRY(hipMalloc((void **)&st->d_totals, 8 * sizeof(int64_t)));
    HIP_TRY(hipHostMalloc((void **)&st->h_totals, 8 * sizeof(int64_t), hipHostMallocDefault));
    // first guess for the catalog buffers; grown on demand by abacus_hod_counts
    for (int t = 0; t < 3; t++) ABACUS_TRY(set_capacity(st, t, (nh + np) / 64));
    HIP_TRY(hipStreamSynchronize(stream()));
    *out = st;
    return 0;
}

int abacus_hod_update(abacus_hod_state *st, const char *field, const double *host) {
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
    return 0;
}

int abacus_hod_populate_async(abacus_hod_state *st, const abacus_hod_params *p) {
    ABACUS_TRY(ensure_init());
    if (!st || !p) return fail("abacus_hod_populate: null argument");
    if (p->enable_ranks && st->np > 0 && (!st->pranks || !st->pranksv || !st->pranksp || !st->pranksr))
        return fail("abacus_hod_populate: enable_ranks set but the rank arrays were not staged");
    if (p->want_ELG && st->np > 0 && !st->pinds)
        return fail("abacus_hod_populate: ELG conformity needs pinds to be staged");
    st->params = *p;
    if (st->ntile_c)
        ABACUS_LAUNCH("hod_decide_cent", hod_decide_cent, dim3(st->ntile_c), dim3(BLOCK), 0, st->nh, st->hmass,
                      st->hmultis, st->hrandoms, st->hdeltac, st->hfenv, st->hshear, *p, st->keep_c, st->tile_counts);
    if (st->ntile_s)
        ABACUS_LAUNCH("hod_decide_sat", hod_decide_sat, dim3(st->ntile_s), dim3(BLOCK), 0, st->np, st->phmass,
                      st->pweights, st->prandoms, st->pdeltac, st->pfenv, st->pshear, st->pranks, st->pranksv,
                      st->pranksp, st->pranksr, st->pinds, st->keep_c, *p, st->keep_s,
                      st->tile_counts + (int64_t)st->ntile_c * 4);
    ABACUS_LAUNCH("hod_scan_tiles", hod_scan_tiles, dim3(1), dim3(SCAN_BLOCK), 0, st->tile_counts, st->ntile_c,
                  st->ntile_s, st->tile_offsets, st->d_totals);
    // speculative emission into the current buffers (writes past capacity are suppressed on the device)
    ABACUS_TRY(launch_emit(st));
    HIP_TRY(hipMemcpyAsync(st->h_totals, st->d_totals, 6 * sizeof(int64_t), hipMemcpyDeviceToHost, stream()));
    st->have_run = true;
    st->counts_valid = false;
    return 0;
}

int abacus_hod_counts(abacus_hod_state *st, int64_t counts[6]) {
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

int abacus_hod_populate(abacus_hod_state *st, const abacus_hod_params *p, int64_t counts[6]) {
    ABACUS_TRY(abacus_hod_populate_async(st, p));
    return abacus_hod_counts(st, counts);
}

int abacus_hod_fetch(abacus_hod_state *st, int tracer, double *x, double *y, double *z, double *vx, double *vy,
                     double *vz, double *mass, int64_t *id) {
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

int abacus_hod_device_columns(abacus_hod_state *st, int tracer, void *cols[8]) {
    if (tracer < 0 || tracer > 2) return fail("abacus_hod_device_columns: tracer %d out of range", tracer);
    ABACUS_TRY(abacus_hod_counts(st, nullptr));
    OutCols o = out_cols(st);
    for (int c = 0; c < 7; c++) cols[c] = o.c[tracer][c];
    cols[7] = o.id[tracer];
    return 0;
}

int abacus_hod_fetch_keep(abacus_hod_state *st, int8_t *keep_cent, int8_t *keep_sat) {
    if (!st || !st->have_run) return fail("abacus_hod_fetch_keep: populate has not been called");
    if (keep_cent && st->nh) HIP_TRY(hipMemcpyAsync(keep_cent, st->keep_c, st->nh, hipMemcpyDeviceToHost, stream()));
    if (keep_sat && st->np) HIP_TRY(hipMemcpyAsync(keep_sat, st->keep_s, st->np, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_hod_free(abacus_hod_state *st) {
    if (!st) return 0;
    (void)hipStreamSynchronize(stream());
    if (st->owns) {
        void *ptrs[] = {st->hpos, st->hvel, st->hmass, st->hid, st->hmultis, st->hrandoms, st->hveldev, st->hdeltac,
                        st->hfenv, st->hshear, st->ppos, st->pvel, st->phvel, st->phmass, st->phid, st->pweights,
                        st->prandoms, st->pdeltac, st->pfenv, st->pshear, st->pranks, st->pranksv, st->pranksp,
                        st->pranksr, st->pinds};
        for (void *q : ptrs)
            if (q) (void)hipFree(q);
    }
    void *work[] = {st->keep_c, st->keep_s, st->tile_counts, st->tile_offsets, st->d_totals};
    for (void *q : work)
        if (q) (void)hipFree(q);
    if (st->h_totals) (void)hipHostFree(st->h_totals);
    for (int t = 0; t < 3; t++) (void)st->out[t].release();
    delete st;
    return 0;
}

}  // extern "C"
