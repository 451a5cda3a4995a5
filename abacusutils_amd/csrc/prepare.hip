// Data-parallel core of prepare_sim.prepare_slab (abacusnbody/hod/prepare_sim.py:296-1052) on the device:
//   abacus_prepare_halo_factors   subsample_halos (:83-108) + the halo mask (:449) + submask_particles' target count (:152-174)
//   abacus_prepare_particles      per-halo particle selection (a random subset of `ntarget` of the halo's subsample particles:
//                                 the reference's per-halo `np.random.choice(replace=False)` loop, :871-875), the new halo
//                                 offsets (:895-897), the list of kept particles with their host halo and `Np`, and the five
//                                 satellite rank columns (:899-977)
// The concentration / environment / shear ranks per mass bin reuse abacus_fenv_rank (staging.hip), the environment masses
// abacus_menv (catalog.hip).  File I/O and the CompaSO readers stay on the host side (out of scope).
// Random numbers: the caller either passes the selection it drew itself (`submask`: abacusutils_amd/hod/prepare_sim.py draws it
// from NumPy's legacy generator in the reference's order, which makes a run comparable with the reference value for value) or a
// Philox seed - then every particle gets a counter-based 32-bit key and a halo keeps the `ntarget` smallest keys of its slice
// (uniform over subsets; one stable radix sort of (halo, key) pairs for the whole slab, no per-halo serial work).
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <vector>

#include "../../include/abacus_hip.h"
#include "common.hpp"

namespace abacus {
int fenv_rank_device(const double *d_M, const double *d_env, int64_t n, const double *d_edges, int n_edges, double *d_out);   // staging.hip
}
using namespace abacus;

namespace {

int grid_for(int64_t n) { return (int)std::min<int64_t>(std::max<int64_t>(ceil_div(n, 256), 1), 256 * 32); }

struct Tmp {   // device allocations of one call, released on every exit path
    std::vector<void *> p;
    ~Tmp() {
        for (void *q : p)
            if (q) scratch_release(q);     // kept for the next call (runtime.hip), not freed
    }
    template <class T>
    int alloc(T **out, size_t count) {
        void *q = nullptr;
        ABACUS_TRY(scratch_acquire(&q, std::max<size_t>(count, 1) * sizeof(T)));
        p.push_back(q);
        *out = static_cast<T *>(q);
        return 0;
    }
    template <class T>
    int upload(T **out, const T *host, size_t count) {
        ABACUS_TRY(alloc(out, count));
        if (count) HIP_TRY(hipMemcpyAsync(*out, host, count * sizeof(T), hipMemcpyHostToDevice, stream()));
        return 0;
    }
};

// ---- halos: kept fraction, mask, particle target --------------------------------------------------------------------------
// mass = N[i] * Mpart (halos counted in whole particles, prepare_slab's `halos['N'] * Mpart`), or mass64[i] when N is null
__global__ void prep_halo_factors(const unsigned int *__restrict__ N, const double *__restrict__ mass64, int64_t n, double Mpart, int MT,
                                  const double *__restrict__ u, const long long *__restrict__ pnum, double *__restrict__ p_out,
                                  unsigned char *__restrict__ mask, int *__restrict__ ntarget) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double m = N ? (double)N[i] * Mpart : mass64[i];
        const double x = log10(m);
        double f;
        if (!MT) {                                             // LRG only (:103-108)
            f = 1.0 / (1.0 + 0.1 * exp(-(x - 11.8) * 10));
            if (x > 13.0) f = 1.0;
        } else if (x < 11.4) {                                 // ELG-capable sample (:88-96)
            f = 0.2 / (1.0 + 10 * exp(-(x - 11.2) * 25));
        } else if (x < 11.6) {
            f = 0.4 / (1.0 + 10 * exp(-(x - 11.3) * 25));
        } else {
            f = 1.0 / (1.0 + 0.1 * exp(-(x - 11.7) * 10));
        }
        p_out[i] = f;
        if (mask) mask[i] = u[i] < f ? 1 : 0;                  // np.random.random(len(halos)) < p_halos (:449)
        if (ntarget) {                                         // submask_particles (:152-174)
            const long long n_in = pnum ? pnum[i] : 0;
            long long t = 0;
            if (MT) {
                if (!(m < 1e11)) t = min(min(n_in, (long long)(1 + 1.5 * pow(10.0, x - 12.5))), 100ll);
            } else {
                if (!(pow(10.0, x) < 1e12)) t = min(n_in, (long long)(1 + 1.5 * pow(10.0, x - 13)));
            }
            ntarget[i] = (int)max(t, 0ll);
        }
    }
}

// ---- particle selection -------------------------------------------------------------------------------------------------
#include "rng_device.hpp"

// The random columns of the two tables drawn on the device (`rng=<seed>` of prepare_slab_arrays; the reference draws them from
// NumPy's global generator, :984-996,1029 - same distributions and dtypes, a different stream).  Row r stands for the object of
// global index g = index0 + (index ? index[r] : r).  Halos (stream 4, five blocks of four words): `randoms` U[0,1) float64;
// `randoms_exp` = sign * Exp(1) * scale per component (:986-988: (randint(0,2)*2-1) * exponential(scale)); `randoms_gaus_vrms`
// = N(0,1) * scale (Box-Muller; log / sin / cos by the fixed float64 evaluations of rng_device.hpp, so the oracle
// restates the stream bit for bit).  One uniform per object on a stream of the caller's (5: the particles' `randoms`, 6: the
// halo mask's draws).
__global__ void prep_halo_randoms(int64_t n, const long long *__restrict__ index, long long index0, uint2 key,
                                  const double *__restrict__ scale, double *__restrict__ rnd, double *__restrict__ rexp,
                                  double *__restrict__ rgaus) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long g = (unsigned long long)(index ? index[r] + index0 : index0 + r);
        const unsigned int g0 = (unsigned int)g, g1 = (unsigned int)(g >> 32);
        const uint4 b0 = philox4x32_10(make_uint4(g0, g1, 4u, 0u), key), b1 = philox4x32_10(make_uint4(g0, g1, 4u, 1u), key),
                    b2 = philox4x32_10(make_uint4(g0, g1, 4u, 2u), key), b3 = philox4x32_10(make_uint4(g0, g1, 4u, 3u), key),
                    b4 = philox4x32_10(make_uint4(g0, g1, 4u, 4u), key);
        const double sc = scale[r];
        rnd[r] = u53(b0.x, b0.y);
        const double e[3] = {-det_log(1.0 - u53(b1.x, b1.y)), -det_log(1.0 - u53(b1.z, b1.w)), -det_log(1.0 - u53(b2.x, b2.y))};
#pragma unroll
        for (int k = 0; k < 3; k++) rexp[3 * r + k] = ((b0.z >> k) & 1u ? 1.0 : -1.0) * (e[k] * sc);
        const double m0 = sqrt(-2.0 * det_log(1.0 - u53(b2.z, b2.w))), m1 = sqrt(-2.0 * det_log(1.0 - u53(b3.z, b3.w)));
        double s0, c0, s1, c1;
        det_sincos2pi(u53(b3.x, b3.y), &s0, &c0);
        det_sincos2pi(u53(b4.x, b4.y), &s1, &c1);
        rgaus[3 * r] = (m0 * c0) * sc, rgaus[3 * r + 1] = (m0 * s0) * sc, rgaus[3 * r + 2] = (m1 * c1) * sc;
    }
}
__global__ void prep_part_randoms(int64_t n, const long long *__restrict__ index, long long index0, uint2 key,
                                  unsigned int stream_id, double *__restrict__ rnd) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long g = (unsigned long long)(index ? index[r] + index0 : index0 + r);
        const uint4 w = philox4x32_10(make_uint4((unsigned int)g, (unsigned int)(g >> 32), stream_id, 0u), key);
        rnd[r] = u53(w.x, w.y);
    }
}

// one wave per halo: host index of the halo's particles; sort key (candidate rank << 32 | Philox word) of candidates
__global__ __launch_bounds__(256) void prep_fill_host(const long long *__restrict__ pstart, const long long *__restrict__ pnum,
                                                      const unsigned char *__restrict__ hmask, int64_t nh, int64_t npart,
                                                      int *__restrict__ host, int *__restrict__ bad) {
    const int lane = threadIdx.x & 63;
    for (int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); j < nh; j += (int64_t)gridDim.x * 4) {
        if (!hmask[j] || pnum[j] <= 0) continue;
        const long long a = pstart[j], b = a + pnum[j];
        if (a < 0 || b > npart) {
            if (lane == 0) atomicOr(bad, 1);
            continue;
        }
        for (long long q = a + lane; q < b; q += 64) host[q] = (int)j;
    }
}
__global__ void prep_keys(const int *__restrict__ host, const int *__restrict__ ntarget, const long long *__restrict__ pnum,
                          int64_t npart, unsigned long long seed, long long part_index0, unsigned long long *__restrict__ key,
                          unsigned int *__restrict__ idx, unsigned char *__restrict__ submask) {
    const uint2 k2 = make_uint2((unsigned int)seed, (unsigned int)(seed >> 32));
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < npart; q += (int64_t)gridDim.x * blockDim.x) {
        const int j = host[q];
        idx[q] = (unsigned int)q;
        unsigned long long kq = ~0ull;                      // not a candidate: behind every candidate
        unsigned char sel = 0;
        if (j >= 0 && ntarget[j] > 0) {
            if ((long long)ntarget[j] >= pnum[j]) sel = 1;  // the whole slice is kept: nothing to draw
            else {
                const unsigned long long g = (unsigned long long)(part_index0 + q);
                const uint4 w = philox4x32_10(make_uint4((unsigned int)g, (unsigned int)(g >> 32), 3u, 0u), k2);   // stream 3
                kq = ((unsigned long long)(unsigned int)j << 32) | w.x;
            }
        }
        key[q] = kq;
        submask[q] = sel;
    }
}
// sorted position p holds candidate idx[p] of halo key >> 32; the halo's candidates start at cstart[halo]
__global__ void prep_pick(const unsigned long long *__restrict__ key, const unsigned int *__restrict__ idx, int64_t npart,
                          const long long *__restrict__ cstart, const int *__restrict__ ntarget, unsigned char *__restrict__ submask) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npart; p += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long k = key[p];
        if (k == ~0ull) continue;
        const int j = (int)(k >> 32);
        if (p - cstart[j] < (long long)ntarget[j]) submask[idx[p]] = 1;
    }
}
__global__ void prep_cand_count(const unsigned char *__restrict__ hmask, const long long *__restrict__ pnum, const int *__restrict__ ntarget,
                                int64_t nh, long long *__restrict__ ccount) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nh; j += (int64_t)gridDim.x * blockDim.x)
        ccount[j] = (hmask[j] && pnum[j] > 0 && ntarget[j] > 0 && (long long)ntarget[j] < pnum[j]) ? pnum[j] : 0;
}

// one wave per halo: kept particles of the halo
__global__ __launch_bounds__(256) void prep_count(const long long *__restrict__ pstart, const long long *__restrict__ pnum,
                                                  const unsigned char *__restrict__ hmask, const unsigned char *__restrict__ submask,
                                                  int64_t nh, long long *__restrict__ kept) {
    const int lane = threadIdx.x & 63;
    for (int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); j < nh; j += (int64_t)gridDim.x * 4) {
        long long c = 0;
        if (hmask[j] && pnum[j] > 0)
            for (long long q = pstart[j] + lane; q < pstart[j] + pnum[j]; q += 64) c += submask[q];
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
        if (lane == 0) kept[j] = c;
    }
}
__global__ void prep_halo_offsets(const unsigned char *__restrict__ hmask, const long long *__restrict__ pnum,
                                  const long long *__restrict__ kept, const long long *__restrict__ kstart, int64_t nh,
                                  double *__restrict__ pstart_new, double *__restrict__ pnum_new) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nh; j += (int64_t)gridDim.x * blockDim.x) {
        const bool live = hmask[j] && pnum[j] > 0;            // (:868, :979-981)
        pstart_new[j] = live ? (double)kstart[j] : -1.0;
        pnum_new[j] = live ? (double)kept[j] : -1.0;
    }
}
__global__ void prep_flags(const unsigned char *__restrict__ submask, const int *__restrict__ host, int64_t npart, long long *__restrict__ flag) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < npart; q += (int64_t)gridDim.x * blockDim.x)
        flag[q] = (submask[q] && host[q] >= 0) ? 1 : 0;
}
__global__ void prep_emit(const long long *__restrict__ flag, const long long *__restrict__ slot, const int *__restrict__ host,
                          const long long *__restrict__ kept, int64_t npart, long long *__restrict__ sel_idx,
                          long long *__restrict__ sel_host, double *__restrict__ sel_np) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < npart; q += (int64_t)gridDim.x * blockDim.x)
        if (flag[q]) {
            const long long s = slot[q];
            sel_idx[s] = q;
            sel_host[s] = host[q];
            sel_np[s] = (double)kept[host[q]];
        }
}

// ---- satellite ranks (:899-977): one workgroup per halo with at least two kept particles -------------------------------------
struct RankArgs {
    const float *pos, *vel;              // (npart, 3)
    const float *hpos, *hvel;            // (nh, 3)
    const unsigned int *N;
    const float *r25, *r98;
    const long long *pstart, *pnum, *kept, *kstart;
    const long long *sel_idx;            // kept particles in output order (ascending input index: halo after halo)
    double Mpart, h;
    double *ranks, *ranksv, *ranksp, *ranksr, *ranksc;   // (n_sel)
    const int *work;                     // halos to rank
    int kmax;
};

__device__ __forceinline__ bool key_before(double a, int ia, double b, int ib) {   // ascending, NaN last, ties by index
    const bool na = a != a, nb = b != b;
    if (na || nb) return (!na && nb) || (na && nb && ia < ib);
    return a < b || (a == b && ia < ib);
}

// NumPy VERSION ASSUMPTION: the perihelion iteration below follows the reference's dtypes under NumPy >= 2 scalar promotion
// (NEP 50): a float32 array element times a Python float stays float32 (1 / (log(1 + c) - c / (1 + c)) * 2 * 6.67e-11 and the
// first iteration are float32).  Under NumPy 1.x value-based casting the same reference lines run in float64, alpha and x2
// differ in their low bits and near-ties of `ranksp` may order differently: the goldens (oracle/make_golden.py, which
// asserts numpy >= 2) and the "value for value" statement of DESIGN.md section 2 hold against a reference run under NumPy 2.
__global__ __launch_bounds__(256) void prep_ranks(RankArgs A, int nwork) {
    extern __shared__ __align__(16) unsigned char smem[];
    for (int wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
        const int j = A.work[wi];
        const int k = (int)A.kept[j];
        const long long o0 = A.kstart[j];
        float *px = reinterpret_cast<float *>(smem);          // [3][k] positions of the kept particles
        double *key = reinterpret_cast<double *>(smem + (size_t)((3 * k * 4 + 15) / 16) * 16);   // [5][k]
        const float hx = A.hpos[3 * j], hy = A.hpos[3 * j + 1], hz = A.hpos[3 * j + 2];
        const float hvx = A.hvel[3 * j], hvy = A.hvel[3 * j + 1], hvz = A.hvel[3 * j + 2];
        const double m = (double)A.N[j] * A.Mpart / A.h;      // halos['N'][j] * Mpart / h (:934)
        const float rs = A.r25[j];
        const float c = A.r98[j] / rs;
        const float onec = 1.f + c;
        const float lc = (float)log((double)onec);            // np.log of a float32 scalar
        const float t = lc - c / onec;
        float a1 = 1.0f / t;
        a1 = a1 * 2.f;
        a1 = a1 * (float)6.67e-11;
        __syncthreads();
        for (int i = threadIdx.x; i < k; i += blockDim.x) {
            const long long q = A.sel_idx[o0 + i];
            const float x = A.pos[3 * q], y = A.pos[3 * q + 1], z = A.pos[3 * q + 2];
            px[i] = x, px[k + i] = y, px[2 * k + i] = z;
            const float rx = x - hx, ry = y - hy, rz = z - hz;
            const float d2 = (rx * rx + ry * ry) + rz * rz;    // np.sum(r_rel**2, axis=1) in float32
            const float vx = A.vel[3 * q] - hvx, vy = A.vel[3 * q + 1] - hvy, vz = A.vel[3 * q + 2] - hvz;
            const float v2 = (vx * vx + vy * vy) + vz * vz;
            const float r0 = __fsqrt_rn(d2);
            const float nx = rx / r0, ny = ry / r0, nz = rz / r0;
            const float vrad = (vx * nx + vy * ny) + vz * nz;
            const float vrad2 = vrad * vrad, vtan2 = v2 - vrad2;
            const float r0k = r0 * 1000.f;
            double al = (double)a1 * m;
            al = al * 2e30;
            al = al / (double)r0k;
            al = al / 3.086e19;
            al = al / 1e6;
            const float Af = vtan2 + vrad2;
            float x2f = vtan2 / Af;
            const float Bf = (float)log((double)(1.f + r0k / rs));
            // first iteration: float32 until alpha (float64) enters (:958-965)
            const float ox = __fsqrt_rn(x2f);
            float lg = ox * r0k;
            lg = lg / rs;
            lg = 1.f + lg;
            lg = (float)log((double)lg);
            lg = lg / ox;
            lg = lg - Bf;
            double x2 = (double)vtan2 / ((double)Af + al * (double)lg);
            for (int it = 1; it < 20; it++) {
                const double od = sqrt(x2);
                double w = od * (double)r0k;
                w = w / (double)rs;
                w = log(1.0 + w) / od - (double)Bf;
                x2 = (double)vtan2 / ((double)Af + al * w);
            }
            if (x2 != x2) x2 = 1.0;
            key[i] = (double)d2;
            key[k + i] = (double)v2;
            key[2 * k + i] = (double)(r0k * r0k) * x2;
            key[3 * k + i] = (double)vrad;
        }
        __syncthreads();
        // nearest other particle among ALL subsample particles of the halo (cKDTree.query(k=2)[0][:, 1], :912-913)
        const long long a = A.pstart[j], n_in = A.pnum[j];
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int i = wave; i < k; i += blockDim.x / 64) {
            const long long self = A.sel_idx[o0 + i];
            const double x = (double)px[i], y = (double)px[k + i], z = (double)px[2 * k + i];
            double best = INFINITY;
            for (long long q = a + lane; q < a + n_in; q += 64) {
                if (q == self) continue;
                const double dx = x - (double)A.pos[3 * q], dy = y - (double)A.pos[3 * q + 1], dz = z - (double)A.pos[3 * q + 2];
                const double d = (dx * dx + dy * dy) + dz * dz;
                best = d < best ? d : best;
            }
            for (int off = 32; off > 0; off >>= 1) {
                const double o = __shfl_down(best, off, 64);
                best = o < best ? o : best;
            }
            if (lane == 0) key[4 * k + i] = sqrt(best);
        }
        __syncthreads();
        const double mean = 0.5 * (double)(k - 1);             // np.mean of the ranks 0 .. k-1
        double *outs[5] = {A.ranks, A.ranksv, A.ranksp, A.ranksr, A.ranksc};
        for (int e = threadIdx.x; e < 5 * k; e += blockDim.x) {
            const int col = e / k, i = e - col * k;
            const double *kc = key + (size_t)col * k;
            const double v = kc[i];
            int r = 0;
            for (int q = 0; q < k; q++) r += key_before(kc[q], q, v, i) ? 1 : 0;
            outs[col][o0 + i] = ((double)r - mean) / mean;     // (newranks - mean) / mean
        }
        __syncthreads();
    }
}

__global__ void prep_rank_work(const long long *__restrict__ kept, int64_t nh, int *__restrict__ work, int *__restrict__ nwork,
                               int *__restrict__ kmax) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nh; j += (int64_t)gridDim.x * blockDim.x)
        if (kept[j] >= 2) {
            work[atomicAdd(nwork, 1)] = (int)j;
            atomicMax(kmax, (int)min(kept[j], 0x7fffffffll));
        }
}
__global__ void prep_rank_single(const long long *__restrict__ kept, const long long *__restrict__ kstart, int64_t nh,
                                 double *r0, double *r1, double *r2, double *r3, double *r4) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nh; j += (int64_t)gridDim.x * blockDim.x)
        if (kept[j] == 1) {                                   // a lone particle ranks 0 in every column (:889-895)
            const long long s = kstart[j];
            r0[s] = r1[s] = r2[s] = r3[s] = r4[s] = 0.0;
        }
}

template <class T>
int exclusive_sum(const T *in, T *out, int64_t n, Tmp &tmp) {
    size_t bytes = 0;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, (int)n, stream()));
    void *ws;
    ABACUS_TRY(tmp.alloc((unsigned char **)&ws, bytes));
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(ws, bytes, in, out, (int)n, stream()));
    return 0;
}

// ---- the whole slab on the device (abacus_prepare_slab): gathers of the kept rows ------------------------------------------
bool on_device(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();   // plain host memory: not an error
        return false;
    }
    return a.type == hipMemoryTypeDevice;
}
// a column of the caller's: used in place when it already lives in HBM (the reader's unpack kernels leave pos / vel there), else uploaded
template <class T>
int stage_col(Tmp &t, const T *src, size_t count, const T **out) {
    if (!src || on_device(src)) {
        *out = src;
        return 0;
    }
    T *d;
    ABACUS_TRY(t.upload(&d, src, count));
    *out = d;
    return 0;
}
__global__ void prep_mass_deltac(const unsigned int *__restrict__ N, const float *__restrict__ r25, const float *__restrict__ r98, int64_t n,
                                 double Mpart, double *__restrict__ mass, double *__restrict__ dval) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        mass[i] = (double)N[i] * Mpart;                      // halos['N'] * Mpart (:445)
        dval[i] = (double)(r98[i] / r25[i]);                 // float32 ratio, ranked in float64 (:762-773)
    }
}
__global__ void prep_flag_u8(const unsigned char *__restrict__ m, int64_t n, long long *__restrict__ flag) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) flag[i] = m[i] ? 1 : 0;
}
__global__ void prep_compact(const long long *__restrict__ flag, const long long *__restrict__ slot, int64_t n, long long *__restrict__ idx) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (flag[i]) idx[slot[i]] = i;
}
// rows of W 4-byte words
template <int W>
__global__ void prep_gather_w(const unsigned int *__restrict__ src, const long long *__restrict__ idx, int64_t n, unsigned int *__restrict__ dst) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n * W; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / W;
        dst[e] = src[idx[r] * W + (e - r * W)];
    }
}
__global__ void prep_gather_f3d(const float *__restrict__ src, const long long *__restrict__ idx, int64_t n, double *__restrict__ dst) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n * 3; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / 3;
        dst[e] = (double)src[idx[r] * 3 + (e - r * 3)];      // hvel[host].astype(float64)
    }
}
__global__ void prep_gather_recip(const double *__restrict__ p, const long long *__restrict__ idx, int64_t n, double *__restrict__ dst) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) dst[r] = 1.0 / p[idx[r]];
}
__global__ void prep_gather_scale(const float *__restrict__ sig, const long long *__restrict__ idx, int64_t n, double *__restrict__ dst) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x)
        dst[r] = (double)sig[idx[r]] / 1.7320508075688772;   // sigmav3d[kept] / np.sqrt(3): float32 column, float64 scalar
}

// what one abacus_prepare_slab call leaves in HBM for abacus_prepare_slab_fetch
struct SlabOut {
    Tmp *tmp = nullptr;
    int64_t nk = 0, ns = 0;
    int want_ranks = 0;
    bool ranks_on_side = false;      // the rank columns are being written on g_slab_side: the fetch copies them behind g_slab_ranked
    void *hcol[ABACUS_PREP_HALO_COLS] = {};
    size_t hbytes[ABACUS_PREP_HALO_COLS] = {};
    void *pcol[ABACUS_PREP_PART_COLS] = {};
    size_t pbytes[ABACUS_PREP_PART_COLS] = {};
    void clear();
};
SlabOut g_slab;
// prep_ranks (2.7 of the slab's 4 ms of kernels) runs on a stream of its own, so that the copies of the other columns - the copy
// engine - go out under it; with the profiler on it stays on the library stream (whose event pairs time it)
hipStream_t g_slab_side = nullptr;
hipEvent_t g_slab_ready = nullptr, g_slab_ranked = nullptr;
void SlabOut::clear() {
    if (ranks_on_side && g_slab_side) (void)hipStreamSynchronize(g_slab_side);   // nothing may still read what goes back to the pool
    delete tmp;
    *this = SlabOut();
}

template <int W>
int gather_w(const void *src, const long long *idx, int64_t n, void *dst) {
    if (n > 0) ABACUS_LAUNCH("prep_gather", (prep_gather_w<W>), dim3(grid_for(n * W)), dim3(256), 0, (const unsigned int *)src, idx, n, (unsigned int *)dst);
    return 0;
}

}  // namespace

extern "C" {

int abacus_prepare_halo_factors(const uint32_t *N, int64_t n, double Mpart, int MT, const double *u, const int64_t *pnum,
                                double *p_halos, uint8_t *mask, int32_t *ntarget) {
    ABACUS_ENTER();
    if (n < 0 || (n > 0 && (!N || !p_halos))) return fail("abacus_prepare_halo_factors: null argument");
    if (mask && !u) return fail("abacus_prepare_halo_factors: the halo mask needs the uniform draws");
    if (ntarget && !pnum) return fail("abacus_prepare_halo_factors: the particle targets need npoutA");
    if (n == 0) return 0;
    Tmp tmp;
    unsigned int *dN;
    double *du = nullptr, *dp;
    long long *dnum = nullptr;
    unsigned char *dm = nullptr;
    int *dt = nullptr;
    ABACUS_TRY(tmp.upload(&dN, (const unsigned int *)N, (size_t)n));
    if (mask) ABACUS_TRY(tmp.upload(&du, u, (size_t)n));
    if (ntarget) ABACUS_TRY(tmp.upload(&dnum, (const long long *)pnum, (size_t)n));
    ABACUS_TRY(tmp.alloc(&dp, (size_t)n));
    if (mask) ABACUS_TRY(tmp.alloc(&dm, (size_t)n));
    if (ntarget) ABACUS_TRY(tmp.alloc(&dt, (size_t)n));
    ABACUS_LAUNCH("prep_halo_factors", prep_halo_factors, dim3(grid_for(n)), dim3(256), 0, (const unsigned int *)dN, (const double *)nullptr, n,
                  Mpart, MT, du, dnum, dp, dm, dt);
    HIP_TRY(hipMemcpyAsync(p_halos, dp, (size_t)n * 8, hipMemcpyDeviceToHost, stream()));
    if (mask) HIP_TRY(hipMemcpyAsync(mask, dm, (size_t)n, hipMemcpyDeviceToHost, stream()));
    if (ntarget) HIP_TRY(hipMemcpyAsync(ntarget, dt, (size_t)n * 4, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_prepare_halo_factors_mass(const double *mass, int64_t n, int MT, double *p_halos) {
    ABACUS_ENTER();
    if (n < 0 || (n > 0 && (!mass || !p_halos))) return fail("abacus_prepare_halo_factors_mass: null argument");
    if (n == 0) return 0;
    Tmp tmp;
    double *dmass, *dp;
    ABACUS_TRY(tmp.upload(&dmass, mass, (size_t)n));
    ABACUS_TRY(tmp.alloc(&dp, (size_t)n));
    ABACUS_LAUNCH("prep_halo_factors", prep_halo_factors, dim3(grid_for(n)), dim3(256), 0, (const unsigned int *)nullptr, (const double *)dmass, n, 1.0,
                  MT, (const double *)nullptr, (const long long *)nullptr, dp, (unsigned char *)nullptr, (int *)nullptr);
    HIP_TRY(hipMemcpyAsync(p_halos, dp, (size_t)n * 8, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_prepare_particles(int64_t nh, const uint8_t *hmask, const int64_t *pstart, const int64_t *pnum, const uint32_t *N,
                             const float *hpos, const float *hvel, const float *r25, const float *r98, int64_t npart,
                             const float *pos, const float *vel, const uint8_t *submask_in, const int32_t *ntarget, uint64_t seed,
                             int64_t part_index0, double Mpart, double h, int want_ranks, double *pstart_new, double *pnum_new,
                             int64_t *n_sel, int64_t cap_sel, int64_t *sel_idx, int64_t *sel_host, double *sel_np, double *ranks,
                             double *ranksv, double *ranksp, double *ranksr, double *ranksc, uint8_t *submask_out) {
    ABACUS_ENTER();
    if (nh < 0 || npart < 0 || !n_sel) return fail("abacus_prepare_particles: bad arguments");
    if (nh > 0 && (!hmask || !pstart || !pnum || !pstart_new || !pnum_new)) return fail("abacus_prepare_particles: null halo column");
    if (!submask_in && !ntarget) return fail("abacus_prepare_particles: pass either the selection (submask) or the per-halo targets");
    if (want_ranks && (!N || !hpos || !hvel || !r25 || !r98 || !pos || !vel || !ranks || !ranksv || !ranksp || !ranksr || !ranksc))
        return fail("abacus_prepare_particles: the rank columns need positions, velocities, N, r25, r98 and five outputs");
    if (nh >= ((int64_t)1 << 31) || npart >= ((int64_t)1 << 31)) return fail("abacus_prepare_particles: slab too large for 32-bit indices");
    *n_sel = 0;
    if (nh == 0) return 0;
    Tmp tmp;
    unsigned char *d_hmask, *d_sub;
    long long *d_pstart, *d_pnum, *d_kept, *d_kstart, *d_flag, *d_slot;
    int *d_host, *d_bad, *d_nt = nullptr;
    const size_t np1 = (size_t)std::max<int64_t>(npart, 1);
    ABACUS_TRY(tmp.upload(&d_hmask, (const unsigned char *)hmask, (size_t)nh));
    ABACUS_TRY(tmp.upload(&d_pstart, (const long long *)pstart, (size_t)nh));
    ABACUS_TRY(tmp.upload(&d_pnum, (const long long *)pnum, (size_t)nh));
    ABACUS_TRY(tmp.alloc(&d_host, np1));
    ABACUS_TRY(tmp.alloc(&d_bad, 4));
    ABACUS_TRY(tmp.alloc(&d_sub, np1));
    HIP_TRY(hipMemsetAsync(d_host, 0xff, np1 * 4, stream()));
    HIP_TRY(hipMemsetAsync(d_bad, 0, 16, stream()));
    ABACUS_LAUNCH("prep_fill_host", prep_fill_host, dim3((unsigned int)std::min<int64_t>(ceil_div(nh, 4), 256 * 32)), dim3(256), 0,
                  d_pstart, d_pnum, d_hmask, nh, npart, d_host, d_bad);
    if (submask_in) {
        HIP_TRY(hipMemcpyAsync(d_sub, submask_in, (size_t)npart, hipMemcpyHostToDevice, stream()));
    } else {
        // Philox keys, one stable sort of (halo, key), the first ntarget of every halo's run
        ABACUS_TRY(tmp.upload(&d_nt, (const int *)ntarget, (size_t)nh));
        unsigned long long *k0, *k1;
        unsigned int *i0, *i1;
        long long *ccount, *cstart;
        ABACUS_TRY(tmp.alloc(&k0, np1));
        ABACUS_TRY(tmp.alloc(&k1, np1));
        ABACUS_TRY(tmp.alloc(&i0, np1));
        ABACUS_TRY(tmp.alloc(&i1, np1));
        ABACUS_TRY(tmp.alloc(&ccount, (size_t)nh));
        ABACUS_TRY(tmp.alloc(&cstart, (size_t)nh));
        ABACUS_LAUNCH("prep_keys", prep_keys, dim3(grid_for(npart)), dim3(256), 0, d_host, d_nt, d_pnum, npart,
                      (unsigned long long)seed, (long long)part_index0, k0, i0, d_sub);
        ABACUS_LAUNCH("prep_cand_count", prep_cand_count, dim3(grid_for(nh)), dim3(256), 0, d_hmask, d_pnum, d_nt, nh, ccount);
        ABACUS_TRY(exclusive_sum(ccount, cstart, nh, tmp));
        if (npart > 0) {
            size_t bytes = 0;
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k0, k1, i0, i1, (int)npart, 0, 64, stream()));
            void *ws;
            ABACUS_TRY(tmp.alloc((unsigned char **)&ws, bytes));
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(ws, bytes, k0, k1, i0, i1, (int)npart, 0, 64, stream()));
            ABACUS_LAUNCH("prep_pick", prep_pick, dim3(grid_for(npart)), dim3(256), 0, k1, i1, npart, cstart, d_nt, d_sub);
        }
    }
    ABACUS_TRY(tmp.alloc(&d_kept, (size_t)nh));
    ABACUS_TRY(tmp.alloc(&d_kstart, (size_t)nh));
    ABACUS_LAUNCH("prep_count", prep_count, dim3((unsigned int)std::min<int64_t>(ceil_div(nh, 4), 256 * 32)), dim3(256), 0, d_pstart,
                  d_pnum, d_hmask, d_sub, nh, d_kept);
    ABACUS_TRY(exclusive_sum(d_kept, d_kstart, nh, tmp));
    double *d_psn, *d_pnn;
    ABACUS_TRY(tmp.alloc(&d_psn, (size_t)nh));
    ABACUS_TRY(tmp.alloc(&d_pnn, (size_t)nh));
    ABACUS_LAUNCH("prep_halo_offsets", prep_halo_offsets, dim3(grid_for(nh)), dim3(256), 0, d_hmask, d_pnum, d_kept, d_kstart, nh, d_psn, d_pnn);
    HIP_TRY(hipMemcpyAsync(pstart_new, d_psn, (size_t)nh * 8, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipMemcpyAsync(pnum_new, d_pnn, (size_t)nh * 8, hipMemcpyDeviceToHost, stream()));
    long long last_start = 0, last_kept = 0;
    int bad = 0;
    HIP_TRY(hipMemcpyAsync(&last_start, d_kstart + (nh - 1), 8, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipMemcpyAsync(&last_kept, d_kept + (nh - 1), 8, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    if (bad) return fail("abacus_prepare_particles: a halo's [npstartA, npstartA + npoutA) lies outside the particle array");
    const int64_t nsel = last_start + last_kept;
    *n_sel = nsel;
    if (submask_out && npart > 0) HIP_TRY(hipMemcpyAsync(submask_out, d_sub, (size_t)npart, hipMemcpyDeviceToHost, stream()));
    if (nsel > cap_sel || !sel_idx) {      // size query (the count depends on the draw): the caller allocates and calls again
        HIP_TRY(hipStreamSynchronize(stream()));
        return 0;
    }
    if (nsel == 0) {
        HIP_TRY(hipStreamSynchronize(stream()));
        return 0;
    }
    long long *d_sidx, *d_shost;
    double *d_snp;
    ABACUS_TRY(tmp.alloc(&d_flag, np1));
    ABACUS_TRY(tmp.alloc(&d_slot, np1));
    ABACUS_TRY(tmp.alloc(&d_sidx, (size_t)nsel));
    ABACUS_TRY(tmp.alloc(&d_shost, (size_t)nsel));
    ABACUS_TRY(tmp.alloc(&d_snp, (size_t)nsel));
    ABACUS_LAUNCH("prep_flags", prep_flags, dim3(grid_for(npart)), dim3(256), 0, d_sub, d_host, npart, d_flag);
    ABACUS_TRY(exclusive_sum(d_flag, d_slot, npart, tmp));
    ABACUS_LAUNCH("prep_emit", prep_emit, dim3(grid_for(npart)), dim3(256), 0, d_flag, d_slot, d_host, d_kept, npart, d_sidx, d_shost, d_snp);
    HIP_TRY(hipMemcpyAsync(sel_idx, d_sidx, (size_t)nsel * 8, hipMemcpyDeviceToHost, stream()));
    if (sel_host) HIP_TRY(hipMemcpyAsync(sel_host, d_shost, (size_t)nsel * 8, hipMemcpyDeviceToHost, stream()));
    if (sel_np) HIP_TRY(hipMemcpyAsync(sel_np, d_snp, (size_t)nsel * 8, hipMemcpyDeviceToHost, stream()));
    if (want_ranks) {
        RankArgs A;
        float *d_pos, *d_vel, *d_hpos, *d_hvel, *d_r25, *d_r98;
        unsigned int *d_N;
        double *d_r[5];
        int *d_work, *d_cnt;
        ABACUS_TRY(tmp.upload(&d_pos, pos, (size_t)npart * 3));
        ABACUS_TRY(tmp.upload(&d_vel, vel, (size_t)npart * 3));
        ABACUS_TRY(tmp.upload(&d_hpos, hpos, (size_t)nh * 3));
        ABACUS_TRY(tmp.upload(&d_hvel, hvel, (size_t)nh * 3));
        ABACUS_TRY(tmp.upload(&d_r25, r25, (size_t)nh));
        ABACUS_TRY(tmp.upload(&d_r98, r98, (size_t)nh));
        ABACUS_TRY(tmp.upload(&d_N, (const unsigned int *)N, (size_t)nh));
        for (int c = 0; c < 5; c++) ABACUS_TRY(tmp.alloc(&d_r[c], (size_t)nsel));
        ABACUS_TRY(tmp.alloc(&d_work, (size_t)nh));
        ABACUS_TRY(tmp.alloc(&d_cnt, 4));
        HIP_TRY(hipMemsetAsync(d_cnt, 0, 16, stream()));
        ABACUS_LAUNCH("prep_rank_work", prep_rank_work, dim3(grid_for(nh)), dim3(256), 0, d_kept, nh, d_work, d_cnt, d_cnt + 1);
        ABACUS_LAUNCH("prep_rank_single", prep_rank_single, dim3(grid_for(nh)), dim3(256), 0, d_kept, d_kstart, nh, d_r[0], d_r[1],
                      d_r[2], d_r[3], d_r[4]);
        int cnt[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(cnt, d_cnt, 8, hipMemcpyDeviceToHost, stream()));
        HIP_TRY(hipStreamSynchronize(stream()));
        if (cnt[0] > 0) {
            const size_t lds = (size_t)((3 * (size_t)cnt[1] * 4 + 15) / 16) * 16 + (size_t)5 * cnt[1] * 8;
            if (lds > 160 * 1024) return fail("abacus_prepare_particles: %d kept particles in one halo exceed the rank kernel's LDS", cnt[1]);
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(prep_ranks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            A.pos = d_pos, A.vel = d_vel, A.hpos = d_hpos, A.hvel = d_hvel, A.N = d_N, A.r25 = d_r25, A.r98 = d_r98;
            A.pstart = d_pstart, A.pnum = d_pnum, A.kept = d_kept, A.kstart = d_kstart, A.sel_idx = d_sidx;
            A.Mpart = Mpart, A.h = h;
            A.ranks = d_r[0], A.ranksv = d_r[1], A.ranksp = d_r[2], A.ranksr = d_r[3], A.ranksc = d_r[4];
            A.work = d_work, A.kmax = cnt[1];
            ABACUS_LAUNCH("prep_ranks", prep_ranks, dim3((unsigned int)std::min(cnt[0], 256 * 8)), dim3(256), lds, A, cnt[0]);
        }
        double *dst[5] = {ranks, ranksv, ranksp, ranksr, ranksc};
        for (int c = 0; c < 5; c++) HIP_TRY(hipMemcpyAsync(dst[c], d_r[c], (size_t)nsel * 8, hipMemcpyDeviceToHost, stream()));
    }
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_prepare_randoms(int64_t n, const int64_t *index, int64_t index0, uint64_t seed, int stream_id, const double *scale,
                            double *randoms, double *randoms_exp, double *randoms_gaus) {
    ABACUS_ENTER();
    if (n < 0 || (n > 0 && !randoms)) return fail("abacus_prepare_randoms: null output");
    if ((randoms_exp || randoms_gaus) && !(scale && randoms_exp && randoms_gaus))
        return fail("abacus_prepare_randoms: the halo columns need the scale and both vector outputs");
    if (randoms_exp ? stream_id != 4 : (stream_id < 5 || stream_id > 255))
        return fail("abacus_prepare_randoms: stream %d (4 = the halo columns, 5 .. 255 = one uniform per object)", stream_id);
    if (n == 0) return 0;
    Tmp tmp;
    long long *d_idx = nullptr;
    double *d_sc = nullptr, *d_r, *d_e = nullptr, *d_g = nullptr;
    if (index) ABACUS_TRY(tmp.upload(&d_idx, (const long long *)index, (size_t)n));
    ABACUS_TRY(tmp.alloc(&d_r, (size_t)n));
    const uint2 key = make_uint2((unsigned int)seed, (unsigned int)(seed >> 32));
    if (randoms_exp) {
        ABACUS_TRY(tmp.upload(&d_sc, scale, (size_t)n));
        ABACUS_TRY(tmp.alloc(&d_e, (size_t)3 * n));
        ABACUS_TRY(tmp.alloc(&d_g, (size_t)3 * n));
        ABACUS_LAUNCH("prep_halo_randoms", prep_halo_randoms, dim3(grid_for(n)), dim3(256), 0, n, d_idx, (long long)index0, key, d_sc, d_r,
                      d_e, d_g);
        HIP_TRY(hipMemcpyAsync(randoms_exp, d_e, (size_t)n * 24, hipMemcpyDeviceToHost, stream()));
        HIP_TRY(hipMemcpyAsync(randoms_gaus, d_g, (size_t)n * 24, hipMemcpyDeviceToHost, stream()));
    } else {
        ABACUS_LAUNCH("prep_part_randoms", prep_part_randoms, dim3(grid_for(n)), dim3(256), 0, n, d_idx, (long long)index0, key,
                      (unsigned int)stream_id, d_r);
    }
    HIP_TRY(hipMemcpyAsync(randoms, d_r, (size_t)n * 8, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

// ---- one slab, device-resident from the inputs to the two tables (hod/prepare_sim.py:296-1052 between loader and writer) --------
// Everything abacusutils_amd/hod/prepare_sim.py::prepare_slab_arrays does with rng = <seed> - the halo mask, the down-sampling
// factors, the mass-bin rank of the concentration, the per-halo particle selection, the new offsets, the five satellite rank columns,
// the kept rows of every column of both tables and their random columns - without a column crossing PCIe twice: inputs are used in
// place when they are device pointers (the reader's unpack kernels leave pos / vel in HBM) or uploaded once, the tables stay in HBM
// until abacus_prepare_slab_fetch copies each column out (one copy per column, into the caller's - ideally page-locked - arrays).
// Same Philox streams as the column-by-column path (6: halo mask, 3: selection keys, 4: halo randoms, 5: particle randoms), so
// both paths produce identical tables (tests/test_prepare_gpu.py).
int abacus_prepare_slab(const abacus_prepare_slab_args *a, int64_t *n_halo_kept, int64_t *n_part_kept, uint8_t *mask_out) {
    ABACUS_ENTER();
    if (!a || !n_halo_kept || !n_part_kept) return fail("abacus_prepare_slab: null argument");
    const int64_t nh = a->nh, npart = a->npart;
    if (nh < 0 || npart < 0) return fail("abacus_prepare_slab: negative size");
    if (nh >= ((int64_t)1 << 31) || npart >= ((int64_t)1 << 31)) return fail("abacus_prepare_slab: slab too large for 32-bit indices");
    if (nh > 0 && (!a->N || !a->x || !a->v || !a->r25 || !a->r90 || !a->r98 || !a->npstartA || !a->npoutA || !a->id || !a->sigmav))
        return fail("abacus_prepare_slab: null halo column");
    if (npart > 0 && (!a->pos || !a->vel)) return fail("abacus_prepare_slab: null particle column");
    if (a->n_edges != 0 && (a->n_edges < 2 || a->n_edges > 60000 || !a->mbins)) return fail("abacus_prepare_slab: %d mass-bin edges", a->n_edges);
    for (int b = 0; b + 1 < a->n_edges; b++)
        if (!(a->mbins[b + 1] > a->mbins[b])) return fail("abacus_prepare_slab: the mass bin edges must increase");
    g_slab.clear();
    *n_halo_kept = *n_part_kept = 0;
    if (nh == 0) return 0;
    g_slab.tmp = new Tmp();
    Tmp &t = *g_slab.tmp;
    Tmp &w = t;       // (work arrays too stay until the fetch: the rank kernel reads them on a stream of its own)
    const size_t np1 = (size_t)std::max<int64_t>(npart, 1);
    const uint2 key = make_uint2((unsigned int)a->seed, (unsigned int)(a->seed >> 32));
    // ---- inputs
    const unsigned int *dN;
    const float *dx, *dv, *dr25, *dr90, *dr98, *dsig, *dpos, *dvel;
    const long long *dps, *dpn, *did;
    ABACUS_TRY(stage_col(w, (const unsigned int *)a->N, (size_t)nh, &dN));
    ABACUS_TRY(stage_col(w, a->x, (size_t)nh * 3, &dx));
    ABACUS_TRY(stage_col(w, a->v, (size_t)nh * 3, &dv));
    ABACUS_TRY(stage_col(w, a->r25, (size_t)nh, &dr25));
    ABACUS_TRY(stage_col(w, a->r90, (size_t)nh, &dr90));
    ABACUS_TRY(stage_col(w, a->r98, (size_t)nh, &dr98));
    ABACUS_TRY(stage_col(w, a->sigmav, (size_t)nh, &dsig));
    ABACUS_TRY(stage_col(w, (const long long *)a->npstartA, (size_t)nh, &dps));
    ABACUS_TRY(stage_col(w, (const long long *)a->npoutA, (size_t)nh, &dpn));
    ABACUS_TRY(stage_col(w, (const long long *)a->id, (size_t)nh, &did));
    ABACUS_TRY(stage_col(w, a->pos, (size_t)npart * 3, &dpos));
    ABACUS_TRY(stage_col(w, a->vel, (size_t)npart * 3, &dvel));
    // ---- halos: mask draws (stream 6), kept fraction, targets; masses and the ranked concentration
    double *d_u, *d_p, *d_mass, *d_dval, *d_deltac, *d_fenv, *d_shear;
    unsigned char *d_hmask;
    int *d_nt;
    ABACUS_TRY(w.alloc(&d_u, (size_t)nh));
    ABACUS_TRY(w.alloc(&d_p, (size_t)nh));
    ABACUS_TRY(w.alloc(&d_hmask, (size_t)nh));
    ABACUS_TRY(w.alloc(&d_nt, (size_t)nh));
    ABACUS_TRY(w.alloc(&d_mass, (size_t)nh));
    ABACUS_TRY(w.alloc(&d_dval, (size_t)nh));
    ABACUS_TRY(w.alloc(&d_deltac, (size_t)nh));
    ABACUS_TRY(w.alloc(&d_fenv, (size_t)nh));
    ABACUS_TRY(w.alloc(&d_shear, (size_t)nh));
    ABACUS_LAUNCH("prep_part_randoms", prep_part_randoms, dim3(grid_for(nh)), dim3(256), 0, nh, (const long long *)nullptr, (long long)a->halo_index0,
                  key, 6u, d_u);
    ABACUS_LAUNCH("prep_halo_factors", prep_halo_factors, dim3(grid_for(nh)), dim3(256), 0, dN, (const double *)nullptr, nh, a->Mpart, a->MT,
                  (const double *)d_u, dpn, d_p, d_hmask, d_nt);
    ABACUS_LAUNCH("prep_mass_deltac", prep_mass_deltac, dim3(grid_for(nh)), dim3(256), 0, dN, dr25, dr98, nh, a->Mpart, d_mass, d_dval);
    if (a->n_edges) {
        double *d_edges;
        ABACUS_TRY(w.upload(&d_edges, a->mbins, (size_t)a->n_edges));
        ABACUS_TRY(fenv_rank_device(d_mass, d_dval, nh, d_edges, a->n_edges, d_deltac));
    } else HIP_TRY(hipMemsetAsync(d_deltac, 0, (size_t)nh * 8, stream()));
    if (a->fenv_rank) HIP_TRY(hipMemcpyAsync(d_fenv, a->fenv_rank, (size_t)nh * 8, hipMemcpyDefault, stream()));
    else HIP_TRY(hipMemsetAsync(d_fenv, 0, (size_t)nh * 8, stream()));
    if (a->shear_rank) HIP_TRY(hipMemcpyAsync(d_shear, a->shear_rank, (size_t)nh * 8, hipMemcpyDefault, stream()));
    else HIP_TRY(hipMemsetAsync(d_shear, 0, (size_t)nh * 8, stream()));
    // ---- particle selection: Philox keys, one stable sort of (halo, key), the first ntarget of every halo's run
    unsigned char *d_sub;
    long long *d_kept, *d_kstart;
    int *d_host, *d_bad;
    ABACUS_TRY(w.alloc(&d_host, np1));
    ABACUS_TRY(w.alloc(&d_bad, 4));
    ABACUS_TRY(w.alloc(&d_sub, np1));
    HIP_TRY(hipMemsetAsync(d_host, 0xff, np1 * 4, stream()));
    HIP_TRY(hipMemsetAsync(d_bad, 0, 16, stream()));
    ABACUS_LAUNCH("prep_fill_host", prep_fill_host, dim3((unsigned int)std::min<int64_t>(ceil_div(nh, 4), 256 * 32)), dim3(256), 0, dps, dpn,
                  (const unsigned char *)d_hmask, nh, npart, d_host, d_bad);
    {
        unsigned long long *k0, *k1;
        unsigned int *i0, *i1;
        long long *ccount, *cstart;
        ABACUS_TRY(w.alloc(&k0, np1));
        ABACUS_TRY(w.alloc(&k1, np1));
        ABACUS_TRY(w.alloc(&i0, np1));
        ABACUS_TRY(w.alloc(&i1, np1));
        ABACUS_TRY(w.alloc(&ccount, (size_t)nh));
        ABACUS_TRY(w.alloc(&cstart, (size_t)nh));
        ABACUS_LAUNCH("prep_keys", prep_keys, dim3(grid_for(npart)), dim3(256), 0, (const int *)d_host, (const int *)d_nt, dpn, npart,
                      (unsigned long long)a->seed, (long long)a->part_index0, k0, i0, d_sub);
        ABACUS_LAUNCH("prep_cand_count", prep_cand_count, dim3(grid_for(nh)), dim3(256), 0, (const unsigned char *)d_hmask, dpn, (const int *)d_nt, nh, ccount);
        ABACUS_TRY(exclusive_sum(ccount, cstart, nh, w));
        if (npart > 0) {
            size_t bytes = 0;
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k0, k1, i0, i1, (int)npart, 0, 64, stream()));
            void *ws;
            ABACUS_TRY(w.alloc((unsigned char **)&ws, bytes));
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(ws, bytes, k0, k1, i0, i1, (int)npart, 0, 64, stream()));
            ABACUS_LAUNCH("prep_pick", prep_pick, dim3(grid_for(npart)), dim3(256), 0, (const unsigned long long *)k1, (const unsigned int *)i1, npart,
                          (const long long *)cstart, (const int *)d_nt, d_sub);
        }
    }
    ABACUS_TRY(w.alloc(&d_kept, (size_t)nh));
    ABACUS_TRY(w.alloc(&d_kstart, (size_t)nh));
    ABACUS_LAUNCH("prep_count", prep_count, dim3((unsigned int)std::min<int64_t>(ceil_div(nh, 4), 256 * 32)), dim3(256), 0, dps, dpn,
                  (const unsigned char *)d_hmask, (const unsigned char *)d_sub, nh, d_kept);
    ABACUS_TRY(exclusive_sum(d_kept, d_kstart, nh, w));
    double *d_psn, *d_pnn;
    ABACUS_TRY(w.alloc(&d_psn, (size_t)nh));
    ABACUS_TRY(w.alloc(&d_pnn, (size_t)nh));
    ABACUS_LAUNCH("prep_halo_offsets", prep_halo_offsets, dim3(grid_for(nh)), dim3(256), 0, (const unsigned char *)d_hmask, dpn, (const long long *)d_kept,
                  (const long long *)d_kstart, nh, d_psn, d_pnn);
    // ---- kept halos: flags -> slots
    long long *d_hflag, *d_hslot;
    ABACUS_TRY(w.alloc(&d_hflag, (size_t)nh));
    ABACUS_TRY(w.alloc(&d_hslot, (size_t)nh));
    ABACUS_LAUNCH("prep_flags", prep_flag_u8, dim3(grid_for(nh)), dim3(256), 0, (const unsigned char *)d_hmask, nh, d_hflag);
    ABACUS_TRY(exclusive_sum(d_hflag, d_hslot, nh, w));
    // halos with two or more kept particles are the rank kernel's work list; its size and the largest halo come back with the counts
    int *d_work = nullptr, *d_cnt = nullptr;
    int cnt[2] = {0, 0};
    if (a->want_ranks) {
        ABACUS_TRY(w.alloc(&d_work, (size_t)nh));
        ABACUS_TRY(w.alloc(&d_cnt, 4));
        HIP_TRY(hipMemsetAsync(d_cnt, 0, 16, stream()));
        ABACUS_LAUNCH("prep_rank_work", prep_rank_work, dim3(grid_for(nh)), dim3(256), 0, (const long long *)d_kept, nh, d_work, d_cnt, d_cnt + 1);
        HIP_TRY(hipMemcpyAsync(cnt, d_cnt, 8, hipMemcpyDeviceToHost, stream()));
    }
    long long tail[4] = {0, 0, 0, 0};
    int bad = 0;
    HIP_TRY(hipMemcpyAsync(&tail[0], d_kstart + (nh - 1), 8, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipMemcpyAsync(&tail[1], d_kept + (nh - 1), 8, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipMemcpyAsync(&tail[2], d_hslot + (nh - 1), 8, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipMemcpyAsync(&tail[3], d_hflag + (nh - 1), 8, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, stream()));
    if (mask_out) HIP_TRY(hipMemcpyAsync(mask_out, d_hmask, (size_t)nh, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    if (bad) {
        g_slab.clear();
        return fail("abacus_prepare_slab: a halo's [npstartA, npstartA + npoutA) lies outside the particle array");
    }
    const int64_t ns = tail[0] + tail[1], nk = tail[2] + tail[3];
    g_slab.nk = nk, g_slab.ns = ns, g_slab.want_ranks = a->want_ranks;
    *n_halo_kept = nk, *n_part_kept = ns;
    // ---- the particle table
    long long *d_sidx = nullptr, *d_shost = nullptr;
    if (ns > 0) {
        long long *d_flag, *d_slot;
        double *d_snp;
        ABACUS_TRY(w.alloc(&d_flag, np1));
        ABACUS_TRY(w.alloc(&d_slot, np1));
        ABACUS_TRY(w.alloc(&d_sidx, (size_t)ns));
        ABACUS_TRY(w.alloc(&d_shost, (size_t)ns));
        ABACUS_TRY(t.alloc(&d_snp, (size_t)ns));
        ABACUS_LAUNCH("prep_flags", prep_flags, dim3(grid_for(npart)), dim3(256), 0, (const unsigned char *)d_sub, (const int *)d_host, npart, d_flag);
        ABACUS_TRY(exclusive_sum(d_flag, d_slot, npart, w));
        ABACUS_LAUNCH("prep_emit", prep_emit, dim3(grid_for(npart)), dim3(256), 0, (const long long *)d_flag, (const long long *)d_slot, (const int *)d_host,
                      (const long long *)d_kept, npart, d_sidx, d_shost, d_snp);
        auto pout = [&](int c, size_t bytes, void **dev) {
            unsigned char *q;
            int rc = t.alloc(&q, bytes);
            g_slab.pcol[c] = q, g_slab.pbytes[c] = bytes, *dev = q;
            return rc;
        };
        void *o;
        ABACUS_TRY(pout(ABACUS_PREP_P_POS, (size_t)ns * 12, &o));
        ABACUS_TRY(gather_w<3>(dpos, d_sidx, ns, o));
        ABACUS_TRY(pout(ABACUS_PREP_P_VEL, (size_t)ns * 12, &o));
        ABACUS_TRY(gather_w<3>(dvel, d_sidx, ns, o));
        if (a->want_ranks) {
            double *d_r[5];
            for (int c = 0; c < 5; c++) {
                ABACUS_TRY(pout(ABACUS_PREP_P_RANKS + c, (size_t)ns * 8, &o));
                d_r[c] = (double *)o;
            }
            const size_t lds = (size_t)((3 * (size_t)cnt[1] * 4 + 15) / 16) * 16 + (size_t)5 * cnt[1] * 8;
            if (cnt[0] > 0 && lds > 160 * 1024) {
                g_slab.clear();
                return fail("abacus_prepare_slab: %d kept particles in one halo exceed the rank kernel's LDS", cnt[1]);
            }
            RankArgs A;
            A.pos = dpos, A.vel = dvel, A.hpos = dx, A.hvel = dv, A.N = dN, A.r25 = dr25, A.r98 = dr98;
            A.pstart = dps, A.pnum = dpn, A.kept = d_kept, A.kstart = d_kstart, A.sel_idx = d_sidx;
            A.Mpart = a->Mpart, A.h = a->h;
            A.ranks = d_r[0], A.ranksv = d_r[1], A.ranksp = d_r[2], A.ranksr = d_r[3], A.ranksc = d_r[4];
            A.work = d_work, A.kmax = cnt[1];
            if (cnt[0] > 0) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(prep_ranks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (prof_enabled()) {
                ABACUS_LAUNCH("prep_rank_single", prep_rank_single, dim3(grid_for(nh)), dim3(256), 0, (const long long *)d_kept, (const long long *)d_kstart, nh,
                              d_r[0], d_r[1], d_r[2], d_r[3], d_r[4]);
                if (cnt[0] > 0) ABACUS_LAUNCH("prep_ranks", prep_ranks, dim3((unsigned int)std::min(cnt[0], 256 * 8)), dim3(256), lds, A, cnt[0]);
            } else {
                if (!g_slab_side) {
                    HIP_TRY(hipStreamCreateWithFlags(&g_slab_side, hipStreamNonBlocking));
                    HIP_TRY(hipEventCreateWithFlags(&g_slab_ready, hipEventDisableTiming));
                    HIP_TRY(hipEventCreateWithFlags(&g_slab_ranked, hipEventDisableTiming));
                }
                HIP_TRY(hipEventRecord(g_slab_ready, stream()));          // kept list, offsets, inputs: all written on the library stream
                HIP_TRY(hipStreamWaitEvent(g_slab_side, g_slab_ready, 0));
                prep_rank_single<<<dim3(grid_for(nh)), dim3(256), 0, g_slab_side>>>((const long long *)d_kept, (const long long *)d_kstart, nh, d_r[0], d_r[1],
                                                                                 d_r[2], d_r[3], d_r[4]);
                if (cnt[0] > 0) prep_ranks<<<dim3((unsigned int)std::min(cnt[0], 256 * 8)), dim3(256), lds, g_slab_side>>>(A, cnt[0]);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventRecord(g_slab_ranked, g_slab_side));
                g_slab.ranks_on_side = true;
            }
        }
        ABACUS_TRY(pout(ABACUS_PREP_P_DOWNSAMPLE, (size_t)ns * 8, &o));
        ABACUS_TRY(gather_w<2>(d_p, d_shost, ns, o));
        ABACUS_TRY(pout(ABACUS_PREP_P_HALO_VEL, (size_t)ns * 24, &o));
        ABACUS_LAUNCH("prep_gather", prep_gather_f3d, dim3(grid_for(ns * 3)), dim3(256), 0, dv, (const long long *)d_shost, ns, (double *)o);
        ABACUS_TRY(pout(ABACUS_PREP_P_HALO_MASS, (size_t)ns * 8, &o));
        ABACUS_TRY(gather_w<2>(d_mass, d_shost, ns, o));
        g_slab.pcol[ABACUS_PREP_P_NP] = d_snp, g_slab.pbytes[ABACUS_PREP_P_NP] = (size_t)ns * 8;
        ABACUS_TRY(pout(ABACUS_PREP_P_HALO_ID, (size_t)ns * 8, &o));
        ABACUS_TRY(gather_w<2>(did, d_shost, ns, o));
        ABACUS_TRY(pout(ABACUS_PREP_P_RANDOMS, (size_t)ns * 8, &o));
        ABACUS_LAUNCH("prep_part_randoms", prep_part_randoms, dim3(grid_for(ns)), dim3(256), 0, ns, (const long long *)d_sidx, (long long)a->part_index0, key,
                      5u, (double *)o);
        ABACUS_TRY(pout(ABACUS_PREP_P_DELTAC, (size_t)ns * 8, &o));
        ABACUS_TRY(gather_w<2>(d_deltac, d_shost, ns, o));
        ABACUS_TRY(pout(ABACUS_PREP_P_FENV, (size_t)ns * 8, &o));
        ABACUS_TRY(gather_w<2>(d_fenv, d_shost, ns, o));
        ABACUS_TRY(pout(ABACUS_PREP_P_SHEAR, (size_t)ns * 8, &o));
        ABACUS_TRY(gather_w<2>(d_shear, d_shost, ns, o));
    }
    // ---- the halo table
    if (nk > 0) {
        long long *d_kidx;
        ABACUS_TRY(w.alloc(&d_kidx, (size_t)nk));
        ABACUS_LAUNCH("prep_compact", prep_compact, dim3(grid_for(nh)), dim3(256), 0, (const long long *)d_hflag, (const long long *)d_hslot, nh, d_kidx);
        auto hout = [&](int c, size_t bytes, void **dev) {
            unsigned char *q;
            int rc = t.alloc(&q, bytes);
            g_slab.hcol[c] = q, g_slab.hbytes[c] = bytes, *dev = q;
            return rc;
        };
        void *o;
        ABACUS_TRY(hout(ABACUS_PREP_H_N, (size_t)nk * 4, &o));
        ABACUS_TRY(gather_w<1>(dN, d_kidx, nk, o));
        ABACUS_TRY(hout(ABACUS_PREP_H_X, (size_t)nk * 12, &o));
        ABACUS_TRY(gather_w<3>(dx, d_kidx, nk, o));
        ABACUS_TRY(hout(ABACUS_PREP_H_V, (size_t)nk * 12, &o));
        ABACUS_TRY(gather_w<3>(dv, d_kidx, nk, o));
        ABACUS_TRY(hout(ABACUS_PREP_H_R25, (size_t)nk * 4, &o));
        ABACUS_TRY(gather_w<1>(dr25, d_kidx, nk, o));
        ABACUS_TRY(hout(ABACUS_PREP_H_R90, (size_t)nk * 4, &o));
        ABACUS_TRY(gather_w<1>(dr90, d_kidx, nk, o));
        ABACUS_TRY(hout(ABACUS_PREP_H_R98, (size_t)nk * 4, &o));
        ABACUS_TRY(gather_w<1>(dr98, d_kidx, nk, o));
        ABACUS_TRY(hout(ABACUS_PREP_H_NPSTART, (size_t)nk * 8, &o));
        ABACUS_TRY(gather_w<2>(d_psn, d_kidx, nk, o));
        ABACUS_TRY(hout(ABACUS_PREP_H_NPOUT, (size_t)nk * 8, &o));
        ABACUS_TRY(gather_w<2>(d_pnn, d_kidx, nk, o));
        ABACUS_TRY(hout(ABACUS_PREP_H_ID, (size_t)nk * 8, &o));
        ABACUS_TRY(gather_w<2>(did, d_kidx, nk, o));
        ABACUS_TRY(hout(ABACUS_PREP_H_SIGMAV, (size_t)nk * 4, &o));
        ABACUS_TRY(gather_w<1>(dsig, d_kidx, nk, o));
        ABACUS_TRY(hout(ABACUS_PREP_H_MASK, (size_t)nk, &o));
        HIP_TRY(hipMemsetAsync(o, 1, (size_t)nk, stream()));
        ABACUS_TRY(hout(ABACUS_PREP_H_MULTI, (size_t)nk * 8, &o));
        ABACUS_LAUNCH("prep_gather", prep_gather_recip, dim3(grid_for(nk)), dim3(256), 0, (const double *)d_p, (const long long *)d_kidx, nk, (double *)o);
        ABACUS_TRY(hout(ABACUS_PREP_H_FENV, (size_t)nk * 8, &o));
        ABACUS_TRY(gather_w<2>(d_fenv, d_kidx, nk, o));
        ABACUS_TRY(hout(ABACUS_PREP_H_DELTAC, (size_t)nk * 8, &o));
        ABACUS_TRY(gather_w<2>(d_deltac, d_kidx, nk, o));
        ABACUS_TRY(hout(ABACUS_PREP_H_SHEAR, (size_t)nk * 8, &o));
        ABACUS_TRY(gather_w<2>(d_shear, d_kidx, nk, o));
        double *d_scale, *d_rnd, *d_rexp, *d_rgaus;
        ABACUS_TRY(w.alloc(&d_scale, (size_t)nk));
        ABACUS_LAUNCH("prep_gather", prep_gather_scale, dim3(grid_for(nk)), dim3(256), 0, dsig, (const long long *)d_kidx, nk, d_scale);
        ABACUS_TRY(hout(ABACUS_PREP_H_RANDOMS, (size_t)nk * 8, &o));
        d_rnd = (double *)o;
        ABACUS_TRY(hout(ABACUS_PREP_H_REXP, (size_t)nk * 24, &o));
        d_rexp = (double *)o;
        ABACUS_TRY(hout(ABACUS_PREP_H_RGAUS, (size_t)nk * 24, &o));
        d_rgaus = (double *)o;
        ABACUS_LAUNCH("prep_halo_randoms", prep_halo_randoms, dim3(grid_for(nk)), dim3(256), 0, nk, (const long long *)d_kidx, (long long)a->halo_index0, key,
                      (const double *)d_scale, d_rnd, d_rexp, d_rgaus);
    }
    // the work arrays go back to the pool behind the kernels above (one stream); the table columns stay until the fetch
    return 0;
}

// copies the columns of the tables the last abacus_prepare_slab left in HBM into the caller's arrays (NULL = not wanted; sizes as
// the two counts of that call say) and releases them.  halo_cols[ABACUS_PREP_HALO_COLS], part_cols[ABACUS_PREP_PART_COLS].
int abacus_prepare_slab_fetch(void *const *halo_cols, void *const *part_cols) {
    ABACUS_ENTER();
    if (!g_slab.tmp) return fail("abacus_prepare_slab_fetch: no prepared slab (call abacus_prepare_slab first)");
    for (int c = 0; c < ABACUS_PREP_HALO_COLS; c++)
        if (halo_cols && halo_cols[c] && g_slab.hcol[c] && g_slab.hbytes[c])
            HIP_TRY(hipMemcpyAsync(halo_cols[c], g_slab.hcol[c], g_slab.hbytes[c], hipMemcpyDeviceToHost, stream()));
    auto is_rank = [](int c) { return c >= ABACUS_PREP_P_RANKS && c < ABACUS_PREP_P_RANKS + 5; };
    for (int pass = 0; pass < 2; pass++) {        // the rank columns last: their kernel may still be running beside these copies
        if (pass == 1 && g_slab.ranks_on_side) HIP_TRY(hipStreamWaitEvent(stream(), g_slab_ranked, 0));
        for (int c = 0; c < ABACUS_PREP_PART_COLS; c++)
            if (is_rank(c) == (pass == 1) && part_cols && part_cols[c] && g_slab.pcol[c] && g_slab.pbytes[c])
                HIP_TRY(hipMemcpyAsync(part_cols[c], g_slab.pcol[c], g_slab.pbytes[c], hipMemcpyDeviceToHost, stream()));
    }
    if (g_slab.ranks_on_side) HIP_TRY(hipStreamWaitEvent(stream(), g_slab_ranked, 0));   // (also when no rank column was asked for)
    HIP_TRY(hipStreamSynchronize(stream()));
    g_slab.clear();
    return 0;
}

}  // extern "C"
