// Power-spectrum estimator on MI355X (gfx950): replaces the calc_power chain of the reference
// (abacusnbody/analysis/power_spectrum.py): get_field :808-857 (deposit + normalize_field :860-901),
// scipy.fft.rfftn :980,986,1059 + _normalize :1073-1078, get_interlaced_field_fft / shift_field_fft :951-998,
// :904-948, the compensation divide :1063-1069, get_raw_power :707-727 and bin_kmu :150-300.
//
// Data layout in HBM: one float32 mesh per field in the in-place R2C layout (n, n, 2*(n/2+1)); after the FFT the
// same buffer is the complex64 half-spectrum (n, n, n/2+1).  Nothing else of mesh size is allocated (plus rocFFT's
// work area); the spectrum never goes back over PCIe in the fused path.
//
// Passes over mesh-sized data (algorithmic bytes 36*M non-interlaced):
//   tsc_tile_deposit  writes the mesh once, normalisation delta = rho*M/N - 1 fused into the tile flush (4M)
//   hipFFT/rocFFT R2C in place                                                              (~3 x (4M + 4M))
//   spectrum_bin      reads the half-spectrum once and fuses scale (1/M), interlacing combine, compensation,
//                     |delta_k|^2 (or the cross power) and the (k, mu) / multipole binning            (4M, 8M, 16M)
//
// spectrum_bin: persistent workgroups (one per CU).  Each stages a block of consecutive (kx, ky) rows as float32
// power values in LDS (coalesced HBM reads), then every thread walks a contiguous run of kz of one row exactly like
// the reference's inner loop (monotone bin search, :246-256) but accumulates the run in registers and only touches
// the workgroup's LDS histogram when the bin changes.  Histograms are float64 / integer; they are flushed to HBM
// with one atomic per non-empty bin per workgroup at the very end.  (The reference keeps float32 per-thread
// accumulators, :221-229; float64 sums are strictly more accurate and thread-count independent.)
#include <hipfft/hipfft.h>

#include <cmath>
#include <cstring>
#include <map>
#include <vector>

#include "../../include/abacus_hip.h"
#include "common.hpp"

using namespace abacus;

namespace abacus {
int tsc_deposit_f32(float *pos, int64_t n, const float *w, float *grid, int nmesh, int64_t zstride, double box,
                    double offset, int wrap, double norm, int cic);
int tsc_release_work();
}  // namespace abacus

namespace {

constexpr int MAX_POLES = 8;     // requested multipoles
constexpr int BIN_THREADS = 1024;

struct SpecArgs {
    int n, kzlen;
    int mode;                 // 0: raw fields (deltak API), 1: FFT output needing scale/interlace/compensation
    int interlaced, compensated, cross;
    float inv_size;           // f32(1/M)            (:1058)
    float half_inv_size;      // f32(0.5/M)          (:934)
    const float2 *a, *as, *b, *bs;   // field 1 (+ shifted), field 2 (+ shifted); b == nullptr -> auto power
    const float *W;           // (n,) window or nullptr
    const float2 *phase;      // (2n,) e^{i*pi*m/n}
};

__device__ __forceinline__ int fold(int i, int n) { return i < n / 2 ? i : i - n; }   // (:234,237,940-942)

// final delta_k of one field at (i, j, k): what get_field_fft returns (:1046-1070)
__device__ __forceinline__ float2 field_value(const SpecArgs &s, const float2 *f, const float2 *fs, int64_t idx,
                                              int i, int j, int k) {
    float2 v = f[idx];
    if (s.mode == 0) return v;
    if (s.interlaced) {
        // (delta_k + delta'_k * exp(i*(d/2)*(kx+ky+kz))) * f32(0.5/M); (d/2)*dk = pi/n, so the phase only depends
        // on m = i' + j' + k (mod 2n): taken from a table of exact angles
        int m = fold(i, s.n) + fold(j, s.n) + k;
        m %= 2 * s.n;
        if (m < 0) m += 2 * s.n;
        const float2 ph = s.phase[m];
        const float2 w = fs[idx];
        const float re = w.x * ph.x - w.y * ph.y, im = w.x * ph.y + w.y * ph.x;
        v.x = (v.x + re) * s.half_inv_size;
        v.y = (v.y + im) * s.half_inv_size;
    } else {
        v.x *= s.inv_size;
        v.y *= s.inv_size;
    }
    if (s.compensated) {
        const float w = (s.W[i] * s.W[j]) * s.W[k];   // (:1065-1069), NumPy divides complex by real as *(1/w)
        const float scl = 1.0f / w;
        v.x *= scl;
        v.y *= scl;
    }
    return v;
}

// in-place finalisation for abacus_field_fft (spectrum returned to the host)
__global__ void spectrum_apply(SpecArgs s, float2 *out) {
    const int64_t total = (int64_t)s.n * s.n * s.kzlen;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx % s.kzlen);
        const int64_t row = idx / s.kzlen;
        const int j = (int)(row % s.n), i = (int)(row / s.n);
        out[idx] = field_value(s, s.a, s.as, idx, i, j, k);
    }
}

struct BinArgs {
    int Nk, Nmu, Np;          // Np = number of requested poles with ell != 0 (ell = 0 comes from the wedges)
    int rows;                 // rows staged per tile
    int chunk;                // consecutive elements per thread (odd -> conflict-free LDS reads)
    const float *kedges2;     // (Nk+1) f32((kedges/dk)^2)  (:217)
    const float *muedges2;    // (Nmu+1) f32(muedges^2)     (:218)
    float polecoef[MAX_POLES][6];   // (2l+1) * P_l as a polynomial in mu^2: sum_m c[m] * (mu^2)^m
    unsigned long long *g_cnt;      // (Nk*Nmu)
    double *g_sum, *g_ksum;         // (Nk*Nmu)
    double *g_pole;                 // (Np*Nk)
};

// number of edges[1..N] strictly below v  ==  the bin the reference's `while v > edges[b+1]: b += 1` stops at
__device__ __forceinline__ int lower_bin(const float *edges, int N, float v) {
    int lo = 0, hi = N;   // answer in [lo, hi]
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (v > edges[mid + 1]) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(BIN_THREADS) void spectrum_bin(SpecArgs s, BinArgs b) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int nb = b.Nk * b.Nmu;
    // LDS carve-up: [sum f64 nb][ksum f64 nb][pole f64 Np*Nk][cnt u32 nb][kedges2 Nk+1][muedges2 Nmu+1][tile f32]
    double *h_sum = reinterpret_cast<double *>(smem);
    double *h_ksum = h_sum + nb;
    double *h_pole = h_ksum + nb;
    unsigned int *h_cnt = reinterpret_cast<unsigned int *>(h_pole + (size_t)b.Np * b.Nk);
    float *ke = reinterpret_cast<float *>(h_cnt + nb);
    float *me = ke + (b.Nk + 1);
    float *tile = me + (b.Nmu + 1);
    const int tid = threadIdx.x;
    for (int q = tid; q < nb; q += BIN_THREADS) {
        h_sum[q] = 0.0;
        h_ksum[q] = 0.0;
        h_cnt[q] = 0u;
    }
    for (int q = tid; q < b.Np * b.Nk; q += BIN_THREADS) h_pole[q] = 0.0;
    for (int q = tid; q <= b.Nk; q += BIN_THREADS) ke[q] = b.kedges2[q];
    for (int q = tid; q <= b.Nmu; q += BIN_THREADS) me[q] = b.muedges2[q];
    __syncthreads();
    const float klo = ke[0], khi = ke[b.Nk];
    const int n = s.n, kzlen = s.kzlen;
    const int64_t nrows = (int64_t)n * n;
    const int64_t ntiles = (nrows + b.rows - 1) / b.rows;
    const int lane = tid & 63, wave = tid >> 6, nwaves = BIN_THREADS / 64;

    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t row0 = t * b.rows;
        const int nr = (int)min((int64_t)b.rows, nrows - row0);
        // ---- stage: power of every mode of these rows -> LDS (coalesced reads along kz) ----
        for (int r = wave; r < nr; r += nwaves) {
            const int64_t row = row0 + r;
            const int j = (int)(row % n), i = (int)(row / n);
            for (int k = lane; k < kzlen; k += 64) {
                const int64_t idx = row * kzlen + k;
                const float2 va = field_value(s, s.a, s.as, idx, i, j, k);
                float p;
                if (s.cross) {
                    const float2 vb = field_value(s, s.b, s.bs, idx, i, j, k);
                    p = va.x * vb.x + va.y * vb.y;   // Re(conj(a) b)  (:724)
                } else {
                    p = va.x * va.x + va.y * va.y;   // |a|^2          (:726)
                }
                tile[r * kzlen + k] = p;
            }
        }
        __syncthreads();
        // ---- bin: thread walks `chunk` consecutive kz (may cross into the next row) ----
        const int total = nr * kzlen;
        int e = tid * b.chunk;
        const int e1 = min(e + b.chunk, total);
        if (e < e1) {
            int r = e / kzlen, k = e - r * kzlen;
            int cur = -1, bk = 0, bmu = 0;          // open run
            int cnt = 0;
            float sp = 0.f, sk = 0.f, spole[MAX_POLES];
#pragma unroll
            for (int q = 0; q < MAX_POLES; q++) spole[q] = 0.f;
            bool fresh = true, dead = false;
            long long r2 = 0;
            auto flush = [&]() {
                if (cnt) {
                    atomicAdd(&h_cnt[cur], (unsigned int)cnt);
                    atomicAdd(&h_sum[cur], (double)sp);
                    atomicAdd(&h_ksum[cur], (double)sk);
                    for (int q = 0; q < b.Np; q++) atomicAdd(&h_pole[q * b.Nk + bk], (double)spole[q]);
                }
                cnt = 0;
                sp = sk = 0.f;
#pragma unroll
                for (int q = 0; q < MAX_POLES; q++) spole[q] = 0.f;
            };
            for (; e < e1; e++, k++) {
                if (k == kzlen) {
                    k = 0;
                    r++;
                    fresh = true;
                }
                if (fresh) {
                    flush();
                    cur = -1;
                    const int64_t row = row0 + r;
                    const int jj = fold((int)(row % n), n), ii = fold((int)(row / n), n);
                    r2 = (long long)ii * ii + (long long)jj * jj;
                    dead = false;
                }
                if (dead) {
                    fresh = false;
                    continue;
                }
                const float kmag2 = (float)(r2 + (long long)k * k);   // dtype(i2 + j2 + k**2)   (:239)
                float mu2 = 0.f;
                if (kmag2 > 0.f) mu2 = (float)((long long)k * k) * (1.0f / kmag2);   // (:240-244)
                if (kmag2 < klo) {
                    fresh = false;   // `continue` (:246): bins are searched again when the row enters the range
                    cur = -1;
                    continue;
                }
                if (kmag2 >= khi) {  // `break` (:249): nothing further along kz can be in range
                    dead = true;
                    fresh = false;
                    continue;
                }
                int nbk, nbmu;
                if (cur < 0) {
                    nbk = lower_bin(ke, b.Nk - 1, kmag2);
                    nbmu = lower_bin(me, b.Nmu - 1, mu2);
                } else {
                    nbk = bk;
                    nbmu = bmu;
                    while (kmag2 > ke[nbk + 1]) nbk++;                           // (:252-253)
                    while (nbmu + 1 < b.Nmu && mu2 > me[nbmu + 1]) nbmu++;       // (:255-256)
                }
                const int nb_idx = nbk * b.Nmu + nbmu;
                if (nb_idx != cur) {
                    flush();
                    cur = nb_idx;
                    bk = nbk;
                    bmu = nbmu;
                }
                fresh = false;
                const float p = tile[e];
                const float wgt = k == 0 ? 1.f : 2.f;
                cnt += k == 0 ? 1 : 2;
                sp += wgt * p;
                sk += wgt * sqrtf(kmag2);
                for (int q = 0; q < b.Np; q++) {
                    const float *c = b.polecoef[q];
                    const float L = c[0] + mu2 * (c[1] + mu2 * (c[2] + mu2 * (c[3] + mu2 * (c[4] + mu2 * c[5]))));
                    spole[q] += wgt * p * L;
                }
            }
            flush();
        }
        __syncthreads();
    }
    // ---- flush the workgroup histogram ----
    for (int q = tid; q < nb; q += BIN_THREADS) {
        if (h_cnt[q]) {
            atomicAdd(&b.g_cnt[q], (unsigned long long)h_cnt[q]);
            atomicAdd(&b.g_sum[q], h_sum[q]);
            atomicAdd(&b.g_ksum[q], h_ksum[q]);
        }
    }
    for (int q = tid; q < b.Np * b.Nk; q += BIN_THREADS)
        if (h_pole[q] != 0.0) atomicAdd(&b.g_pole[q], h_pole[q]);
}

// ---- host side ------------------------------------------------------------------------------------------
struct PowerCtx {
    std::map<int, hipfftHandle> plans;
    DevBuf mesh[4];       // field1, field1 shifted, field2, field2 shifted
    DevBuf W, phase, edges, accum, pos, pos2, w, w2;
    int phase_n = 0;
};
PowerCtx g_ctx;

int fft_check(hipfftResult r, const char *what) {
    if (r != HIPFFT_SUCCESS) return fail("%s failed (hipfftResult %d)", what, (int)r);
    return 0;
}

int get_plan(int n, hipfftHandle *out) {
    auto it = g_ctx.plans.find(n);
    if (it == g_ctx.plans.end()) {
        hipfftHandle h;
        ABACUS_TRY(fft_check(hipfftPlan3d(&h, n, n, n, HIPFFT_R2C), "hipfftPlan3d"));
        it = g_ctx.plans.emplace(n, h).first;
    }
    ABACUS_TRY(fft_check(hipfftSetStream(it->second, stream()), "hipfftSetStream"));
    *out = it->second;
    return 0;
}

size_t mesh_bytes(int n) { return (size_t)n * n * (2 * (n / 2 + 1)) * sizeof(float); }

int ensure_phase(int n) {
    if (g_ctx.phase_n == n) return 0;
    std::vector<float2> h((size_t)2 * n);
    for (int m = 0; m < 2 * n; m++) {
        const double th = M_PI * (double)m / (double)n;
        h[m] = make_float2((float)cos(th), (float)sin(th));
    }
    ABACUS_TRY(g_ctx.phase.reserve(h.size() * sizeof(float2)));
    HIP_TRY(hipMemcpyAsync(g_ctx.phase.p, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    g_ctx.phase_n = n;
    return 0;
}

// deposit + FFT of one particle set into mesh slots [slot] (and [slot+1] when interlaced); device particle arrays
int field_fft_dev(float *pos, int64_t n, const float *w, double L, int nmesh, int paste, int interlaced, int slot) {
    if (n <= 0) return fail("power: empty particle set");
    hipfftHandle plan;
    ABACUS_TRY(get_plan(nmesh, &plan));
    const int64_t zstride = 2 * (nmesh / 2 + 1);
    const double M = (double)nmesh * nmesh * nmesh;
    const double norm = (double)(float)(M / (double)n);   // dtype(field.size / tot_weight), tot_weight = len(pos) (:856,894)
    const double d = L / nmesh;
    for (int s = 0; s < (interlaced ? 2 : 1); s++) {
        ABACUS_TRY(g_ctx.mesh[slot + s].reserve(mesh_bytes(nmesh)));
        float *mesh = g_ctx.mesh[slot + s].as<float>();
        // tsc_parallel wraps pos in place on the first call (tsc.py:171-173); the shifted deposit sees wrapped pos
        ABACUS_TRY(tsc_deposit_f32(pos, n, w, mesh, nmesh, zstride, L, s == 0 ? 0.0 : 0.5 * d, paste == 0, norm, paste));
        prof_begin("hipfft_r2c");
        hipfftResult r = hipfftExecR2C(plan, (hipfftReal *)mesh, (hipfftComplex *)mesh);
        prof_end("hipfft_r2c");
        ABACUS_TRY(fft_check(r, "hipfftExecR2C"));
    }
    return 0;
}

void fill_spec(SpecArgs &s, int nmesh, int mode, int interlaced, const float *W_dev, bool cross) {
    s.n = nmesh;
    s.kzlen = nmesh / 2 + 1;
    s.mode = mode;
    s.interlaced = interlaced;
    s.compensated = W_dev != nullptr;
    s.cross = cross;
    const double M = (double)nmesh * nmesh * nmesh;
    s.inv_size = (float)(1.0 / M);
    s.half_inv_size = (float)(0.5 / M);
    s.W = W_dev;
    s.phase = g_ctx.phase.as<float2>();
}

// coefficients of (2l+1) * P_l(mu) as a polynomial in x = mu^2 (even l <= 10): P_n (:121-147)
int pole_coefs(int l, float c[6]) {
    if (l < 0 || l > 10 || (l & 1)) return fail("power: multipole l=%d unsupported (even l <= 10, like P_n's tested range)", l);
    auto binom = [](int n, int k) {
        double r = 1;
        for (int i = 1; i <= k; i++) r = r * (n - k + i) / i;
        return r;
    };
    for (int m = 0; m < 6; m++) c[m] = 0.f;
    for (int k = 0; k <= l / 2; k++) {
        double f = binom(l, k) * binom(2 * l - 2 * k, l) * std::pow(0.5, l) * ((k & 1) ? -1.0 : 1.0) * (2 * l + 1);
        c[(l - 2 * k) / 2] = (float)f;   // mu^(l-2k) = x^((l-2k)/2)
    }
    return 0;
}

int run_bin(const SpecArgs &s, double Lbox, const double *kedges, int Nk, const double *muedges, int Nmu,
            const int64_t *poles, int Np_all, float *power, int64_t *N_mode, float *binned_poles,
            int64_t *N_mode_poles, float *k_avg) {
    if (Nk < 1 || Nmu < 1) return fail("power: need at least one k bin and one mu bin");
    if (Np_all > MAX_POLES) return fail("power: more than %d multipoles requested", MAX_POLES);
    const int nmesh = s.n, kzlen = s.kzlen;
    const double dk = 2.0 * M_PI / Lbox;
    BinArgs b;
    b.Nk = Nk;
    b.Nmu = Nmu;
    int nz_index[MAX_POLES];   // requested pole -> slot among the ell != 0 accumulators
    b.Np = 0;
    for (int q = 0; q < Np_all; q++) {
        nz_index[q] = -1;
        if (poles[q] != 0) {
            ABACUS_TRY(pole_coefs((int)poles[q], b.polecoef[b.Np]));
            nz_index[q] = b.Np++;
        }
    }
    // edges in units of dk, squared, float32 (:217-218)
    std::vector<float> e2((size_t)Nk + 1 + Nmu + 1);
    for (int q = 0; q <= Nk; q++) e2[q] = (float)((kedges[q] / dk) * (kedges[q] / dk));
    for (int q = 0; q <= Nmu; q++) e2[Nk + 1 + q] = (float)(muedges[q] * muedges[q]);
    ABACUS_TRY(g_ctx.edges.reserve(e2.size() * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(g_ctx.edges.p, e2.data(), e2.size() * sizeof(float), hipMemcpyHostToDevice, stream()));
    b.kedges2 = g_ctx.edges.as<float>();
    b.muedges2 = b.kedges2 + Nk + 1;
    const size_t nb = (size_t)Nk * Nmu, npk = (size_t)b.Np * Nk;
    const size_t acc_bytes = nb * 8 * 3 + npk * 8;
    ABACUS_TRY(g_ctx.accum.reserve(acc_bytes));
    HIP_TRY(hipMemsetAsync(g_ctx.accum.p, 0, acc_bytes, stream()));
    b.g_cnt = g_ctx.accum.as<unsigned long long>();
    b.g_sum = reinterpret_cast<double *>(b.g_cnt + nb);
    b.g_ksum = b.g_sum + nb;
    b.g_pole = b.g_ksum + nb;
    // LDS budget: histogram + edges + tile
    const size_t hist_bytes = nb * (8 + 8 + 4) + npk * 8 + (size_t)(Nk + 1 + Nmu + 1) * 4 + 64;
    const size_t lds_max = 160 * 1024;
    if (hist_bytes + (size_t)kzlen * 4 * 2 > lds_max)
        return fail("power: %d x %d bins with %d multipoles do not fit the 160 KiB LDS histogram", Nk, Nmu, b.Np);
    int rows = (int)((lds_max - hist_bytes) / ((size_t)kzlen * 4));
    const int rows_wanted = std::max(1, (BIN_THREADS * 17 + kzlen - 1) / kzlen);   // ~17 modes per thread
    rows = std::max(1, std::min(rows, rows_wanted));
    int chunk = (int)(((int64_t)rows * kzlen + BIN_THREADS - 1) / BIN_THREADS);
    if (chunk % 2 == 0) chunk++;
    b.rows = rows;
    b.chunk = chunk;
    const size_t lds = hist_bytes + (size_t)rows * kzlen * 4;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(spectrum_bin), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    int dev = 0, ncu = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t ntiles = ((int64_t)nmesh * nmesh + rows - 1) / rows;
    const int grid = (int)std::min<int64_t>(ntiles, ncu);
    ABACUS_LAUNCH("spectrum_bin", spectrum_bin, dim3(grid), dim3(BIN_THREADS), lds, s, b);
    // tiny read-back and the normalisation of bin_kmu (:276-293) / calc_pk_from_deltak (:789-792) in float64
    std::vector<unsigned char> host(acc_bytes);
    HIP_TRY(hipMemcpyAsync(host.data(), g_ctx.accum.p, acc_bytes, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    const unsigned long long *cnt = reinterpret_cast<const unsigned long long *>(host.data());
    const double *sum = reinterpret_cast<const double *>(cnt + nb);
    const double *ksum = sum + nb;
    const double *pole = ksum + nb;
    const double L3 = Lbox * Lbox * Lbox;
    for (int i = 0; i < Nk; i++) {
        int64_t cp = 0;
        double wedge = 0;
        for (int j = 0; j < Nmu; j++) {
            const size_t q = (size_t)i * Nmu + j;
            const int64_t c = (int64_t)cnt[q];
            N_mode[q] = c;
            power[q] = (float)((c ? sum[q] / (double)c : sum[q]) * L3);
            k_avg[q] = (float)(c ? ksum[q] * dk / (double)c : ksum[q] * dk);
            cp += c;
            wedge += sum[q];
        }
        N_mode_poles[i] = cp;
        for (int q = 0; q < Np_all; q++) {
            const double v = poles[q] == 0 ? wedge : pole[(size_t)nz_index[q] * Nk + i];   // l=0 from the wedges (:282-284)
            binned_poles[(size_t)q * Nk + i] = (float)((cp ? v / (double)cp : v) * L3);
        }
    }
    return 0;
}

int upload_W(const float *W_host, int nmesh, const float **W_dev) {
    *W_dev = nullptr;
    if (!W_host) return 0;
    ABACUS_TRY(g_ctx.W.reserve((size_t)nmesh * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(g_ctx.W.p, W_host, (size_t)nmesh * sizeof(float), hipMemcpyHostToDevice, stream()));
    *W_dev = g_ctx.W.as<float>();
    return 0;
}

int check_common(int nmesh, int paste) {
    if (nmesh < 2 || nmesh > 32767) return fail("power: nmesh %d out of range", nmesh);
    if (paste != 0 && paste != 1) return fail("power: unknown paste code %d", paste);
    return 0;
}

int power_dev(float *pos, int64_t n, const float *w, float *pos2, int64_t n2, const float *w2, double Lbox, int nmesh,
              int paste, const float *W_host, int interlaced, const double *kedges, int Nk, const double *muedges,
              int Nmu, const int64_t *poles, int Np, float *power, int64_t *N_mode, float *binned_poles,
              int64_t *N_mode_poles, float *k_avg) {
    ABACUS_TRY(check_common(nmesh, paste));
    ABACUS_TRY(ensure_phase(nmesh));
    const float *W_dev;
    ABACUS_TRY(upload_W(W_host, nmesh, &W_dev));
    ABACUS_TRY(field_fft_dev(pos, n, w, Lbox, nmesh, paste, interlaced, 0));
    const bool cross = pos2 != nullptr;
    if (cross) ABACUS_TRY(field_fft_dev(pos2, n2, w2, Lbox, nmesh, paste, interlaced, 2));
    SpecArgs s;
    fill_spec(s, nmesh, 1, interlaced, W_dev, cross);
    s.a = g_ctx.mesh[0].as<float2>();
    s.as = interlaced ? g_ctx.mesh[1].as<float2>() : nullptr;
    s.b = cross ? g_ctx.mesh[2].as<float2>() : nullptr;
    s.bs = cross && interlaced ? g_ctx.mesh[3].as<float2>() : nullptr;
    return run_bin(s, Lbox, kedges, Nk, muedges, Nmu, poles, Np, power, N_mode, binned_poles, N_mode_poles, k_avg);
}

// upload a host particle set into the context buffers; wrapped positions are copied back like the reference mutates them
int stage_particles(float *pos, int64_t n, const float *w, DevBuf &dpos, DevBuf &dw, float **pd, float **wd) {
    ABACUS_TRY(dpos.reserve((size_t)std::max<int64_t>(n, 1) * 12));
    HIP_TRY(hipMemcpyAsync(dpos.p, pos, (size_t)n * 12, hipMemcpyHostToDevice, stream()));
    *pd = dpos.as<float>();
    *wd = nullptr;
    if (w) {
        ABACUS_TRY(dw.reserve((size_t)std::max<int64_t>(n, 1) * 4));
        HIP_TRY(hipMemcpyAsync(dw.p, w, (size_t)n * 4, hipMemcpyHostToDevice, stream()));
        *wd = dw.as<float>();
    }
    return 0;
}

}  // namespace

extern "C" {

int abacus_power_from_particles_dev(float *pos, int64_t n, const float *w, float *pos2, int64_t n2, const float *w2,
                                    double Lbox, int nmesh, int paste, const float *W_host, int interlaced,
                                    const double *kedges, int Nk, const double *muedges, int Nmu, const int64_t *poles,
                                    int Np, float *power, int64_t *N_mode, float *binned_poles, int64_t *N_mode_poles,
                                    float *k_avg) {
    ABACUS_TRY(ensure_init());
    return power_dev(pos, n, w, pos2, n2, w2, Lbox, nmesh, paste, W_host, interlaced, kedges, Nk, muedges, Nmu, poles,
                     Np, power, N_mode, binned_poles, N_mode_poles, k_avg);
}

int abacus_power_from_particles(float *pos, int64_t n, const float *w, float *pos2, int64_t n2, const float *w2,
                                double Lbox, int nmesh, int paste, const float *W_host, int interlaced,
                                const double *kedges, int Nk, const double *muedges, int Nmu, const int64_t *poles,
                                int Np, float *power, int64_t *N_mode, float *binned_poles, int64_t *N_mode_poles,
                                float *k_avg) {
    ABACUS_TRY(ensure_init());
    float *pd, *wd, *pd2 = nullptr, *wd2 = nullptr;
    ABACUS_TRY(stage_particles(pos, n, w, g_ctx.pos, g_ctx.w, &pd, &wd));
    if (pos2) ABACUS_TRY(stage_particles(pos2, n2, w2, g_ctx.pos2, g_ctx.w2, &pd2, &wd2));
    ABACUS_TRY(power_dev(pd, n, wd, pd2, n2, wd2, Lbox, nmesh, paste, W_host, interlaced, kedges, Nk, muedges, Nmu,
                         poles, Np, power, N_mode, binned_poles, N_mode_poles, k_avg));
    if (paste == 0) {  // TSC wraps the caller's positions in place (tsc.py:171-173); CIC does not wrap (cic.py)
        HIP_TRY(hipMemcpyAsync(pos, pd, (size_t)n * 12, hipMemcpyDeviceToHost, stream()));
        if (pos2) HIP_TRY(hipMemcpyAsync(pos2, pd2, (size_t)n2 * 12, hipMemcpyDeviceToHost, stream()));
        HIP_TRY(hipStreamSynchronize(stream()));
    }
    return 0;
}

int abacus_field_fft(float *pos, int64_t n, const float *w, double Lbox, int nmesh, int paste, const float *W_host,
                     int interlaced, void *out_c64) {
    ABACUS_TRY(ensure_init());
    ABACUS_TRY(check_common(nmesh, paste));
    ABACUS_TRY(ensure_phase(nmesh));
    const float *W_dev;
    ABACUS_TRY(upload_W(W_host, nmesh, &W_dev));
    float *pd, *wd;
    ABACUS_TRY(stage_particles(pos, n, w, g_ctx.pos, g_ctx.w, &pd, &wd));
    ABACUS_TRY(field_fft_dev(pd, n, wd, Lbox, nmesh, paste, interlaced, 0));
    SpecArgs s;
    fill_spec(s, nmesh, 1, interlaced, W_dev, false);
    s.a = g_ctx.mesh[0].as<float2>();
    s.as = interlaced ? g_ctx.mesh[1].as<float2>() : nullptr;
    s.b = s.bs = nullptr;
    const int64_t total = (int64_t)nmesh * nmesh * (nmesh / 2 + 1);
    const int grid = (int)std::min<int64_t>(ceil_div(total, 256), 256 * 32);
    ABACUS_LAUNCH("spectrum_apply", spectrum_apply, dim3(grid), dim3(256), 0, s, g_ctx.mesh[0].as<float2>());
    HIP_TRY(hipMemcpyAsync(out_c64, g_ctx.mesh[0].p, (size_t)total * 8, hipMemcpyDeviceToHost, stream()));
    if (paste == 0) HIP_TRY(hipMemcpyAsync(pos, pd, (size_t)n * 12, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_pk_from_deltak(const void *field, const void *field2, int nmesh, double Lbox, const double *kedges, int Nk,
                          const double *muedges, int Nmu, const int64_t *poles, int Np, float *power, int64_t *N_mode,
                          float *binned_poles, int64_t *N_mode_poles, float *k_avg) {
    ABACUS_TRY(ensure_init());
    if (!field) return fail("abacus_pk_from_deltak: null field");
    if (nmesh < 2 || nmesh > 32767) return fail("power: nmesh %d out of range", nmesh);
    const size_t bytes = (size_t)nmesh * nmesh * (nmesh / 2 + 1) * 8;
    ABACUS_TRY(g_ctx.mesh[0].reserve(std::max(bytes, mesh_bytes(nmesh))));
    HIP_TRY(hipMemcpyAsync(g_ctx.mesh[0].p, field, bytes, hipMemcpyHostToDevice, stream()));
    if (field2) {
        ABACUS_TRY(g_ctx.mesh[2].reserve(std::max(bytes, mesh_bytes(nmesh))));
        HIP_TRY(hipMemcpyAsync(g_ctx.mesh[2].p, field2, bytes, hipMemcpyHostToDevice, stream()));
    }
    SpecArgs s;
    fill_spec(s, nmesh, 0, 0, nullptr, field2 != nullptr);
    s.a = g_ctx.mesh[0].as<float2>();
    s.b = field2 ? g_ctx.mesh[2].as<float2>() : nullptr;
    s.as = s.bs = nullptr;
    return run_bin(s, Lbox, kedges, Nk, muedges, Nmu, poles, Np, power, N_mode, binned_poles, N_mode_poles, k_avg);
}

int abacus_power_release(void) {
    for (auto &kv : g_ctx.plans) (void)hipfftDestroy(kv.second);
    g_ctx.plans.clear();
    for (auto &m : g_ctx.mesh) ABACUS_TRY(m.release());
    for (DevBuf *b : {&g_ctx.W, &g_ctx.phase, &g_ctx.edges, &g_ctx.accum, &g_ctx.pos, &g_ctx.pos2, &g_ctx.w, &g_ctx.w2})
        ABACUS_TRY(b->release());
    g_ctx.phase_n = 0;
    return tsc_release_work();
}

}  // extern "C"
