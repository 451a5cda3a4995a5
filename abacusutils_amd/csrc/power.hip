// Power-spectrum estimator on MI355X (gfx950): replaces the calc_power chain of the reference
// (abacusnbody/analysis/power_spectrum.py): get_field :808-857 (deposit + normalize_field :860-901),
// scipy.fft.rfftn :980,986,1059 + _normalize :1073-1078, get_interlaced_field_fft / shift_field_fft :951-998,
// :904-948, the compensation divide :1063-1069, get_raw_power :707-727 and bin_kmu :150-300.
//
// Data layout in HBM: one float32 mesh per field, rows of `pitch_r` floats with pitch_r = roundup(n + 2, 32): an
// in-place R2C layout whose rows start on 128-B boundaries (with the minimal n + 2 pitch every 128-B row segment of a
// tile straddles two cache lines and partial-line writes run 4x slower - measured 1.3 vs 5.5 TB/s).  After the FFT the
// same buffer is the complex64 half-spectrum, rows of pitch_r/2 complex of which n/2+1 are valid.  Nothing else of
// mesh size is allocated (plus rocFFT's work area); the spectrum never goes back over PCIe in the fused path.
//
// Passes over mesh-sized data (algorithmic bytes 36*M non-interlaced):
//   tsc_tile_deposit  writes the mesh once, normalisation delta = rho*M/N - 1 fused into the tile flush (4M)
//   hipFFT/rocFFT R2C in place                                                              (~3 x (4M + 4M))
//   spectrum_bin      reads the half-spectrum once and fuses scale (1/M), interlacing combine, compensation,
//                     |delta_k|^2 (or the cross power) and the (k, mu) / multipole binning            (4M, 8M, 16M)
//
// spectrum_bin: persistent workgroups (one per CU, 1024 threads).  A tile is a flat range of the half-spectrum;
// while a tile is binned the next one is already in flight into registers (16-B loads).  The power of each mode is
// staged in LDS; every thread then walks 16 consecutive kz exactly like the reference's inner loop (monotone bin
// search, :246-256), accumulating runs of equal bins in registers and touching the workgroup's LDS histogram only
// when the bin changes.  Histograms are float64 / integer and are flushed to HBM with one atomic per non-empty bin
// per workgroup at the very end.  (The reference keeps float32 per-thread accumulators, :221-229; float64 sums are
// strictly more accurate and thread-count independent.)
#include <hipfft/hipfft.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "../../include/abacus_hip.h"
#include "common.hpp"
#include "bin_device.hpp"

using namespace abacus;

namespace abacus {
int tsc_deposit_f32(float *pos, int64_t n, const float *w, float *grid, int nmesh, int64_t zstride, double box,
                    double offset, int wrap, double norm, int cic, int list_mode = 0, double sub = 1.0, int zero_grid = 1);
void tsc_wrapped_reset();
int tsc_wrapped_seen();
void tsc_lines_defer(int on);
int tsc_lines_deferred_check();
int tsc_deposit_f64mesh(void *pos, int pos_f64, int64_t n, const void *w, double *grid, int nmesh, int64_t zstride, double box,
                        double offset, int wrap, double norm, int cic, double sub);
bool gfft_supported(int n, int is_double);
int gfft_r2c_inplace_f64(double *mesh, int n, int pitch_r);
int tsc_deposit_f64pos(double *pos, int64_t n, const double *w, float *grid, int nmesh, int64_t zstride, double box,
                       double offset, int wrap, double norm, int cic, double sub = 1.0);
int tsc_release_work();
bool fft_native_supported(int n);
bool fft_native_pow2(int n);
int fft_native_r2c_inplace(float *mesh, int n, int pitch_r, float xcut = 0.f);
int fft_native_zy(float *mesh, int n, int pitch_r, int64_t nx_local);
int fft_native_x(float *mesh, int n, int pitch_r, int64_t ny_local, int64_t x_stride, int64_t y_stride);
int tsc_deposit_slab_f32(float *pos, int64_t n, const float *w, float *grid, int nmesh, int xoff, int nx_local,
                         int64_t zstride, double box, double offset, int wrap, double norm, int cic, double sub, int xoff2, int nx_alloc = 0);
int fft_native_release();
int fft_native_fused_supported(int n);
int fft_native_r2c_fused(float *mesh, int n, int pitch_r, float xcut = 0.f);   // rows come out in the permuted order of fft.hip's fused form
int fft_native_r2c_fused_zy(float *mesh, int n, int pitch_r, float xcut = 0.f);
bool xbin_supported(int n, int Nk, int Nmu, const BinArgs &b, bool comp);
int fft_x_bin_run(const float *mesh, int n, int pitch_r, float inv_size, const float *W_dev, const BinArgs &b, int dbg,
                  int y0 = 0, int ny_local = 0, int put_geom = 1, int layout = 0, int world = 1, const float *mesh_shifted = nullptr,
                  const float2 *phase = nullptr, const unsigned int *row_off = nullptr, int64_t plane_elems = 0,
                  const float *mesh_b = nullptr, const float *mesh_b_shifted = nullptr);
bool xbin2_supported(int n, const BinArgs &b, bool comp);
bool gfft_supported(int n, int is_double);
int gfft_r2c_zy_f32(float *mesh, int n, int pitch_r);
bool gfft_xbin_supported(int n, const BinArgs &b, bool comp, bool inter = false);
int gfft_x_bin_run(const float *mesh, int n, int pitch_r, float inv_size, const float *W_dev, const BinArgs &b, const float *mesh2 = nullptr,
                   const void *phase = nullptr);
int fft_native_fused_zy_slab(float *mesh, int n, int pitch_r, int h, int64_t xsep, int xg0, int p0, int pc, float *pack_out,
                             int world, float cut = 0.f);
int slab_layout_query(int n, int W, float cut, int pitch_c, int64_t *P_out, const unsigned int **row_off_dev);
int fft_native_fused_x_slab(float *mesh, int n, int pitch_r, int64_t ny_local);
double xbin_last_build_ms();
int xbin_last_gen();
int xbin_release();
}  // namespace abacus

namespace {

constexpr int BIN_THREADS = 1024;

struct SpecArgs {
    int n, kzlen, pitch;      // pitch: complex elements per (kx, ky) row in memory (>= kzlen)
    int rowmode, y0;          // 0: row = kx*n + ky (full spectrum); 1: row = ky_local*n + kx, ky = y0 + ky_local (y-slab)
    int permshift;            // > 0: rows are in the order of the fused transform: index r of x and y holds frequency
                              // 2 (r mod n/2) + (r div n/2); permshift = log2(n/2)
    int64_t nrows;            // rows held in memory (n*n, or ny_local*n for a y-slab)
    int mode;                 // 0: raw fields (deltak API), 1: FFT output needing scale/interlace/compensation
    int interlaced, compensated, cross;
    int linear;               // mode 0 only: the value binned is Re(a) (caller-supplied real weights), not |a|^2
    float inv_size;           // f32(1/M)            (:1058)
    float half_inv_size;      // f32(0.5/M)          (:934)
    const float2 *a, *as, *b, *bs;   // field 1 (+ shifted), field 2 (+ shifted); b == nullptr -> auto power
    const float *W;           // (n,) window or nullptr
    const float2 *phase;      // (2n,) e^{i*pi*m/n}
    int lds_tables;           // spectrum_bin: W and phase are copied into LDS behind the tile buffer (they fit)
};

__device__ __forceinline__ int fold(int i, int n) { return i < n / 2 ? i : i - n; }   // (:234,237,940-942)
__device__ __forceinline__ void row_ij(const SpecArgs &s, int64_t row, int &i, int &j) {
    if (s.rowmode == 0) {
        j = (int)(row % s.n), i = (int)(row / s.n);
        if (s.permshift) {
            const int hm = (1 << s.permshift) - 1;
            i = ((i & hm) << 1) | (i >> s.permshift);
            j = ((j & hm) << 1) | (j >> s.permshift);
        }
    } else {
        i = (int)(row % s.n), j = s.y0 + (int)(row / s.n);
        if (s.permshift) {
            const int hm = (1 << s.permshift) - 1;
            i = ((i & hm) << 1) | (i >> s.permshift);
            j = ((j & hm) << 1) | (j >> s.permshift);
        }
    }
}

// x / d and x % d for 0 <= x < 2^53, 0 < d < 2^31 without the 64-bit integer division (about a hundred instructions on
// this hardware): one float64 multiply by 1/d, exact product, remainder fixed up by at most one step
__device__ __forceinline__ void divmod_inv(int64_t x, int d, double inv_d, int64_t &q, int &r) {
    q = (int64_t)((double)x * inv_d);
    int64_t rem = x - q * d;
    if (rem < 0) q--, rem += d;
    else if (rem >= d) q++, rem -= d;
    r = (int)rem;
}
// (i, j) of the row whose slow / fast row digits are (hi, lo) = (row / n, row % n): row_ij without the divisions
__device__ __forceinline__ void hilo_ij(const SpecArgs &s, int hi, int lo, int &i, int &j) {
    if (s.rowmode == 0) {
        if (s.permshift) {
            const int hm = (1 << s.permshift) - 1;
            hi = ((hi & hm) << 1) | (hi >> s.permshift);
            lo = ((lo & hm) << 1) | (lo >> s.permshift);
        }
        i = hi, j = lo;
    } else {
        i = lo, j = s.y0 + hi;
        if (s.permshift) {
            const int hm = (1 << s.permshift) - 1;
            i = ((i & hm) << 1) | (i >> s.permshift);
            j = ((j & hm) << 1) | (j >> s.permshift);
        }
    }
}

// final delta_k of one field at (i, j, k) from the raw FFT output v (and the shifted field's w): what get_field_fft
// returns (:1046-1070)
// W / phase: the tables of SpecArgs, or copies of them in LDS (spectrum_bin)
__device__ __forceinline__ float2 finish_value(const SpecArgs &s, float2 v, float2 w, int i, int j, int k, const float *W,
                                               const float2 *phase) {
    if (s.mode == 0) return v;
    if (s.interlaced) {
        // (delta_k + delta'_k * exp(i*(d/2)*(kx+ky+kz))) * f32(0.5/M); (d/2)*dk = pi/n, so the phase only depends
        // on m = i' + j' + k (mod 2n): taken from a table of exact angles.  |i'|, |j'| <= n/2 and 0 <= k < n: m lies in
        // (-2n, 2n), one conditional add replaces the modulo
        int m = fold(i, s.n) + fold(j, s.n) + k;
        if (m < 0) m += 2 * s.n;
        const float2 ph = phase[m];
        const float re = w.x * ph.x - w.y * ph.y, im = w.x * ph.y + w.y * ph.x;
        v.x = (v.x + re) * s.half_inv_size;
        v.y = (v.y + im) * s.half_inv_size;
    } else {
        v.x *= s.inv_size;
        v.y *= s.inv_size;
    }
    if (s.compensated) {
        const float wgt = (W[i] * W[j]) * W[k];   // (:1065-1069), NumPy divides complex by real as *(1/w)
        const float scl = 1.0f / wgt;
        v.x *= scl;
        v.y *= scl;
    }
    return v;
}

// in-place finalisation for abacus_field_fft (spectrum returned to the host)
__global__ void spectrum_apply(SpecArgs s, float2 *out) {
    const int64_t total = s.nrows * s.kzlen;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(q % s.kzlen);
        const int64_t row = q / s.kzlen;
        int i, j;
        row_ij(s, row, i, j);
        const int64_t idx = row * s.pitch + k;
        const float2 w = s.interlaced ? s.as[idx] : make_float2(0.f, 0.f);
        out[idx] = finish_value(s, s.a[idx], w, i, j, k, s.W, s.phase);
    }
}

// Tile geometry: a tile is a flat range of the (pitched) half-spectrum, EPT consecutive elements per thread.
// Registers hold the next tile of every field that is read (1, 2 or 4 fields): 16 per thread, 8 when all four are live.
template <bool INTER, bool CROSS>
struct BinCfg {
    static constexpr int EPT = (INTER && CROSS) ? 4 : ((INTER || CROSS) ? 8 : 16);   // 2 or 4 fields in flight: shorter runs keep the prefetch in registers
    static constexpr int TILE_MODES = BIN_THREADS * EPT;
    static constexpr int LOADS = EPT / 2;                   // float4 loads per thread per field and tile
};

template <bool INTER, bool CROSS>
struct TileRegs {
    float4 a[BinCfg<INTER, CROSS>::LOADS];
    float4 as[INTER ? BinCfg<INTER, CROSS>::LOADS : 1];
    float4 b[CROSS ? BinCfg<INTER, CROSS>::LOADS : 1];
    float4 bs[(INTER && CROSS) ? BinCfg<INTER, CROSS>::LOADS : 1];
};

__device__ __forceinline__ float4 ld4(const float2 *p, int64_t idx, int64_t total) {
    if (idx + 1 < total) return *reinterpret_cast<const float4 *>(p + idx);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < total) {
        const float2 t = p[idx];
        v.x = t.x, v.y = t.y;
    }
    return v;
}

template <bool INTER, bool CROSS>
__device__ __forceinline__ void tile_load(const SpecArgs &s, int64_t base, int64_t total, TileRegs<INTER, CROSS> &r) {
#pragma unroll
    for (int q = 0; q < BinCfg<INTER, CROSS>::LOADS; q++) {
        // wave-local: wave w loads the modes [w * 64 * EPT, (w + 1) * 64 * EPT) its own lanes bin, 1 KB per load
        const int64_t idx = base + (int64_t)(threadIdx.x >> 6) * (64 * BinCfg<INTER, CROSS>::EPT) + (q * 64 + (threadIdx.x & 63)) * 2;
        r.a[q] = ld4(s.a, idx, total);
        if (INTER) r.as[q] = ld4(s.as, idx, total);
        if (CROSS) r.b[q] = ld4(s.b, idx, total);
        if (INTER && CROSS) r.bs[q] = ld4(s.bs, idx, total);
    }
}

// LDS tile index with one pad word per EPT modes: a thread's run starts EPT+1 words after its neighbour's -> no conflicts
template <int EPT>
__device__ __forceinline__ int tpad(int e) { return e + e / EPT; }

template <bool INTER, bool CROSS>
__device__ __forceinline__ void tile_store(const SpecArgs &s, int64_t base, int64_t total,
                                           const TileRegs<INTER, CROSS> &r, float *tile, const float *W,
                                           const float2 *phase, double inv_pitch, double inv_n) {
    constexpr int EPT = BinCfg<INTER, CROSS>::EPT;
    const bool need_idx = s.mode == 1 && (INTER || s.compensated);
    // (row, k) and the row's two slow indices (hi = row / n, lo = row % n) of this lane's first pair: the only 64-bit
    // divisions, once per thread and tile; the lane's later pairs are 128 elements apart and advance by carries
    const int e_first = (threadIdx.x >> 6) * (64 * EPT) + (threadIdx.x & 63) * 2;
    int64_t row = 0;
    int k = 0, hi = 0, lo = 0;
    if (need_idx) {
        int64_t h64;
        divmod_inv(base + e_first, s.pitch, inv_pitch, row, k);
        divmod_inv(row, s.n, inv_n, h64, lo);
        hi = (int)h64;
    }
#pragma unroll
    for (int q = 0; q < BinCfg<INTER, CROSS>::LOADS; q++) {
        const int e = e_first + q * 128;
        int i = 0, j = 0, kc = 0;
        if (need_idx) {
            if (row < s.nrows) {      // rows past the end (last tile) are never binned: indices stay 0
                hilo_ij(s, hi, lo, i, j);
                kc = k;
            }
            k += 128;                 // next pair of this lane
            while (k >= s.pitch) {
                k -= s.pitch;
                row++;
                if (++lo == s.n) lo = 0, hi++;
            }
        }
        const float4 zs = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 ras = INTER ? r.as[q] : zs, rbs = (INTER && CROSS) ? r.bs[q] : zs;
        // pitch is even and e is even: the pair (kc, kc + 1) never straddles two rows; the row padding (k >= kzlen) is
        // never binned, its table indices are clamped
        const int k0 = kc < s.kzlen ? kc : 0, k1 = kc + 1 < s.kzlen ? kc + 1 : 0;
        const float2 va0 = finish_value(s, make_float2(r.a[q].x, r.a[q].y), make_float2(ras.x, ras.y), i, j, k0, W, phase);
        const float2 va1 = finish_value(s, make_float2(r.a[q].z, r.a[q].w), make_float2(ras.z, ras.w), i, j, k1, W, phase);
        float p0, p1;
        if (CROSS) {
            const float2 vb0 = finish_value(s, make_float2(r.b[q].x, r.b[q].y), make_float2(rbs.x, rbs.y), i, j, k0, W, phase);
            const float2 vb1 = finish_value(s, make_float2(r.b[q].z, r.b[q].w), make_float2(rbs.z, rbs.w), i, j, k1, W, phase);
            p0 = va0.x * vb0.x + va0.y * vb0.y;   // Re(conj(a) b)  (:724)
            p1 = va1.x * vb1.x + va1.y * vb1.y;
        } else {
            p0 = s.linear ? va0.x : va0.x * va0.x + va0.y * va0.y;   // |a|^2 (:726), or the caller's weight itself
            p1 = s.linear ? va1.x : va1.x * va1.x + va1.y * va1.y;
        }
        tile[tpad<EPT>(e)] = p0;
        tile[tpad<EPT>(e + 1)] = p1;
    }
}

// NP: compile-time number of ell != 0 multipoles (0..3), or -1 for the generic loop over b.Np
// PD: degree (in mu^2) of the Horner evaluation of the Legendre weights: 2 (ell <= 4; -1 with the generic NP); the coefficients are
// hoisted into registers, zero-padded above a pole's own degree (0 * mu^2 + c is c: bit-identical to starting at the
// pole's degree) - indexing the kernel-argument arrays in the loop costs a scalar memory load + wait per mode
template <bool INTER, bool CROSS, int NP, int PD>
__global__ __launch_bounds__(BIN_THREADS) void spectrum_bin(SpecArgs s, BinArgs b) {
    constexpr int BIN_EPT = BinCfg<INTER, CROSS>::EPT, BIN_TILE = BinCfg<INTER, CROSS>::TILE_MODES;
    constexpr int NPC = NP < 0 ? MAX_POLES : (NP > 0 ? NP : 1);
    extern __shared__ __align__(16) unsigned char smem[];
    const int nb = b.Nk * b.Nmu;
    const int np = NP < 0 ? b.Np : NP;
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
    // window and interlacing-phase tables: gathered per mode in the staging, from LDS when they fit
    const float *Wt = s.W;
    const float2 *pht = s.phase;
    if (s.lds_tables) {
        float2 *phl = reinterpret_cast<float2 *>(tile + (BIN_TILE + BIN_THREADS + 16));   // 8-B aligned: even float count
        float *Wl = reinterpret_cast<float *>(phl + (INTER ? 2 * s.n : 0));
        if (INTER)
            for (int q = tid; q < 2 * s.n; q += BIN_THREADS) phl[q] = s.phase[q];
        if (s.compensated)
            for (int q = tid; q < s.n; q += BIN_THREADS) Wl[q] = s.W[q];
        Wt = Wl, pht = phl;
    }
    __syncthreads();
    const float klo = ke[0], khi = ke[b.Nk];
    const int n = s.n, kzlen = s.kzlen, pitch = s.pitch;
    const double inv_pitch = 1.0 / (double)pitch, inv_n = 1.0 / (double)n;
    constexpr int PDC = PD < 0 ? 0 : PD;      // PD < 0 (generic pole count): coefficients stay in the kernel arguments
    float pc[PD < 0 ? 1 : NPC][PDC + 1];
    if constexpr (PD >= 0) {
#pragma unroll
        for (int q = 0; q < NPC; q++)
#pragma unroll
            for (int m = 0; m <= PDC; m++) pc[q][m] = (q < np && m <= b.poledeg[q]) ? b.polecoef[q][m] : 0.f;
    }
    const int64_t total = s.nrows * pitch;
    const int64_t ntiles = (total + BIN_TILE - 1) / BIN_TILE;

    TileRegs<INTER, CROSS> regs;
    int64_t t = blockIdx.x;
    if (t < ntiles) tile_load<INTER, CROSS>(s, t * BIN_TILE, total, regs);
    for (; t < ntiles; t += gridDim.x) {
        const int64_t base = t * BIN_TILE;
        // ---- power of this tile's modes: registers -> LDS; then prefetch the next tile into the registers, so its
        //      HBM latency is covered by the binning below ----
        if (!(b.dbg & 2)) tile_store<INTER, CROSS>(s, base, total, regs, tile, Wt, pht, inv_pitch, inv_n);
        wave_sync();   // a wave stages and bins its own 64 * EPT modes: no workgroup barrier, the waves drift apart
        if (t + gridDim.x < ntiles) tile_load<INTER, CROSS>(s, (t + gridDim.x) * BIN_TILE, total, regs);
        // ---- bin: every thread walks EPT consecutive elements: at most two row segments.  Along a row |k| and mu only
        //      grow with kz, so a segment is located once (binary searches at its first binned mode) and the inner loop
        //      only advances: bins change by `while` steps, a change of the target bin flushes the register accumulators
        //      of the finished run into the LDS histogram (one flush site in the loop, one behind it) ----
        const int e0 = tid * BIN_EPT;
        const int64_t idx0 = base + e0;
        if (idx0 < total && !(b.dbg & 1)) {
            int64_t row, h64;
            int k, hi, lo;
            divmod_inv(idx0, pitch, inv_pitch, row, k);
            divmod_inv(row, n, inv_n, h64, lo);
            hi = (int)h64;
            int cur = -1, cur_bk = 0, cnt = 0;
            float sp = 0.f, sk = 0.f, spole[NPC];
#pragma unroll
            for (int q = 0; q < NPC; q++) spole[q] = 0.f;
            auto flush = [&]() {
                if (cnt) {
                    atomicAdd(&h_cnt[cur], (unsigned int)cnt);
                    atomicAdd(&h_sum[cur], (double)sp);
                    atomicAdd(&h_ksum[cur], (double)sk);
#pragma unroll
                    for (int q = 0; q < NPC; q++)
                        if (q < np) atomicAdd(&h_pole[q * b.Nk + cur_bk], (double)spole[q]);
                }
                cnt = 0;
                sp = sk = 0.f;
#pragma unroll
                for (int q = 0; q < NPC; q++) spole[q] = 0.f;
            };
            const float *trun = tile + tpad<BIN_EPT>(e0);   // the run is contiguous in the padded tile
            int left = BIN_EPT, epos = 0;
#pragma unroll 1
            while (left > 0 && row < s.nrows) {
                const int seg = min(left, pitch - k);
                int r2;                               // i'^2 + j'^2 <= 2*(n/2)^2 < 2^30 for n <= 32767
                {
                    int ii, jj;
                    hilo_ij(s, hi, lo, ii, jj);
                    ii = fold(ii, n), jj = fold(jj, n);
                    r2 = ii * ii + jj * jj;
                }
                int ka = k;
                const int kb = min(k + seg, kzlen);
                while (ka < kb && (float)(r2 + ka * ka) < klo) ka++;   // `continue` below the first edge (:246)
                if (ka < kb && (float)(r2 + ka * ka) < khi) {
                    // locate the first binned mode of the segment exactly as the reference's running search would
                    const float kmag2a = (float)(r2 + ka * ka);
                    int bk = lower_bin(ke, b.Nk - 1, kmag2a);
                    float ke_hi = ke[bk + 1];
                    const float mu2a = kmag2a > 0.f ? (float)(ka * ka) * (1.0f / kmag2a) : 0.f;   // IEEE division
                    int bmu = lower_bin(me, b.Nmu - 1, mu2a);
                    float me_lo = me[bmu], me_hi = me[bmu + 1];
#pragma unroll 1
                    for (int kk = ka; kk < kb; kk++) {
                        const int k2 = kk * kk;
                        const float kmag2 = (float)(r2 + k2);   // dtype(i2 + j2 + k**2) (:239); exact integer below 2^24
                        if (kmag2 >= khi) break;                // `break` from the last edge on (:249)
                        while (kmag2 > ke_hi) {                 // (:252-253)
                            bk++;
                            ke_hi = ke[bk + 1];
                        }
                        // mu^2 = f32(k^2) * (1/kmag2) (:240-244).  The hardware reciprocal (1 ulp) picks the bin unless
                        // mu^2 lands within a few ulp of an edge; only then the correctly rounded value is formed.
                        const float k2f = (float)k2;
                        float mu2 = kmag2 > 0.f ? k2f * __builtin_amdgcn_rcpf(kmag2) : 0.f;
                        while (bmu + 1 < b.Nmu && mu2 > me_hi) {   // (:255-256)
                            bmu++;
                            me_lo = me_hi;
                            me_hi = me[bmu + 1];
                        }
                        const float tol = 6e-7f * mu2;
                        if (fabsf(mu2 - me_hi) <= tol || fabsf(mu2 - me_lo) <= tol) {
                            mu2 = kmag2 > 0.f ? k2f * (1.0f / kmag2) : 0.f;   // IEEE division, as the reference rounds it
                            bmu = lower_bin(me, b.Nmu - 1, mu2);
                            me_lo = me[bmu];
                            me_hi = me[bmu + 1];
                        }
                        const int tb = bk * b.Nmu + bmu;
                        if (tb != cur) {
                            flush();
                            cur = tb;
                            cur_bk = bk;
                        }
                        const float p = trun[epos + (kk - k)];
                        const float wgt = kk == 0 ? 1.f : 2.f;
                        cnt += kk == 0 ? 1 : 2;
                        const float wp = wgt * p;
                        sp += wp;
                        sk += wgt * __builtin_amdgcn_sqrtf(kmag2);
#pragma unroll
                        for (int q = 0; q < NPC; q++)
                            if (q < np) {
                                float Lq;
                                if constexpr (PD >= 0) {
                                    Lq = pc[q][PDC];
#pragma unroll
                                    for (int m = PDC - 1; m >= 0; m--) Lq = Lq * mu2 + pc[q][m];
                                } else {
                                    const float *c = b.polecoef[q];
                                    Lq = c[b.poledeg[q]];
                                    for (int m = b.poledeg[q] - 1; m >= 0; m--) Lq = Lq * mu2 + c[m];
                                }
                                spole[q] += wp * Lq;
                            }
                    }
                }
                left -= seg;
                epos += seg;
                k = 0;
                row++;
                if (++lo == n) lo = 0, hi++;
            }
            flush();
        }
        wave_sync();   // this wave's part of the tile buffer is free for its next staging
    }
    __syncthreads();
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


// ---- x-slab / y-slab building blocks of the multi-GPU estimator -----------------------------------
// Host code (abacusutils_amd/parallel/slab_power.py) owns the communication: ghost planes by send/recv, the
// pencil transpose by all-to-all, the histogram by all-reduce; everything of mesh size stays on the device.
__global__ void slab_axpy(float *__restrict__ dst, const float *__restrict__ src, int64_t n4, float add) {
    // dst = dst + src + add  (float4 lanes; src may be null)
    float4 *d = reinterpret_cast<float4 *>(dst);
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = d[i];
        if (s4) {
            const float4 w = s4[i];
            v.x += w.x, v.y += w.y, v.z += w.z, v.w += w.w;
        }
        v.x += add, v.y += add, v.z += add, v.w += add;
        d[i] = v;
    }
}

// Folded slabs: a rank owns h plane pairs (x, x + n/2); `data` holds the first planes of the pairs, the second ones xsep
// planes behind them.  send[q][s h + p][yl][k] = data[s xsep + p][q nyl + yl][k] for the pairs p in [p0, p0 + pc) (rows of
// `pitch` complex, copied as 16-B pieces); a chunk of pairs at a time, so that the transpose of a chunk is on the links while
// the next is transformed
__global__ void slab_pack(const float4 *__restrict__ data, float4 *__restrict__ send, int n, int h, int64_t xsep, int nyl,
                          int world, int pitch4, int p0, int pc) {
    const int64_t rows = (int64_t)world * 2 * pc * nyl;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const int yl = (int)(r % nyl), u = (int)((r / nyl) % (2 * pc)), q = (int)(r / ((int64_t)nyl * 2 * pc));
        const int sh = u / pc, p = p0 + u % pc;
        const float4 *src = data + ((sh * xsep + p) * n + (q * nyl + yl)) * pitch4;
        float4 *dst = send + (((int64_t)q * 2 * h + sh * h + p) * nyl + yl) * pitch4;
        for (int t = threadIdx.x; t < pitch4; t += blockDim.x) dst[t] = src[t];
    }
}
// out[yl][s n/2 + q h + p][k] = recv[q][s h + p][yl][k]: behind the transpose row s n/2 + i is plane i of half s of x - the
// plain transform's plane x = s n/2 + i, or in the fused form the sum (s = 0) / twiddled difference (s = 1) of planes i, i + n/2
__global__ void slab_unpack(const float4 *__restrict__ recv, float4 *__restrict__ out, int n, int h, int nyl, int world,
                            int pitch4) {
    const int64_t rows = (int64_t)world * 2 * h * nyl;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const int yl = (int)(r % nyl), u = (int)((r / nyl) % (2 * h)), q = (int)(r / ((int64_t)nyl * 2 * h));
        const int sh = u / h, p = u % h;
        const float4 *src = recv + r * pitch4;
        float4 *dst = out + ((int64_t)yl * n + (sh * (n / 2) + q * h + p)) * pitch4;
        for (int t = threadIdx.x; t < pitch4; t += blockDim.x) dst[t] = src[t];
    }
}

// ---- ZCV-facing helpers (analysis/power_spectrum.py:303-660): elementwise generators and (k_perp, pi) binning --------
// real weights (n, n, zdim) -> complex (w, 0) rows of kzlen = n/2+1 (the binning kernel's caller-supplied layout)
__global__ void helper_real_to_c64(const float *__restrict__ w, float2 *__restrict__ out, int n, int zdim, int kzlen,
                                   float scale) {
    const int64_t total = (int64_t)n * n * kzlen;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(q % kzlen);
        const int64_t row = q / kzlen;
        out[q] = make_float2(w[row * zdim + k] * scale, 0.f);
    }
}

__device__ __forceinline__ void helper_mode(int64_t q, int n, int kzlen, int &i2j2, int &k) {
    k = (int)(q % kzlen);
    const int64_t row = q / kzlen;
    const int j = fold((int)(row % n), n), i = fold((int)(row / n), n);
    i2j2 = i * i + j * j;
}

// get_smoothing (:527-577): exp(-kmag2 * dk^2 * R^2 / 2) in float32
__global__ void helper_smoothing(float *__restrict__ out, int n, float dk2, float R2) {
    const int kzlen = n / 2 + 1;
    const int64_t total = (int64_t)n * n * kzlen;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        int r2, k;
        helper_mode(q, n, kzlen, r2, k);
        const float kmag2 = (float)(r2 + k * k);
        out[q] = expf(-kmag2 * dk2 * R2 / 2.0f);
    }
}

// get_delta_mu2 (:580-617): delta * mu^2, mu^2 = f32(k^2) * kmag2**-1 (0 at k = 0)
__global__ void helper_delta_mu2(const float2 *__restrict__ d, float2 *__restrict__ out, int n) {
    const int kzlen = n / 2 + 1;
    const int64_t total = (int64_t)n * n * kzlen;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        int r2, k;
        helper_mode(q, n, kzlen, r2, k);
        const float kmag2 = (float)(r2 + k * k);
        const float mu2 = kmag2 > 0.f ? (float)(k * k) * (1.0f / kmag2) : 0.f;
        out[q] = make_float2(d[q].x * mu2, d[q].y * mu2);
    }
}

// get_raw_power (:707-727): |a|^2, or Re(conj(a) b)
__global__ void helper_raw_power(const float2 *__restrict__ a, const float2 *__restrict__ b, float *__restrict__ out, int64_t total) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const float2 u = a[q];
        if (b) {
            const float2 v = b[q];
            out[q] = u.x * v.x + u.y * v.y;                      // (conj(u) v).real
        } else {
            const float m = hypotf(u.x, u.y);                    // np.abs(z) ** 2, like the reference
            out[q] = m * m;
        }
    }
}

// shift_field_fft (:904-948): f += s * exp(i (d / 2) (kx + ky + kz)); f *= 0.5 / n^3 - float32 wavenumbers as the reference
// forms them (k = f32(i) dk below n / 2, f32(i - n) dk from there on: the Nyquist plane takes the negative branch)
__global__ void helper_shift_field(float2 *__restrict__ f, const float2 *__restrict__ s, int n, float dk, float halfd, float norm) {
    const int kzlen = n / 2 + 1;
    const int64_t total = (int64_t)n * n * kzlen;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(q % kzlen), j = (int)((q / kzlen) % n), i = (int)(q / ((int64_t)kzlen * n));
        const float kx = i < n / 2 ? (float)i * dk : (float)(i - n) * dk, ky = j < n / 2 ? (float)j * dk : (float)(j - n) * dk,
                    kz = (float)k * dk;
        const float ph = halfd * (kx + ky + kz);
        float sn, cs;
        sincosf(ph, &sn, &cs);
        const float2 a = f[q], b = s[q];
        f[q] = make_float2((a.x + (b.x * cs - b.y * sn)) * norm, (a.y + (b.x * sn + b.y * cs)) * norm);
    }
}

// expand_poles_to_3d (:451-505) with linear_interp (:508-537): Pk = sum_l interp(P_l)(|k|) * P_l(mu)
__global__ void helper_expand_poles(float *__restrict__ out, int n, float dk, const float *__restrict__ k_ell,
                                    const float *__restrict__ P_ell, int nk, BinArgs b, int np_all,
                                    const int *__restrict__ pole_slot) {
    const int kzlen = n / 2 + 1;
    const int64_t total = (int64_t)n * n * kzlen;
    const float x0 = k_ell[0], x1 = k_ell[nk - 1], dx = k_ell[1] - k_ell[0];
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        int r2, k;
        helper_mode(q, n, kzlen, r2, k);
        const float kmag2 = (float)(r2 + k * k);
        const float mu2 = kmag2 > 0.f ? (float)(k * k) * (1.0f / kmag2) : 0.f;
        const float xd = sqrtf(kmag2) * dk;
        float acc = 0.f;
        for (int ip = 0; ip < np_all; ip++) {
            const float *y = P_ell + (int64_t)ip * nk;
            float yd;
            if (xd <= x0) yd = y[0];
            else if (xd >= x1) yd = y[nk - 1];
            else {
                const float f = (xd - x0) / dx;
                const int fl = (int)f;
                yd = y[fl] + (f - (float)fl) * (y[fl + 1] - y[fl]);
            }
            const int sl = pole_slot[ip];
            if (sl >= 0) {   // P_l(mu) from the (2l+1) P_l polynomial of the binning kernel
                const float *c = b.polecoef[sl];
                float Lq = c[b.poledeg[sl]];
                for (int m = b.poledeg[sl] - 1; m >= 0; m--) Lq = Lq * mu2 + c[m];
                yd *= Lq / (float)(4 * b.poledeg[sl] + 1);
            }
            acc += yd;
        }
        out[q] = acc;
    }
}

// bin_kppi (:303-412): one workgroup per i; the j loop of the reference BREAKS at the first k_perp^2 >= last edge
// (the rest of the row, negative frequencies included, is never visited), kz loop breaks at the last pi edge
__global__ __launch_bounds__(256) void helper_bin_kppi(const float *__restrict__ w, int n, int zdim,
                                                       const float *__restrict__ ke2, int Nk,
                                                       const float *__restrict__ pe2, int Npi,
                                                       unsigned long long *__restrict__ cnt, double *__restrict__ sum) {
    __shared__ int jbreak;
    const int i = blockIdx.x, kzlen = n / 2 + 1;
    const int fi = fold(i, n), i2 = fi * fi;
    if (threadIdx.x == 0) jbreak = n;
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += 256) {
        const int fj = fold(j, n);
        if ((float)(i2 + fj * fj) >= ke2[Nk]) atomicMin(&jbreak, j);
    }
    __syncthreads();
    const int jb = jbreak;
    for (int q = threadIdx.x; q < jb * kzlen; q += 256) {
        const int j = q / kzlen, k = q - j * kzlen;
        const int fj = fold(j, n);
        const float kp2 = (float)(i2 + fj * fj), kz2 = (float)(k * k);
        if (kp2 < ke2[0] || kz2 >= pe2[Npi]) continue;
        int bk = 0, bp = 0;
        while (kp2 > ke2[bk + 1]) bk++;
        while (kz2 > pe2[bp + 1]) bp++;
        const double v = (double)w[((int64_t)i * n + j) * zdim + k];
        atomicAdd(&cnt[bk * Npi + bp], k == 0 ? 1ull : 2ull);
        atomicAdd(&sum[bk * Npi + bp], k == 0 ? v : 2.0 * v);
    }
}

// ---- host side ------------------------------------------------------------------------------------------
struct PowerCtx {
    std::map<int, hipfftHandle> plans;
    std::map<int, hipfftHandle> plans64;    // padded in-place D2Z (float64 meshes of sizes the mixed-radix kernels do not cover)
    DevBuf mesh[4];       // field1, field1 shifted, field2, field2 shifted
    DevBuf W, phase, edges, accum, pos, pos2, w, w2;
    DevBuf helper_in, helper_tab;          // staging of caller-supplied real grids / small tables (ZCV helpers)
    std::map<int, hipfftHandle> c2r_plans;  // contiguous 3-D C2R (pk_to_xi)
    int phase_n = 0;
    std::vector<float> edges_host;         // the float32 squared edges of the last prepare_bins (BinArgs::h_edges2)
    // multi-tracer spectra: delta_k of every tracer kept in HBM (abacus_power_field_* / abacus_power_from_fields)
    static constexpr int NFIELD = 8;
    DevBuf field[NFIELD][2];                // [slot][unshifted, half-cell shifted]
    struct FieldInfo {
        int nmesh = 0, paste = 0, interlaced = 0, fused = 0;
        double Lbox = 0;
        int64_t n = 0;
    } finfo[NFIELD];
};
PowerCtx g_ctx;

// Positions a host entry point has NOT uploaded yet: field_fft_dev uploads them itself - in batches on a copy stream, each
// batch deposited (accumulating into the mesh) while the next is on the PCIe link - when the mesh is small enough for its
// read-modify-write per batch to hide behind a batch's upload (BASELINE config 3: 1.2 GB at 56 GB/s = 21 ms of upload in front
// of 10 ms of kernels; the reference's calc_power takes NumPy positions, analysis/power_spectrum.py:1131)
struct HostSrc {
    const void *host = nullptr;
    float *dev = nullptr;
    int64_t n = 0;
    bool pending = false;
};
HostSrc g_hsrc[2];
hipStream_t g_copy_stream = nullptr;
hipEvent_t g_copy_ev[3] = {nullptr, nullptr, nullptr};   // [0], [1]: a batch has landed; [2]: the library stream's work so far
double g_last_batches = 0;   // diagnostic: batches of the last host upload (abacus_power_last_batches)

int upload_batches(int64_t n, int nmesh, int interlaced) {
    if (option("pk_nobatch")) return 1;
    // the mesh read and written once per batch - both meshes of an interlaced pair
    const double t_up = 12.0 * (double)n / 56e9, t_rmw = (interlaced ? 2.0 : 1.0) * 8.0 * (double)nmesh * nmesh * nmesh / 4.5e12;
    const int K = (int)std::floor(0.8 * t_up / std::max(t_rmw, 1e-9));
    // batches of at least 8e6 particles: the list build of a batch must keep its two levels
    return (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)K, (int64_t)8, n / 8000000}));
}

int fft_check(hipfftResult r, const char *what) {
    if (r != HIPFFT_SUCCESS) return fail("%s failed (hipfftResult %d)", what, (int)r);
    return 0;
}

// real-row pitch (floats): rows start on 128-B boundaries, >= n + 2 for the in-place R2C layout
int pitch_r(int n) { return (n + 2 + 31) / 32 * 32; }

// hipFFT allocates its work area when a plan is made: with idle scratch blocks of this library around, give them back and try
// once more before reporting the failure
template <typename F>
hipfftResult plan_with_retry(F make) {
    hipfftResult r = make();
    if (r != HIPFFT_SUCCESS) {
        (void)hipGetLastError();
        if (scratch_trim_idle() == 0) r = make();
    }
    return r;
}

int get_plan(int n, hipfftHandle *out) {
    auto it = g_ctx.plans.find(n);
    if (it == g_ctx.plans.end()) {
        hipfftHandle h;
        int dims[3] = {n, n, n};
        int inembed[3] = {n, n, pitch_r(n)}, onembed[3] = {n, n, pitch_r(n) / 2};
        ABACUS_TRY(fft_check(plan_with_retry([&] { return hipfftPlanMany(&h, 3, dims, inembed, 1, 1, onembed, 1, 1, HIPFFT_R2C, 1); }),   // batch 1: dist unused
                             "hipfftPlanMany"));
        it = g_ctx.plans.emplace(n, h).first;
    }
    ABACUS_TRY(fft_check(hipfftSetStream(it->second, stream()), "hipfftSetStream"));
    *out = it->second;
    return 0;
}

size_t mesh_bytes(int n) { return (size_t)n * n * pitch_r(n) * sizeof(float); }

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
// `fused`: use fft.hip's fused form (permuted row order; the caller sets SpecArgs::permshift)
bool use_fused_fft(int nmesh) {
    return fft_native_supported(nmesh) && !option("fft_hipfft") && !option("fft_nofuse") &&
           fft_native_fused_supported(nmesh);
}

// pf64: `pos` / `w` point to float64 values (cloud arithmetic in float64, analysis/tsc.py:400; float32 mesh)
int field_fft_dev(float *pos, int64_t n, const float *w, double L, int nmesh, int paste, int interlaced, int slot,
                  bool fused = false, bool skip_x = false, DevBuf *dest = nullptr, int pf64 = 0, float xcut = 0.f) {
    if (!dest) dest = &g_ctx.mesh[slot];
    if (n <= 0) return fail("power: empty particle set");
    const bool native = fft_native_supported(nmesh) && !option("fft_hipfft");
    hipfftHandle plan = 0;
    // hipFFT (ROCm 7.2) returns a wrong spectrum for the padded in-place 2048^3 R2C (shot-noise test: 0.94 of the known
    // answer, where 1296^3..2016^3 give 1.0000); production never sends a power of two there, the debug switch must not
    if (!native && nmesh >= 2048 && fft_native_supported(nmesh))
        return fail("power: ABACUS_FFT_HIPFFT is not usable at nmesh %d (hipFFT returns a wrong spectrum at this size)", nmesh);
    if (!native) ABACUS_TRY(get_plan(nmesh, &plan));
    const int64_t zstride = pitch_r(nmesh);
    const double M = (double)nmesh * nmesh * nmesh;
    const double norm = (double)(float)(M / (double)n);   // dtype(field.size / tot_weight), tot_weight = len(pos) (:856,894)
    const double d = L / nmesh;
    // positions still on the host (power_from_host): upload here, batch by batch where that pays
    HostSrc *hs = nullptr;
    for (HostSrc &h : g_hsrc)
        if (h.pending && h.dev == pos && h.n == n) hs = &h;
    bool deposited = false;
    if (hs) {
        hs->pending = false;
        const int K = (paste == 0 && !w && !pf64 && nmesh % 32 == 0 && nmesh >= 256) ? upload_batches(n, nmesh, interlaced) : 1;
        g_last_batches = K;
        if (K == 1) {
            HIP_TRY(hipMemcpyAsync(pos, hs->host, (size_t)n * 12, hipMemcpyHostToDevice, stream()));
        } else {
            if (!g_copy_stream) {
                HIP_TRY(hipStreamCreateWithFlags(&g_copy_stream, hipStreamNonBlocking));
                for (hipEvent_t &e : g_copy_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            }
            for (int s = 0; s < (interlaced ? 2 : 1); s++) ABACUS_TRY(dest[s].reserve(mesh_bytes(nmesh)));
            // the copy stream is non-blocking: whatever the library stream still does with `pos` comes first
            HIP_TRY(hipEventRecord(g_copy_ev[2], stream()));
            HIP_TRY(hipStreamWaitEvent(g_copy_stream, g_copy_ev[2], 0));
            const int64_t nb = (n + K - 1) / K;
            for (int b = 0; b < K; b++) {
                const int64_t p0 = (int64_t)b * nb, pn = std::min<int64_t>(nb, n - p0);
                if (pn <= 0) break;
                const bool last = p0 + pn >= n;
                // (a copy from pageable memory keeps this thread busy for its duration; the kernels of the batch before run meanwhile)
                HIP_TRY(hipMemcpyAsync(pos + 3 * p0, static_cast<const float *>(hs->host) + 3 * p0, (size_t)pn * 12, hipMemcpyHostToDevice, g_copy_stream));
                HIP_TRY(hipEventRecord(g_copy_ev[b & 1], g_copy_stream));
                HIP_TRY(hipStreamWaitEvent(stream(), g_copy_ev[b & 1], 0));
                // rho accumulates raw over the batches; the last flush applies rho * norm - 1 (deposit kernels: norm 0 = none)
                for (int s = 0; s < (interlaced ? 2 : 1); s++)
                    ABACUS_TRY(tsc_deposit_f32(pos + 3 * p0, pn, nullptr, dest[s].as<float>(), nmesh, zstride, L, s == 0 ? 0.0 : 0.5 * d, 1, last ? norm : 0.0,
                                               0, interlaced ? (s == 0 ? 1 : 2) : 0, 1.0, b == 0 ? 1 : 0));
            }
            deposited = true;
        }
    }
    for (int s = 0; s < (interlaced ? 2 : 1); s++) {
        ABACUS_TRY(dest[s].reserve(mesh_bytes(nmesh)));
        float *mesh = dest[s].as<float>();
        // tsc_parallel wraps pos in place on the first call (tsc.py:171-173); the shifted deposit sees wrapped pos
        // interlaced: the lists of the first deposit are built to cover the half-cell-shifted one as well
        if (deposited) {
        } else if (pf64)
            ABACUS_TRY(tsc_deposit_f64pos(reinterpret_cast<double *>(pos), n, reinterpret_cast<const double *>(w), mesh, nmesh,
                                          zstride, L, s == 0 ? 0.0 : 0.5 * d, paste == 0, norm, paste));
        else
            ABACUS_TRY(tsc_deposit_f32(pos, n, w, mesh, nmesh, zstride, L, s == 0 ? 0.0 : 0.5 * d, paste == 0, norm, paste,
                                       interlaced ? (s == 0 ? 1 : 2) : 0));
        if (native && fused) {
            ABACUS_TRY(skip_x ? fft_native_r2c_fused_zy(mesh, nmesh, (int)zstride, xcut) : fft_native_r2c_fused(mesh, nmesh, (int)zstride, xcut));
        } else if (native && skip_x && !fft_native_pow2(nmesh)) {
            ABACUS_TRY(gfft_r2c_zy_f32(mesh, nmesh, (int)zstride));               // rows and y pass; gfft_x_bin follows
        } else if (native) {
            ABACUS_TRY(fft_native_r2c_inplace(mesh, nmesh, (int)zstride, xcut));   // three passes, one per axis
        } else {
            prof_begin("hipfft_r2c");
            hipfftResult r = hipfftExecR2C(plan, (hipfftReal *)mesh, (hipfftComplex *)mesh);
            prof_end("hipfft_r2c");
            ABACUS_TRY(fft_check(r, "hipfftExecR2C"));
        }
    }
    return 0;
}

void fill_spec(SpecArgs &s, int nmesh, int mode, int interlaced, const float *W_dev, bool cross) {
    s.n = nmesh;
    s.kzlen = nmesh / 2 + 1;
    s.pitch = mode == 1 ? pitch_r(nmesh) / 2 : nmesh / 2 + 1;   // caller-supplied spectra (mode 0) are contiguous
    s.rowmode = 0;
    s.y0 = 0;
    s.permshift = 0;
    s.linear = 0;
    s.nrows = (int64_t)nmesh * nmesh;
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

// normalisation of bin_kmu (:276-293) / calc_pk_from_deltak (:789-792) in float64 from the raw sums
// raw layout: [cnt u64 Nk*Nmu][sum f64 Nk*Nmu][ksum f64 Nk*Nmu][pole f64 Np'*Nk], Np' = poles with l != 0 in order
// dk: unit of |k| for k_avg (0: 2 pi / L); scale: factor on the means (0: L^3, calc_pk_from_deltak :789-792)
int finalize_bins(const void *raw, double Lbox, int Nk, int Nmu, const int64_t *poles, int Np_all, float *power,
                  int64_t *N_mode, float *binned_poles, int64_t *N_mode_poles, float *k_avg, double dk_ = 0, double scale = 0) {
    const size_t nb = (size_t)Nk * Nmu;
    const unsigned long long *cnt = reinterpret_cast<const unsigned long long *>(raw);
    const double *sum = reinterpret_cast<const double *>(cnt + nb);
    const double *ksum = sum + nb;
    const double *pole = ksum + nb;
    const double dk = dk_ > 0 ? dk_ : 2.0 * M_PI / Lbox;
    const double L3 = scale > 0 ? scale : Lbox * Lbox * Lbox;
    int nz_index[MAX_POLES], nnz = 0;
    for (int q = 0; q < Np_all; q++) nz_index[q] = poles[q] != 0 ? nnz++ : -1;
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

// host side of a binning launch: pole coefficients, squared float32 edges (:217-218) and zeroed accumulators on the device
// lo_excl / hi_incl (a k range binned in several passes, run_bin): the first edge of a pass that does not start the whole
// range is raised by one float32 step - `v < edge'` is `v <= edge`, its modes ON the edge belong to the pass before, whose
// last bin keeps them because ITS last edge is raised too (`v >= edge'` is `v > edge`)
int prepare_bins(double Lbox, const double *kedges, int Nk, const double *muedges, int Nmu, const int64_t *poles, int Np_all,
                 double dk_, BinArgs &b, size_t &acc_bytes, int lo_excl = 0, int hi_incl = 0) {
    if (Nk < 1 || Nmu < 1) return fail("power: need at least one k bin and one mu bin");
    if (Np_all > MAX_POLES) return fail("power: more than %d multipoles requested", MAX_POLES);
    const double dk = dk_ > 0 ? dk_ : 2.0 * M_PI / Lbox;
    b.Nk = Nk;
    b.Nmu = Nmu;
    b.dbg = option("dbg");
    b.Np = 0;
    for (int q = 0; q < Np_all; q++)
        if (poles[q] != 0) {   // requested pole -> slot among the ell != 0 accumulators, in order
            ABACUS_TRY(pole_coefs((int)poles[q], b.polecoef[b.Np]));
            b.poledeg[b.Np] = (int)poles[q] / 2;
            b.Np++;
        }
    // edges in units of dk, squared, float32 (:217-218)
    std::vector<float> e2((size_t)Nk + 1 + Nmu + 1);
    for (int q = 0; q <= Nk; q++) e2[q] = (float)((kedges[q] / dk) * (kedges[q] / dk));
    for (int q = 0; q <= Nmu; q++) e2[Nk + 1 + q] = (float)(muedges[q] * muedges[q]);
    if (lo_excl) e2[0] = nextafterf(e2[0], INFINITY);
    if (hi_incl) e2[Nk] = nextafterf(e2[Nk], INFINITY);
    ABACUS_TRY(g_ctx.edges.reserve(e2.size() * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(g_ctx.edges.p, e2.data(), e2.size() * sizeof(float), hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));   // e2 is a local
    g_ctx.edges_host.swap(e2);
    b.kedges2 = g_ctx.edges.as<float>();
    b.muedges2 = b.kedges2 + Nk + 1;
    b.h_edges2 = g_ctx.edges_host.data();
    const size_t nb = (size_t)Nk * Nmu, npk = (size_t)b.Np * Nk;
    acc_bytes = nb * 8 * 3 + npk * 8;
    ABACUS_TRY(g_ctx.accum.reserve(acc_bytes));
    HIP_TRY(hipMemsetAsync(g_ctx.accum.p, 0, acc_bytes, stream()));
    b.g_cnt = g_ctx.accum.as<unsigned long long>();
    b.g_sum = reinterpret_cast<double *>(b.g_cnt + nb);
    b.g_ksum = b.g_sum + nb;
    b.g_pole = b.g_ksum + nb;
    return 0;
}

// tiny read-back of the raw sums - counts (u64), sum P, sum k, pole sums (f64) - and bin_kmu's normalisation
int collect_bins(size_t acc_bytes, double Lbox, int Nk, int Nmu, const int64_t *poles, int Np_all, float *power,
                 int64_t *N_mode, float *binned_poles, int64_t *N_mode_poles, float *k_avg, void *raw_out, double dk, double scale) {
    if (raw_out) {
        HIP_TRY(hipMemcpyAsync(raw_out, g_ctx.accum.p, acc_bytes, hipMemcpyDeviceToHost, stream()));
        HIP_TRY(hipStreamSynchronize(stream()));
        return 0;
    }
    std::vector<unsigned char> host(acc_bytes);
    HIP_TRY(hipMemcpyAsync(host.data(), g_ctx.accum.p, acc_bytes, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return finalize_bins(host.data(), Lbox, Nk, Nmu, poles, Np_all, power, N_mode, binned_poles, N_mode_poles, k_avg, dk,
                         scale);
}

// LDS budget of one spectrum_bin launch: histogram + edges + tile
struct BinBudget {
    size_t hist_bytes, tile_bytes;
    int64_t tile_modes;
    static constexpr size_t lds_max = 160 * 1024;
    bool fits() const { return hist_bytes + tile_bytes <= lds_max; }
};
BinBudget bin_budget(const SpecArgs &s, int Nk, int Nmu, int Np_nz) {
    BinBudget g;
    const size_t nb = (size_t)Nk * Nmu, npk = (size_t)Np_nz * Nk;
    g.hist_bytes = nb * (8 + 8 + 4) + npk * 8 + (size_t)(Nk + 1 + Nmu + 1) * 4 + 64;
    const bool inter = s.mode == 1 && s.interlaced, cross = s.cross != 0;
    const int ept = (inter && cross) ? 4 : ((inter || cross) ? 8 : 16);
    g.tile_modes = (int64_t)BIN_THREADS * ept;
    g.tile_bytes = (size_t)(g.tile_modes + BIN_THREADS + 16) * 4;
    return g;
}

int run_bin_once(const SpecArgs &s_in, double Lbox, const double *kedges, int Nk, const double *muedges, int Nmu,
                 const int64_t *poles, int Np_all, float *power, int64_t *N_mode, float *binned_poles,
                 int64_t *N_mode_poles, float *k_avg, void *raw_out, double dk_, double scale, int lo_excl, int hi_incl) {
    SpecArgs s = s_in;
    s.lds_tables = 0;
    const double dk = dk_ > 0 ? dk_ : 2.0 * M_PI / Lbox;
    BinArgs b;
    size_t acc_bytes = 0;
    ABACUS_TRY(prepare_bins(Lbox, kedges, Nk, muedges, Nmu, poles, Np_all, dk_, b, acc_bytes, lo_excl, hi_incl));
    const BinBudget bud = bin_budget(s, Nk, Nmu, b.Np);
    const size_t hist_bytes = bud.hist_bytes, tile_bytes = bud.tile_bytes, lds_max = BinBudget::lds_max;
    const int64_t tile_modes = bud.tile_modes;
    const bool inter = s.mode == 1 && s.interlaced, cross = s.cross != 0;
    if (!bud.fits())
        return fail("power: %d x %d bins with %d multipoles do not fit the 160 KiB LDS histogram", Nk, Nmu, b.Np);
    size_t table_bytes = 0;
    if (s.mode == 1) table_bytes = (inter ? (size_t)2 * s.n * 8 : 0) + (s.compensated ? (size_t)s.n * 4 : 0);
    s.lds_tables = table_bytes > 0 && hist_bytes + tile_bytes + table_bytes + 16 <= lds_max;
    const size_t lds = hist_bytes + tile_bytes + (s.lds_tables ? table_bytes + 16 : 0);
    int dev = 0, ncu = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t ntiles = (s.nrows * s.pitch + tile_modes - 1) / tile_modes;
    const int grid = (int)std::min<int64_t>(ntiles, ncu);
    int maxdeg = 0;
    for (int q = 0; q < b.Np; q++) maxdeg = std::max(maxdeg, b.poledeg[q]);
    if (maxdeg > 5) return fail("power: multipole order beyond the coefficient table");
#define LAUNCH_BIN(I, C, P, D)                                                                                       \
    do {                                                                                                             \
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(spectrum_bin<I, C, P, D>),                        \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                          \
        ABACUS_LAUNCH("spectrum_bin", (spectrum_bin<I, C, P, D>), dim3(grid), dim3(BIN_THREADS), lds, s, b);         \
    } while (0)
#define LAUNCH_NP(I, C)                                              \
    do {                                                             \
        if (b.Np == 0) LAUNCH_BIN(I, C, 0, 2);                       \
        else if (b.Np == 1 && maxdeg <= 2) LAUNCH_BIN(I, C, 1, 2);   \
        else if (b.Np == 2 && maxdeg <= 2) LAUNCH_BIN(I, C, 2, 2);   \
        else LAUNCH_BIN(I, C, -1, -1);                               \
    } while (0)
    if (inter && cross) LAUNCH_NP(true, true);
    else if (inter) LAUNCH_NP(true, false);
    else if (cross) LAUNCH_NP(false, true);
    else LAUNCH_NP(false, false);
#undef LAUNCH_NP
#undef LAUNCH_BIN
    return collect_bins(acc_bytes, Lbox, Nk, Nmu, poles, Np_all, power, N_mode, binned_poles, N_mode_poles, k_avg, raw_out, dk,
                        scale);
}

// The workgroup histogram of spectrum_bin lives in LDS (a few thousand (k, mu) bins).  A finer binning is done in several
// passes over the spectrum, each with a run of consecutive k bins that fits - the reference takes any number of bins
// (power_spectrum.py:150-300).  Modes on an edge shared by two passes are counted once, where one pass would count them
// (prepare_bins).  Raw sums (`raw_out`, the multi-GPU reduction) are assembled in the one-pass layout.
int run_bin(const SpecArgs &s, double Lbox, const double *kedges, int Nk, const double *muedges, int Nmu,
            const int64_t *poles, int Np_all, float *power, int64_t *N_mode, float *binned_poles,
            int64_t *N_mode_poles, float *k_avg, void *raw_out = nullptr, double dk_ = 0, double scale = 0) {
    int Np_nz = 0;
    for (int q = 0; q < Np_all && q < MAX_POLES; q++) Np_nz += poles[q] != 0;
    if (Nk < 1 || Nmu < 1 || bin_budget(s, Nk, Nmu, Np_nz).fits())
        return run_bin_once(s, Lbox, kedges, Nk, muedges, Nmu, poles, Np_all, power, N_mode, binned_poles, N_mode_poles, k_avg,
                            raw_out, dk_, scale, 0, 0);
    int kc = Nk;
    while (kc > 1 && !bin_budget(s, kc, Nmu, Np_nz).fits()) kc = (kc + 1) / 2;
    if (!bin_budget(s, kc, Nmu, Np_nz).fits())
        return fail("power: %d mu bins with %d multipoles do not fit the 160 KiB LDS histogram even for one k bin", Nmu, Np_nz);
    const size_t nb = (size_t)Nk * Nmu;
    unsigned char *raw = static_cast<unsigned char *>(raw_out);
    for (int a = 0; a < Nk; a += kc) {
        const int n = std::min(kc, Nk - a);
        const size_t nbc = (size_t)n * Nmu;
        if (raw) {
            std::vector<unsigned char> part(nbc * 24 + (size_t)Np_nz * n * 8);
            ABACUS_TRY(run_bin_once(s, Lbox, kedges + a, n, muedges, Nmu, poles, Np_all, nullptr, nullptr, nullptr, nullptr, nullptr,
                                    part.data(), dk_, scale, a > 0, a + n < Nk));
            for (int f = 0; f < 3; f++)   // cnt | sum | ksum: (Nk, Nmu) each
                memcpy(raw + ((size_t)f * nb + (size_t)a * Nmu) * 8, part.data() + (size_t)f * nbc * 8, nbc * 8);
            for (int q = 0; q < Np_nz; q++)   // pole sums: (Np', Nk)
                memcpy(raw + (3 * nb + (size_t)q * Nk + a) * 8, part.data() + (3 * nbc + (size_t)q * n) * 8, (size_t)n * 8);
            continue;
        }
        std::vector<float> bp((size_t)std::max(Np_all, 1) * n);
        ABACUS_TRY(run_bin_once(s, Lbox, kedges + a, n, muedges, Nmu, poles, Np_all, power + (size_t)a * Nmu, N_mode + (size_t)a * Nmu,
                                bp.data(), N_mode_poles + a, k_avg + (size_t)a * Nmu, nullptr, dk_, scale, a > 0, a + n < Nk));
        for (int q = 0; q < Np_all; q++) memcpy(binned_poles + (size_t)q * Nk + a, bp.data() + (size_t)q * n, (size_t)n * sizeof(float));
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

int power_dev_once(float *pos, int64_t n, const float *w, float *pos2, int64_t n2, const float *w2, double Lbox, int nmesh,
                   int paste, const float *W_host, int interlaced, const double *kedges, int Nk, const double *muedges,
                   int Nmu, const int64_t *poles, int Np, float *power, int64_t *N_mode, float *binned_poles,
                   int64_t *N_mode_poles, float *k_avg, int pf64);

// The deposits of a spectrum run their list builds in deferred mode (tsc.hip): buffers sized from the previous build of the
// same mesh, tables made on the device, no synchronise inside the deposit.  Every path below ends with the read-back of the
// binned sums - a stream synchronise -, after which the builds' own needs are known: if one did not fit (a catalogue far more
// clustered than the last one), the spectrum is computed again with exact sizes.
int power_dev(float *pos, int64_t n, const float *w, float *pos2, int64_t n2, const float *w2, double Lbox, int nmesh,
              int paste, const float *W_host, int interlaced, const double *kedges, int Nk, const double *muedges,
              int Nmu, const int64_t *poles, int Np, float *power, int64_t *N_mode, float *binned_poles,
              int64_t *N_mode_poles, float *k_avg, int pf64 = 0) {
    tsc_lines_defer(1);
    int rc = power_dev_once(pos, n, w, pos2, n2, w2, Lbox, nmesh, paste, W_host, interlaced, kedges, Nk, muedges, Nmu, poles, Np, power, N_mode,
                            binned_poles, N_mode_poles, k_avg, pf64);
    tsc_lines_defer(0);
    if (rc != 0) {
        (void)hipStreamSynchronize(stream());
        (void)tsc_lines_deferred_check();
        return rc;
    }
    if (tsc_lines_deferred_check())
        rc = power_dev_once(pos, n, w, pos2, n2, w2, Lbox, nmesh, paste, W_host, interlaced, kedges, Nk, muedges, Nmu, poles, Np, power, N_mode,
                            binned_poles, N_mode_poles, k_avg, pf64);
    return rc;
}

int power_dev_once(float *pos, int64_t n, const float *w, float *pos2, int64_t n2, const float *w2, double Lbox, int nmesh,
                   int paste, const float *W_host, int interlaced, const double *kedges, int Nk, const double *muedges,
                   int Nmu, const int64_t *poles, int Np, float *power, int64_t *N_mode, float *binned_poles,
                   int64_t *N_mode_poles, float *k_avg, int pf64) {
    ABACUS_TRY(check_common(nmesh, paste));
    ABACUS_TRY(ensure_phase(nmesh));
    const float *W_dev;
    ABACUS_TRY(upload_W(W_host, nmesh, &W_dev));
    const bool fused = use_fused_fft(nmesh);
    const bool cross = pos2 != nullptr;
    // the spectra of this call feed the binning below and nothing else: modes beyond its last k edge are never read, so the
    // y pass need not write, and the x pass need not transform, what lies entirely beyond it (margin: one part in 1e5 + 1
    // against the float32 edge test of the binning)
    float xcut = 0.f;
    if ((fused || !fft_native_pow2(nmesh)) && Nk > 0) {
        const double e = kedges[Nk] / (2.0 * M_PI / Lbox);
        xcut = (float)(e * e * (1.0 + 1e-5) + 1.0);
    }
    if (fused && !cross && !option("pk_noxbin") && !(interlaced && option("pk_noxbin_inter"))) {
        // auto power of one field (or of its interlaced pair): the last FFT pass bins straight from LDS (xbin.hip) - no
        // spectrum write + re-read; the interlaced pair needs the cached-geometry kernel
        BinArgs b;
        size_t acc_bytes = 0;
        ABACUS_TRY(prepare_bins(Lbox, kedges, Nk, muedges, Nmu, poles, Np, 0, b, acc_bytes));
        if (interlaced ? xbin2_supported(nmesh, b, W_dev != nullptr) : xbin_supported(nmesh, Nk, Nmu, b, W_dev != nullptr)) {
            ABACUS_TRY(field_fft_dev(pos, n, w, Lbox, nmesh, paste, interlaced, 0, true, /*skip_x=*/true, nullptr, pf64, xcut));
            const double M = (double)nmesh * nmesh * nmesh;
            ABACUS_TRY(fft_x_bin_run(g_ctx.mesh[0].as<float>(), nmesh, pitch_r(nmesh), (float)(1.0 / M), W_dev, b, b.dbg, 0, 0, 1, 0, 1,
                                     interlaced ? g_ctx.mesh[1].as<float>() : nullptr, g_ctx.phase.as<float2>()));
            return collect_bins(acc_bytes, Lbox, Nk, Nmu, poles, Np, power, N_mode, binned_poles, N_mode_poles, k_avg, nullptr,
                                2.0 * M_PI / Lbox, 0);
        }
    }
    if (fused && cross && !interlaced && !option("pk_noxbin") && !option("pk_noxbin_cross")) {
        // cross power of two non-interlaced fields: both stop after their y pass, the last pass runs over the pair of tiles and
        // bins Re(conj(a) b) from LDS (fft_x_bin2<.., CROSS>) - two spectrum writes and two re-reads less
        BinArgs b;
        size_t acc_bytes = 0;
        ABACUS_TRY(prepare_bins(Lbox, kedges, Nk, muedges, Nmu, poles, Np, 0, b, acc_bytes));
        if (xbin2_supported(nmesh, b, W_dev != nullptr)) {
            ABACUS_TRY(field_fft_dev(pos, n, w, Lbox, nmesh, paste, 0, 0, true, /*skip_x=*/true, nullptr, pf64, xcut));
            ABACUS_TRY(field_fft_dev(pos2, n2, w2, Lbox, nmesh, paste, 0, 2, true, /*skip_x=*/true, nullptr, pf64, xcut));
            const double M = (double)nmesh * nmesh * nmesh;
            ABACUS_TRY(fft_x_bin_run(g_ctx.mesh[0].as<float>(), nmesh, pitch_r(nmesh), (float)(1.0 / M), W_dev, b, b.dbg, 0, 0, 1, 0, 1,
                                     g_ctx.mesh[2].as<float>(), nullptr));
            return collect_bins(acc_bytes, Lbox, Nk, Nmu, poles, Np, power, N_mode, binned_poles, N_mode_poles, k_avg, nullptr,
                                2.0 * M_PI / Lbox, 0);
        }
    }
    if (fused && cross && interlaced && !option("pk_noxbin") && !option("pk_noxbin_cross") && !option("pk_noxbin_inter")) {
        // cross power of two INTERLACED fields (calc_power's defaults with pos2): all four meshes stop after their y pass and one
        // last pass takes the four tiles through LDS (fft_x_bin2<.., QUAD>) - four spectrum writes and four re-reads less
        BinArgs b;
        size_t acc_bytes = 0;
        ABACUS_TRY(prepare_bins(Lbox, kedges, Nk, muedges, Nmu, poles, Np, 0, b, acc_bytes));
        if (xbin2_supported(nmesh, b, W_dev != nullptr)) {
            ABACUS_TRY(field_fft_dev(pos, n, w, Lbox, nmesh, paste, 1, 0, true, /*skip_x=*/true, nullptr, pf64, xcut));
            ABACUS_TRY(field_fft_dev(pos2, n2, w2, Lbox, nmesh, paste, 1, 2, true, /*skip_x=*/true, nullptr, pf64, xcut));
            const double M = (double)nmesh * nmesh * nmesh;
            ABACUS_TRY(fft_x_bin_run(g_ctx.mesh[0].as<float>(), nmesh, pitch_r(nmesh), (float)(1.0 / M), W_dev, b, b.dbg, 0, 0, 1, 0, 1,
                                     g_ctx.mesh[1].as<float>(), g_ctx.phase.as<float2>(), nullptr, 0, g_ctx.mesh[2].as<float>(),
                                     g_ctx.mesh[3].as<float>()));
            return collect_bins(acc_bytes, Lbox, Nk, Nmu, poles, Np, power, N_mode, binned_poles, N_mode_poles, k_avg, nullptr,
                                2.0 * M_PI / Lbox, 0);
        }
    }
    if (!fused && !cross && !option("pk_noxbin") && !(interlaced && option("pk_noxbin_inter")) && !option("fft_hipfft") && !fft_native_pow2(nmesh) &&
        gfft_supported(nmesh, 0)) {
        // mixed-radix meshes (compute_power's default 550, 768 ...): the same fusion on gfft's natural-order x pass (gfft.hip);
        // an interlaced pair where two tiles fit the LDS (n up to ~900)
        BinArgs b;
        size_t acc_bytes = 0;
        ABACUS_TRY(prepare_bins(Lbox, kedges, Nk, muedges, Nmu, poles, Np, 0, b, acc_bytes));
        if (gfft_xbin_supported(nmesh, b, W_dev != nullptr, interlaced != 0)) {
            ABACUS_TRY(field_fft_dev(pos, n, w, Lbox, nmesh, paste, interlaced, 0, false, /*skip_x=*/true, nullptr, pf64, 0.f));
            const double M = (double)nmesh * nmesh * nmesh;
            ABACUS_TRY(gfft_x_bin_run(g_ctx.mesh[0].as<float>(), nmesh, pitch_r(nmesh), (float)(1.0 / M), W_dev, b,
                                      interlaced ? g_ctx.mesh[1].as<float>() : nullptr, g_ctx.phase.as<float2>()));
            return collect_bins(acc_bytes, Lbox, Nk, Nmu, poles, Np, power, N_mode, binned_poles, N_mode_poles, k_avg, nullptr,
                                2.0 * M_PI / Lbox, 0);
        }
    }
    ABACUS_TRY(field_fft_dev(pos, n, w, Lbox, nmesh, paste, interlaced, 0, fused, false, nullptr, pf64, xcut));
    if (cross) ABACUS_TRY(field_fft_dev(pos2, n2, w2, Lbox, nmesh, paste, interlaced, 2, fused, false, nullptr, pf64, xcut));
    SpecArgs s;
    fill_spec(s, nmesh, 1, interlaced, W_dev, cross);
    if (fused) {
        s.permshift = 0;
        while ((2 << s.permshift) < nmesh) s.permshift++;   // log2(nmesh / 2)
    }
    s.a = g_ctx.mesh[0].as<float2>();
    s.as = interlaced ? g_ctx.mesh[1].as<float2>() : nullptr;
    s.b = cross ? g_ctx.mesh[2].as<float2>() : nullptr;
    s.bs = cross && interlaced ? g_ctx.mesh[3].as<float2>() : nullptr;
    return run_bin(s, Lbox, kedges, Nk, muedges, Nmu, poles, Np, power, N_mode, binned_poles, N_mode_poles, k_avg);
}

// the galaxy columns of the HOD (float64 x | y | z in HBM) as the (n, 3) float32 array calc_power works on: the cast
// `np.stack((x, y, z), axis=1)` + float32 conversion of hod/abacus_hod.py:1405-1409 / analysis/power_spectrum.py
__global__ void pack_pos_soa64(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                               int64_t n, float *__restrict__ pos) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        pos[3 * i] = (float)x[i];
        pos[3 * i + 1] = (float)y[i];
        pos[3 * i + 2] = (float)z[i];
    }
}

// upload a host particle set into the context buffers; wrapped positions are copied back like the reference mutates them
int stage_particles(float *pos, int64_t n, const float *w, DevBuf &dpos, DevBuf &dw, float **pd, float **wd, size_t es = 4, HostSrc *later = nullptr) {
    // es: bytes per value (4 float32, 8 float64 - the pointers are then float64 arrays in disguise)
    ABACUS_TRY(dpos.reserve((size_t)std::max<int64_t>(n, 1) * 3 * es));
    if (later && es == 4 && !w) {      // field_fft_dev uploads (in batches, behind the deposits) - see HostSrc
        later->host = pos, later->dev = dpos.as<float>(), later->n = n, later->pending = true;
    } else {
        if (later) later->pending = false;
        HIP_TRY(hipMemcpyAsync(dpos.p, pos, (size_t)n * 3 * es, hipMemcpyHostToDevice, stream()));
    }
    *pd = dpos.as<float>();
    *wd = nullptr;
    if (w) {
        ABACUS_TRY(dw.reserve((size_t)std::max<int64_t>(n, 1) * es));
        HIP_TRY(hipMemcpyAsync(dw.p, w, (size_t)n * es, hipMemcpyHostToDevice, stream()));
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
    ABACUS_ENTER();
    return power_dev(pos, n, w, pos2, n2, w2, Lbox, nmesh, paste, W_host, interlaced, kedges, Nk, muedges, Nmu, poles,
                     Np, power, N_mode, binned_poles, N_mode_poles, k_avg);
}

int abacus_power_field_soa64(int slot, const double *x, const double *y, const double *z, int64_t n, double Lbox, int nmesh,
                             int paste, int interlaced) {
    ABACUS_ENTER();
    if (slot < 0 || slot >= PowerCtx::NFIELD) return fail("abacus_power_field_soa64: slot %d out of range [0, %d)", slot, PowerCtx::NFIELD);
    if (!x || !y || !z || n <= 0) return fail("abacus_power_field_soa64: empty catalogue");
    ABACUS_TRY(check_common(nmesh, paste));
    ABACUS_TRY(g_ctx.pos.reserve((size_t)n * 12));
    const int grid = (int)std::min<int64_t>(std::max<int64_t>(ceil_div(n, 256), 1), 256 * 32);
    ABACUS_LAUNCH("pack_pos", pack_pos_soa64, dim3(grid), dim3(256), 0, x, y, z, n, g_ctx.pos.as<float>());
    const bool fused = use_fused_fft(nmesh);
    ABACUS_TRY(field_fft_dev(g_ctx.pos.as<float>(), n, nullptr, Lbox, nmesh, paste, interlaced, 0, fused, false, g_ctx.field[slot]));
    PowerCtx::FieldInfo &f = g_ctx.finfo[slot];
    f.nmesh = nmesh, f.paste = paste, f.interlaced = interlaced, f.fused = fused ? 1 : 0, f.Lbox = Lbox, f.n = n;
    return 0;
}

int abacus_power_from_fields(int slot_a, int slot_b, const float *W_host, const double *kedges, int Nk, const double *muedges,
                             int Nmu, const int64_t *poles, int Np, float *power, int64_t *N_mode, float *binned_poles,
                             int64_t *N_mode_poles, float *k_avg) {
    ABACUS_ENTER();
    if (slot_a < 0 || slot_a >= PowerCtx::NFIELD || slot_b < 0 || slot_b >= PowerCtx::NFIELD)
        return fail("abacus_power_from_fields: slot out of range");
    const PowerCtx::FieldInfo &fa = g_ctx.finfo[slot_a], &fb = g_ctx.finfo[slot_b];
    if (!fa.nmesh || !fb.nmesh) return fail("abacus_power_from_fields: a field slot is empty (abacus_power_field_soa64 first)");
    if (fa.nmesh != fb.nmesh || fa.interlaced != fb.interlaced || fa.fused != fb.fused || fa.Lbox != fb.Lbox || fa.paste != fb.paste)
        return fail("abacus_power_from_fields: the two fields were built with different mesh settings");
    const int nmesh = fa.nmesh;
    ABACUS_TRY(ensure_phase(nmesh));
    const float *W_dev;
    ABACUS_TRY(upload_W(W_host, nmesh, &W_dev));
    const bool cross = slot_a != slot_b;
    SpecArgs s;
    fill_spec(s, nmesh, 1, fa.interlaced, W_dev, cross);
    if (fa.fused) {
        s.permshift = 0;
        while ((2 << s.permshift) < nmesh) s.permshift++;   // log2(nmesh / 2)
    }
    s.a = g_ctx.field[slot_a][0].as<float2>();
    s.as = fa.interlaced ? g_ctx.field[slot_a][1].as<float2>() : nullptr;
    s.b = cross ? g_ctx.field[slot_b][0].as<float2>() : nullptr;
    s.bs = cross && fa.interlaced ? g_ctx.field[slot_b][1].as<float2>() : nullptr;
    return run_bin(s, fa.Lbox, kedges, Nk, muedges, Nmu, poles, Np, power, N_mode, binned_poles, N_mode_poles, k_avg);
}

int abacus_power_fields_release(void) {
    ABACUS_ENTER();
    for (int q = 0; q < PowerCtx::NFIELD; q++) {
        ABACUS_TRY(g_ctx.field[q][0].release());
        ABACUS_TRY(g_ctx.field[q][1].release());
        g_ctx.finfo[q] = PowerCtx::FieldInfo();
    }
    return 0;
}

static int power_from_host(void *pos, int64_t n, const void *w, void *pos2, int64_t n2, const void *w2, int pf64, double Lbox,
                           int nmesh, int paste, const float *W_host, int interlaced, const double *kedges, int Nk,
                           const double *muedges, int Nmu, const int64_t *poles, int Np, float *power, int64_t *N_mode,
                           float *binned_poles, int64_t *N_mode_poles, float *k_avg) {
    const size_t es = pf64 ? 8 : 4;
    float *pd, *wd, *pd2 = nullptr, *wd2 = nullptr;
    ABACUS_TRY(stage_particles((float *)pos, n, (const float *)w, g_ctx.pos, g_ctx.w, &pd, &wd, es, &g_hsrc[0]));
    if (pos2) ABACUS_TRY(stage_particles((float *)pos2, n2, (const float *)w2, g_ctx.pos2, g_ctx.w2, &pd2, &wd2, es, &g_hsrc[1]));
    tsc_wrapped_reset();
    const int rc = power_dev(pd, n, wd, pd2, n2, wd2, Lbox, nmesh, paste, W_host, interlaced, kedges, Nk, muedges, Nmu,
                             poles, Np, power, N_mode, binned_poles, N_mode_poles, k_avg, pf64);
    g_hsrc[0].pending = g_hsrc[1].pending = false;
    ABACUS_TRY(rc);
    // TSC wraps the caller's positions in place (tsc.py:171-173); CIC does not wrap (cic.py).  Nothing to copy back when
    // every position already lay inside the box
    if (paste == 0 && tsc_wrapped_seen()) {
        HIP_TRY(hipMemcpyAsync(pos, pd, (size_t)n * 3 * es, hipMemcpyDeviceToHost, stream()));
        if (pos2) HIP_TRY(hipMemcpyAsync(pos2, pd2, (size_t)n2 * 3 * es, hipMemcpyDeviceToHost, stream()));
        HIP_TRY(hipStreamSynchronize(stream()));
    }
    return 0;
}

int abacus_power_from_particles(float *pos, int64_t n, const float *w, float *pos2, int64_t n2, const float *w2,
                                double Lbox, int nmesh, int paste, const float *W_host, int interlaced,
                                const double *kedges, int Nk, const double *muedges, int Nmu, const int64_t *poles,
                                int Np, float *power, int64_t *N_mode, float *binned_poles, int64_t *N_mode_poles,
                                float *k_avg) {
    ABACUS_ENTER();
    return power_from_host(pos, n, w, pos2, n2, w2, 0, Lbox, nmesh, paste, W_host, interlaced, kedges, Nk, muedges, Nmu, poles,
                           Np, power, N_mode, binned_poles, N_mode_poles, k_avg);
}

int abacus_power_from_particles_f64(double *pos, int64_t n, const double *w, double *pos2, int64_t n2, const double *w2,
                                    double Lbox, int nmesh, int paste, const float *W_host, int interlaced,
                                    const double *kedges, int Nk, const double *muedges, int Nmu, const int64_t *poles,
                                    int Np, float *power, int64_t *N_mode, float *binned_poles, int64_t *N_mode_poles,
                                    float *k_avg) {
    ABACUS_ENTER();
    return power_from_host(pos, n, w, pos2, n2, w2, 1, Lbox, nmesh, paste, W_host, interlaced, kedges, Nk, muedges, Nmu, poles,
                           Np, power, N_mode, binned_poles, N_mode_poles, k_avg);
}

int abacus_field(float *pos, int64_t n, const float *w, double Lbox, int nmesh, int paste, double offset, float *field) {
    ABACUS_ENTER();
    ABACUS_TRY(check_common(nmesh, paste));
    if (n <= 0) return fail("power: empty particle set");
    if (!field) return fail("abacus_field: null output");
    float *pd, *wd;
    ABACUS_TRY(stage_particles(pos, n, w, g_ctx.pos, g_ctx.w, &pd, &wd));
    ABACUS_TRY(g_ctx.mesh[0].reserve(mesh_bytes(nmesh)));
    float *mesh = g_ctx.mesh[0].as<float>();
    const int64_t zstride = pitch_r(nmesh);
    const double M = (double)nmesh * nmesh * nmesh;
    const double norm = (double)(float)(M / (double)n);   // dtype(field.size / tot_weight), tot_weight = len(pos) (:856,894)
    tsc_wrapped_reset();
    ABACUS_TRY(tsc_deposit_f32(pd, n, wd, mesh, nmesh, zstride, Lbox, offset, paste == 0, norm, paste));
    HIP_TRY(hipMemcpy2DAsync(field, (size_t)nmesh * 4, mesh, (size_t)zstride * 4, (size_t)nmesh * 4, (size_t)nmesh * nmesh,
                             hipMemcpyDeviceToHost, stream()));
    if (paste == 0 && tsc_wrapped_seen()) HIP_TRY(hipMemcpyAsync(pos, pd, (size_t)n * 12, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_field_fft(float *pos, int64_t n, const float *w, double Lbox, int nmesh, int paste, const float *W_host,
                     int interlaced, void *out_c64) {
    ABACUS_ENTER();
    ABACUS_TRY(check_common(nmesh, paste));
    ABACUS_TRY(ensure_phase(nmesh));
    const float *W_dev;
    ABACUS_TRY(upload_W(W_host, nmesh, &W_dev));
    float *pd, *wd;
    ABACUS_TRY(stage_particles(pos, n, w, g_ctx.pos, g_ctx.w, &pd, &wd));
    tsc_wrapped_reset();
    ABACUS_TRY(field_fft_dev(pd, n, wd, Lbox, nmesh, paste, interlaced, 0));
    SpecArgs s;
    fill_spec(s, nmesh, 1, interlaced, W_dev, false);
    s.a = g_ctx.mesh[0].as<float2>();
    s.as = interlaced ? g_ctx.mesh[1].as<float2>() : nullptr;
    s.b = s.bs = nullptr;
    const int64_t total = (int64_t)nmesh * nmesh * (nmesh / 2 + 1);
    const int grid = (int)std::min<int64_t>(ceil_div(total, 256), 256 * 32);
    ABACUS_LAUNCH("spectrum_apply", spectrum_apply, dim3(grid), dim3(256), 0, s, g_ctx.mesh[0].as<float2>());
    const size_t kz = (size_t)nmesh / 2 + 1;   // rows are pitched on the device, contiguous for the caller
    HIP_TRY(hipMemcpy2DAsync(out_c64, kz * 8, g_ctx.mesh[0].p, (size_t)s.pitch * 8, kz * 8, (size_t)nmesh * nmesh,
                             hipMemcpyDeviceToHost, stream()));
    if (paste == 0 && tsc_wrapped_seen()) HIP_TRY(hipMemcpyAsync(pos, pd, (size_t)n * 12, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_pk_from_deltak(const void *field, const void *field2, int nmesh, double Lbox, const double *kedges, int Nk,
                          const double *muedges, int Nmu, const int64_t *poles, int Np, float *power, int64_t *N_mode,
                          float *binned_poles, int64_t *N_mode_poles, float *k_avg) {
    ABACUS_ENTER();
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


int abacus_slab_pitch(int nmesh) { return pitch_r(nmesh); }

int abacus_slab_deposit_padded_dev(float *pos, int64_t n, const float *w, float *mesh, int nmesh, int xoff, int xoff2, int nx_local,
                                   double Lbox, double offset, double norm, int paste, double sub, int nx_alloc) {
    ABACUS_ENTER();
    ABACUS_TRY(check_common(nmesh, paste));
    if (!fft_native_pow2(nmesh)) return fail("slab path: nmesh must be a power of two in [64, 2048]");
    if (xoff < 0 || xoff >= nmesh || xoff2 >= nmesh || nx_local < 1) return fail("abacus_slab_deposit_dev: window [%d | %d, +%d)", xoff, xoff2, nx_local);
    if (xoff2 < 0 && nx_local == nmesh && xoff == 0)   // the whole periodic mesh on one rank: the single-GPU deposit (fast list build)
        return tsc_deposit_f32(pos, n, w, mesh, nmesh, pitch_r(nmesh), Lbox, offset, paste == 0, norm, paste, 0, sub);
    return tsc_deposit_slab_f32(pos, n, w, mesh, nmesh, xoff, xoff2 < 0 ? nx_local : 2 * nx_local, pitch_r(nmesh), Lbox, offset,
                                paste == 0, norm, paste, sub, xoff2 < 0 ? -1 : xoff2, nx_alloc);
}

int abacus_slab_deposit_dev(float *pos, int64_t n, const float *w, float *mesh, int nmesh, int xoff, int xoff2, int nx_local,
                            double Lbox, double offset, double norm, int paste, double sub) {
    return abacus_slab_deposit_padded_dev(pos, n, w, mesh, nmesh, xoff, xoff2, nx_local, Lbox, offset, norm, paste, sub, 0);
}

int abacus_slab_axpy_dev(float *dst, const float *src, int64_t nfloat, float add) {
    ABACUS_ENTER();
    if (nfloat % 4) return fail("abacus_slab_axpy_dev: length must be a multiple of 4");
    const int grid = (int)std::min<int64_t>(std::max<int64_t>(ceil_div(nfloat / 4, 256), 1), 256 * 32);
    ABACUS_LAUNCH("slab_axpy", slab_axpy, dim3(grid), dim3(256), 0, dst, src, nfloat / 4, add);
    return 0;
}

// the slab path runs the fused form of the transform (fft.hip) wherever the single-GPU path does; `slab_nofuse` (A/B,
// comparator of the tests) keeps the plain three-pass form.  zy, unpack, x and the binning decide alike.
static bool slab_fused(int nmesh) { return use_fused_fft(nmesh) && !option("slab_nofuse"); }

int abacus_slab_fused(int nmesh) { return slab_fused(nmesh) ? 1 : 0; }

static int slab_fold(const char *who, int nmesh, int world, int *h) {
    if (world < 1 || nmesh % (2 * world)) return fail("%s: nmesh %d must be divisible by twice the %d ranks", who, nmesh, world);
    *h = nmesh / (2 * world);
    return 0;
}

static int slab_pack_launch(const void *data, void *send, int nmesh, int world, int h, int64_t xsep, int p0, int pc) {
    const int nyl = nmesh / world, pitch4 = pitch_r(nmesh) / 4;
    const int grid = (int)std::min<int64_t>((int64_t)world * 2 * pc * nyl, 256 * 32);
    ABACUS_LAUNCH("slab_pack", slab_pack, dim3(grid), dim3(256), 0, (const float4 *)data, (float4 *)send, nmesh, h, xsep, nyl,
                  world, pitch4, p0, pc);
    return 0;
}

// z and y passes of the plane pairs [p0, p0 + pc) of a rank's folded slab: `mesh` is the first plane of the rank's first
// half (global plane xg0), the second half (global plane xg0 + nmesh/2) starts xsep planes behind it, h = nmesh / (2 world)
// pairs in all.  send == NULL: in place.  send != NULL: the result goes to the send buffer of the pencil transpose,
// send[peer][s h + p][y_local][k] - written by the y pass itself where the fused form runs, by a pack pass otherwise.
// what the binning's last edge leaves of a row: |k|^2 in fundamental units with the margin of power_dev's xcut
static float slab_cut(double Lbox, double k_last) {
    const double e = k_last / (2.0 * M_PI / Lbox);
    return (float)(e * e * (1.0 + 1e-5) + 1.0);
}
static bool slab_compact_ok(int nmesh, int world) {
    const int nyl = nmesh / std::max(world, 1);
    return slab_fused(nmesh) && (nmesh == 1024 || nmesh == 2048) && world > 1 && world <= 16 && !(nyl & (nyl - 1)) && nyl % 16 == 0 &&
           !option("slab_nopackfuse") && !option("slab_nounpackfuse") && !option("slab_nocompact") && !option("pk_noxbin");
}

// COMPACT pencil transpose: complex elements per plane of the block that goes to every peer (P_out[world]) when the columns beyond
// the binning's last edge k_last stay at home (fft.hip, slab_layout); 1 = not available for this mesh / rank count (regular layout)
int abacus_slab_transpose_layout(int nmesh, int world, double Lbox, double k_last, int64_t *P_out) {
    ABACUS_ENTER();
    if (!P_out || !slab_compact_ok(nmesh, world)) return 1;
    return slab_layout_query(nmesh, world, slab_cut(Lbox, k_last), pitch_r(nmesh) / 2, P_out, nullptr);
}

static int slab_fft_zy(float *mesh, void *send, int nmesh, int world, int64_t xsep, int xg0, int p0, int pc, float cut) {
    if (!fft_native_pow2(nmesh)) return fail("slab path: nmesh must be a power of two in [64, 2048]");
    int h;
    ABACUS_TRY(slab_fold("abacus_slab_fft_zy_dev", nmesh, world, &h));
    if (p0 < 0 || pc < 1 || p0 + pc > h || xsep < h) return fail("abacus_slab_fft_zy_dev: pairs [%d, +%d) of %d, halves %lld planes apart", p0, pc, h, (long long)xsep);
    const int pr = pitch_r(nmesh), nyl = nmesh / world;
    const size_t plane = (size_t)nmesh * pr;
    if (cut > 0.f && (!send || !slab_compact_ok(nmesh, world))) return fail("abacus_slab_fft_zy_compact_dev: no compact layout for this run");
    if (slab_fused(nmesh)) {
        const bool ypack = send && !option("slab_nopackfuse") && world > 1 && !(nyl & (nyl - 1));
        ABACUS_TRY(fft_native_fused_zy_slab(mesh, nmesh, pr, h, xsep, xg0, p0, pc, ypack ? (float *)send : nullptr, world, cut));
        if (ypack || !send) return 0;
    } else {
        ABACUS_TRY(fft_native_zy(mesh + (size_t)p0 * plane, nmesh, pr, pc));
        ABACUS_TRY(fft_native_zy(mesh + ((size_t)xsep + p0) * plane, nmesh, pr, pc));
        if (!send) return 0;
    }
    return slab_pack_launch(mesh, send, nmesh, world, h, xsep, p0, pc);
}
int abacus_slab_fft_zy_dev(float *mesh, void *send, int nmesh, int world, int64_t xsep, int xg0, int p0, int pc) {
    ABACUS_ENTER();
    return slab_fft_zy(mesh, send, nmesh, world, xsep, xg0, p0, pc, 0.f);
}
// the same, writing the COMPACT send buffer of abacus_slab_transpose_layout(nmesh, world, Lbox, k_last)
int abacus_slab_fft_zy_compact_dev(float *mesh, void *send, int nmesh, int world, int64_t xsep, int xg0, int p0, int pc, double Lbox,
                                   double k_last) {
    ABACUS_ENTER();
    return slab_fft_zy(mesh, send, nmesh, world, xsep, xg0, p0, pc, slab_cut(Lbox, k_last));
}

int abacus_slab_pack_dev(const void *data, void *send, int nmesh, int world, int64_t xsep, int p0, int pc) {
    ABACUS_ENTER();
    int h;
    ABACUS_TRY(slab_fold("abacus_slab_pack_dev", nmesh, world, &h));
    if (p0 < 0 || pc < 1 || p0 + pc > h || xsep < h) return fail("abacus_slab_pack_dev: pairs [%d, +%d) of %d", p0, pc, h);
    return slab_pack_launch(data, send, nmesh, world, h, xsep, p0, pc);
}

int abacus_slab_unpack_dev(const void *recv, void *out, int nmesh, int world) {
    ABACUS_ENTER();
    int h;
    ABACUS_TRY(slab_fold("abacus_slab_unpack_dev", nmesh, world, &h));
    const int nyl = nmesh / world, pitch4 = pitch_r(nmesh) / 4;
    const int grid = (int)std::min<int64_t>((int64_t)world * 2 * h * nyl, 256 * 32);
    ABACUS_LAUNCH("slab_unpack", slab_unpack, dim3(grid), dim3(256), 0, (const float4 *)recv, (float4 *)out, nmesh, h, nyl,
                  world, pitch4);
    return 0;
}

int abacus_slab_fft_x_dev(float *data, int nmesh, int ny_local) {
    ABACUS_ENTER();
    if (!fft_native_pow2(nmesh)) return fail("slab path: nmesh must be a power of two in [64, 2048]");
    if (slab_fused(nmesh)) return fft_native_fused_x_slab(data, nmesh, pitch_r(nmesh), ny_local);
    const int pc = pitch_r(nmesh) / 2;
    return fft_native_x(data, nmesh, pitch_r(nmesh), ny_local, pc, (int64_t)nmesh * pc);   // layout (y_local, x, k)
}

int abacus_slab_bin_dev(const void *a, const void *as, const void *b, const void *bs, int nmesh, int y0, int ny_local,
                        double Lbox, const float *W_host, int interlaced, const double *kedges, int Nk,
                        const double *muedges, int Nmu, const int64_t *poles, int Np, void *raw_out) {
    ABACUS_ENTER();
    ABACUS_TRY(ensure_phase(nmesh));
    const float *W_dev;
    ABACUS_TRY(upload_W(W_host, nmesh, &W_dev));
    SpecArgs s;
    fill_spec(s, nmesh, 1, interlaced, W_dev, b != nullptr);
    s.rowmode = 1;
    s.y0 = y0;
    s.nrows = (int64_t)ny_local * nmesh;
    if (slab_fused(nmesh))
        while ((2 << s.permshift) < nmesh) s.permshift++;   // rows of x and y in the fused form's order
    s.a = (const float2 *)a, s.as = (const float2 *)as, s.b = (const float2 *)b, s.bs = (const float2 *)bs;
    return run_bin(s, Lbox, kedges, Nk, muedges, Nmu, poles, Np, nullptr, nullptr, nullptr, nullptr, nullptr, raw_out);
}

// Last pass fused with the binning on a y-slab BEFORE its x pass (auto power of one non-interlaced field).  from_transpose
// = 1: `mesh` is the receive buffer of the pencil transpose of a `world`-rank run as it arrived, (peer, 2 h, y_local, k)
// - with one rank the slab itself, in place -; 0: the unpacked (y_local, x, k) block.  Returns 0 with the raw sums in
// raw_out, 1 if this mesh / histogram is not served by the fused last pass (the caller then runs abacus_slab_unpack_dev,
// abacus_slab_fft_x_dev, abacus_slab_bin_dev), < 0 on error.  put_geom: exactly one rank passes 1 (the mesh-wide N_mode and
// sum |k| come from the cached geometry, not from the y-slab).
// from_transpose = 2: the COMPACT receive buffer (abacus_slab_transpose_layout with the binning's last edge); mesh == NULL: a
// query - 0 if the fused last pass will serve this mesh / histogram from the transpose, 1 if not (nothing is computed, the
// geometry descriptor is built and cached)
int abacus_slab_xbin_dev(const void *mesh, int nmesh, int world, int y0, int ny_local, double Lbox, const float *W_host,
                         const double *kedges, int Nk, const double *muedges, int Nmu, const int64_t *poles, int Np,
                         int put_geom, int from_transpose, void *raw_out) {
    return abacus_slab_xbin_pair_dev(mesh, nullptr, 0, nmesh, world, y0, ny_local, Lbox, W_host, kedges, Nk, muedges, Nmu, poles, Np, put_geom,
                                     from_transpose, raw_out);
}

// the same over TWO fields in the same layout.  pair_mode 2: their cross power Re(conj(a) b), calc_power(pos, pos2=...,
// interlaced=False) over slabs - LRG x ELG of BASELINE config 5; pair_mode 1: mesh2 is the half-cell-shifted deposit of the
// same particles, the auto power of the interlaced combination (calc_power's default mode, power_spectrum.py:951-998);
// pair_mode 0 (mesh2 ignored): abacus_slab_xbin_dev.  The query form (mesh == NULL) answers for the pair_mode it is given
int abacus_slab_xbin_pair_dev(const void *mesh, const void *mesh2, int pair_mode, int nmesh, int world, int y0, int ny_local, double Lbox,
                              const float *W_host, const double *kedges, int Nk, const double *muedges, int Nmu, const int64_t *poles,
                              int Np, int put_geom, int from_transpose, void *raw_out) {
    if (pair_mode < 0 || pair_mode > 2) return fail("abacus_slab_xbin_pair_dev: pair_mode %d", pair_mode);
    return abacus_slab_xbin_quad_dev(mesh, mesh2, nullptr, nullptr, pair_mode, nmesh, world, y0, ny_local, Lbox, W_host, kedges, Nk, muedges, Nmu,
                                     poles, Np, put_geom, from_transpose, raw_out);
}

// ... and over FOUR: pair_mode 3, the cross power of two INTERLACED fields (calc_power's defaults with pos2 over slabs): mesh /
// mesh2 = the first catalogue's unshifted / shifted deposit, mesh3 / mesh4 the second catalogue's, all in one layout
// (fft_x_bin2<.., QUAD>).  pair_mode 0 - 2: abacus_slab_xbin_pair_dev (mesh3 / mesh4 ignored)
int abacus_slab_xbin_quad_dev(const void *mesh, const void *mesh2, const void *mesh3, const void *mesh4, int pair_mode, int nmesh, int world,
                              int y0, int ny_local, double Lbox, const float *W_host, const double *kedges, int Nk, const double *muedges,
                              int Nmu, const int64_t *poles, int Np, int put_geom, int from_transpose, void *raw_out) {
    ABACUS_ENTER();
    if (pair_mode < 0 || pair_mode > 3) return fail("abacus_slab_xbin_quad_dev: pair_mode %d", pair_mode);
    if (mesh && pair_mode && !mesh2) return fail("abacus_slab_xbin_pair_dev: pair_mode %d without a second field", pair_mode);
    if (mesh && pair_mode == 3 && (!mesh3 || !mesh4)) return fail("abacus_slab_xbin_quad_dev: pair_mode 3 needs four fields");
    if ((pair_mode == 1 && option("pk_noxbin_inter")) || (pair_mode == 2 && option("pk_noxbin_cross")) ||
        (pair_mode == 3 && (option("pk_noxbin_inter") || option("pk_noxbin_cross"))))
        return 1;
    if (pair_mode == 1 || pair_mode == 3) ABACUS_TRY(ensure_phase(nmesh));
    if (!pair_mode) mesh2 = nullptr;
    if (pair_mode != 3) mesh3 = mesh4 = nullptr;
    if (!slab_fused(nmesh) || option("pk_noxbin") || (nmesh != 1024 && nmesh != 2048)) return 1;
    if (from_transpose && option("slab_nounpackfuse")) return 1;
    int h;
    ABACUS_TRY(slab_fold("abacus_slab_xbin_dev", nmesh, world, &h));
    if (from_transpose && (h & (h - 1))) return 1;
    const float *W_dev;
    ABACUS_TRY(upload_W(W_host, nmesh, &W_dev));
    BinArgs b;
    size_t acc_bytes = 0;
    ABACUS_TRY(prepare_bins(Lbox, kedges, Nk, muedges, Nmu, poles, Np, 0, b, acc_bytes));
    if (!xbin2_supported(nmesh, b, W_dev != nullptr)) return 1;
    if (!mesh) return 0;
    const unsigned int *row_off = nullptr;
    int64_t P[16] = {0};
    if (from_transpose == 2) {
        if (!slab_compact_ok(nmesh, world) || Nk < 1) return fail("abacus_slab_xbin_dev: no compact layout for this run");
        ABACUS_TRY(slab_layout_query(nmesh, world, slab_cut(Lbox, kedges[Nk]), pitch_r(nmesh) / 2, P, &row_off));
    }
    const double M = (double)nmesh * nmesh * nmesh;
    ABACUS_TRY(fft_x_bin_run((const float *)mesh, nmesh, pitch_r(nmesh), (float)(1.0 / M), W_dev, b, b.dbg, y0, ny_local,
                             put_geom ? 1 : 0, from_transpose ? 2 : 1, world, (const float *)mesh2,
                             (pair_mode == 1 || pair_mode == 3) ? g_ctx.phase.as<float2>() : nullptr, row_off, P[y0 / std::max(ny_local, 1)],
                             (const float *)mesh3, (const float *)mesh4));
    HIP_TRY(hipMemcpyAsync(raw_out, g_ctx.accum.p, acc_bytes, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int64_t abacus_bin_raw_bytes(int Nk, int Nmu, const int64_t *poles, int Np) {
    int nnz = 0;
    for (int q = 0; q < Np; q++) nnz += poles[q] != 0;
    return (int64_t)Nk * Nmu * 24 + (int64_t)nnz * Nk * 8;
}

int abacus_bin_finalize(const void *raw, double Lbox, int Nk, int Nmu, const int64_t *poles, int Np, float *power,
                        int64_t *N_mode, float *binned_poles, int64_t *N_mode_poles, float *k_avg) {
    return finalize_bins(raw, Lbox, Nk, Nmu, poles, Np, power, N_mode, binned_poles, N_mode_poles, k_avg);
}

// ---- ZCV-facing helpers --------------------------------------------------------------------------------------
namespace {
int helper_grid(int64_t total) { return (int)std::min<int64_t>(std::max<int64_t>(ceil_div(total, 256), 1), 256 * 32); }

// bin a real (n, n, zdim) device grid like bin_kmu does a caller-supplied weights array
int bin_real_grid(const float *d_w, int n, int zdim, float prescale, double Lbox, double dk, double scale,
                  const double *kedges, int Nk, const double *muedges, int Nmu, const int64_t *poles, int Np, float *power,
                  int64_t *N_mode, float *binned_poles, int64_t *N_mode_poles, float *k_avg) {
    const int kzlen = n / 2 + 1;
    const int64_t total = (int64_t)n * n * kzlen;
    ABACUS_TRY(g_ctx.mesh[1].reserve((size_t)total * 8));
    ABACUS_LAUNCH("helper_real_to_c64", helper_real_to_c64, dim3(helper_grid(total)), dim3(256), 0, d_w,
                  g_ctx.mesh[1].as<float2>(), n, zdim, kzlen, prescale);
    SpecArgs s;
    fill_spec(s, n, 0, 0, nullptr, false);
    s.linear = 1;
    s.a = g_ctx.mesh[1].as<float2>();
    s.as = s.b = s.bs = nullptr;
    return run_bin(s, Lbox, kedges, Nk, muedges, Nmu, poles, Np, power, N_mode, binned_poles, N_mode_poles, k_avg, nullptr,
                   dk, scale);
}
}  // namespace

int abacus_bin_weights(const float *weights, int n1d, int zdim, double Lbox, int fourier, const double *kedges, int Nk,
                       const double *muedges, int Nmu, const int64_t *poles, int Np, double scale, float *power,
                       int64_t *N_mode, float *binned_poles, int64_t *N_mode_poles, float *k_avg) {
    ABACUS_ENTER();
    if (!weights || n1d < 2 || n1d > 32767 || zdim < n1d / 2 + 1) return fail("abacus_bin_weights: bad grid shape");
    const size_t bytes = (size_t)n1d * n1d * zdim * 4;
    ABACUS_TRY(g_ctx.helper_in.reserve(bytes));
    HIP_TRY(hipMemcpyAsync(g_ctx.helper_in.p, weights, bytes, hipMemcpyHostToDevice, stream()));
    const double dk = fourier ? 2.0 * M_PI / Lbox : Lbox / n1d;   // (:210-213)
    return bin_real_grid(g_ctx.helper_in.as<float>(), n1d, zdim, 1.0f, Lbox, dk, scale > 0 ? scale : 1.0, kedges, Nk, muedges,
                         Nmu, poles, Np, power, N_mode, binned_poles, N_mode_poles, k_avg);
}

int abacus_pk_to_xi(const float *Pk, int n, double Lbox, const double *redges, int Nr, const int64_t *poles, int Np,
                    float *binned_poles, int64_t *N_poles) {
    ABACUS_ENTER();
    if (!Pk || n < 2 || n > 4096) return fail("abacus_pk_to_xi: bad grid size");
    const int kzlen = n / 2 + 1;
    // scipy's irfftn without `s` returns 2 (kzlen - 1) points along z: n for even n, n - 1 for odd n (:645); the
    // reference then bins that (n, n, nz) grid with nmesh = n
    const int nz = 2 * (kzlen - 1);
    const int64_t nc = (int64_t)n * n * kzlen, nr = (int64_t)n * n * nz;
    ABACUS_TRY(g_ctx.helper_in.reserve((size_t)nc * 4));
    ABACUS_TRY(g_ctx.mesh[0].reserve((size_t)nc * 8));
    ABACUS_TRY(g_ctx.mesh[2].reserve((size_t)nr * 4));
    HIP_TRY(hipMemcpyAsync(g_ctx.helper_in.p, Pk, (size_t)nc * 4, hipMemcpyHostToDevice, stream()));
    ABACUS_LAUNCH("helper_real_to_c64", helper_real_to_c64, dim3(helper_grid(nc)), dim3(256), 0, g_ctx.helper_in.as<float>(),
                  g_ctx.mesh[0].as<float2>(), n, kzlen, kzlen, 1.0f);
    auto it = g_ctx.c2r_plans.find(n);
    if (it == g_ctx.c2r_plans.end()) {
        hipfftHandle h;
        ABACUS_TRY(fft_check(plan_with_retry([&] { return hipfftPlan3d(&h, n, n, nz, HIPFFT_C2R); }), "hipfftPlan3d(C2R)"));
        it = g_ctx.c2r_plans.emplace(n, h).first;
    }
    ABACUS_TRY(fft_check(hipfftSetStream(it->second, stream()), "hipfftSetStream"));
    prof_begin("hipfft_c2r");
    const hipfftResult r = hipfftExecC2R(it->second, (hipfftComplex *)g_ctx.mesh[0].p, (hipfftReal *)g_ctx.mesh[2].p);
    prof_end("hipfft_c2r");
    ABACUS_TRY(fft_check(r, "hipfftExecC2R"));
    // Xi = irfftn(Pk) (scipy normalises by the number of output points); multipoles in r bins, times nmesh^3 (:655-659)
    std::vector<float> power(Nr), k_avg(Nr);
    std::vector<int64_t> N_mode(Nr);
    const double mu01[2] = {0.0, 1.0};
    return bin_real_grid(g_ctx.mesh[2].as<float>(), n, nz, (float)(1.0 / ((double)n * n * nz)), Lbox, Lbox / n,
                         (double)n * n * n, redges, Nr, mu01, 1, poles, Np, power.data(), N_mode.data(), binned_poles, N_poles,
                         k_avg.data());
}

int abacus_bin_kppi(const float *weights, int n1d, int zdim, double Lbox, const double *kedges, int Nk, double pimax, int Npi,
                    int fourier, float *mean, int64_t *counts) {
    ABACUS_ENTER();
    if (!weights || n1d < 2 || zdim < n1d / 2 + 1 || Nk < 1 || Npi < 1) return fail("abacus_bin_kppi: bad arguments");
    const double dk = fourier ? 2.0 * M_PI / Lbox : Lbox / n1d;
    std::vector<float> e((size_t)Nk + 1 + Npi + 1);
    for (int q = 0; q <= Nk; q++) e[q] = (float)((kedges[q] / dk) * (kedges[q] / dk));
    for (int q = 0; q <= Npi; q++) {   // np.linspace(0, pimax, Npi + 1) / dk, squared, float32 (:370)
        const double pe = (q == Npi ? pimax : pimax * ((double)q / Npi)) / dk;
        e[Nk + 1 + q] = (float)(pe * pe);
    }
    const size_t bytes = (size_t)n1d * n1d * zdim * 4, nb = (size_t)Nk * Npi;
    ABACUS_TRY(g_ctx.helper_in.reserve(bytes));
    ABACUS_TRY(g_ctx.helper_tab.reserve(e.size() * 4));
    ABACUS_TRY(g_ctx.accum.reserve(nb * 16));
    HIP_TRY(hipMemcpyAsync(g_ctx.helper_in.p, weights, bytes, hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipMemcpyAsync(g_ctx.helper_tab.p, e.data(), e.size() * 4, hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipMemsetAsync(g_ctx.accum.p, 0, nb * 16, stream()));
    unsigned long long *d_cnt = g_ctx.accum.as<unsigned long long>();
    double *d_sum = reinterpret_cast<double *>(d_cnt + nb);
    ABACUS_LAUNCH("helper_bin_kppi", helper_bin_kppi, dim3(n1d), dim3(256), 0, g_ctx.helper_in.as<float>(), n1d, zdim,
                  g_ctx.helper_tab.as<float>(), Nk, g_ctx.helper_tab.as<float>() + Nk + 1, Npi, d_cnt, d_sum);
    std::vector<unsigned long long> h_cnt(nb);
    std::vector<double> h_sum(nb);
    HIP_TRY(hipMemcpyAsync(h_cnt.data(), d_cnt, nb * 8, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipMemcpyAsync(h_sum.data(), d_sum, nb * 8, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    for (size_t q = 0; q < nb; q++) {
        counts[q] = (int64_t)h_cnt[q];
        mean[q] = (float)(h_cnt[q] ? h_sum[q] / (double)h_cnt[q] : h_sum[q]);
    }
    return 0;
}

int abacus_get_smoothing(int n1d, double Lbox, double R, float *out) {
    ABACUS_ENTER();
    if (!out || n1d < 2) return fail("abacus_get_smoothing: bad arguments");
    const int64_t total = (int64_t)n1d * n1d * (n1d / 2 + 1);
    ABACUS_TRY(g_ctx.helper_in.reserve((size_t)total * 4));
    const float dk = (float)(2.0 * M_PI / Lbox);
    ABACUS_LAUNCH("helper_smoothing", helper_smoothing, dim3(helper_grid(total)), dim3(256), 0, g_ctx.helper_in.as<float>(),
                  n1d, dk * dk, (float)(R * R));
    HIP_TRY(hipMemcpyAsync(out, g_ctx.helper_in.p, (size_t)total * 4, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_get_delta_mu2(const void *delta_c64, int n1d, void *out_c64) {
    ABACUS_ENTER();
    if (!delta_c64 || !out_c64 || n1d < 2) return fail("abacus_get_delta_mu2: bad arguments");
    const int64_t total = (int64_t)n1d * n1d * (n1d / 2 + 1);
    ABACUS_TRY(g_ctx.mesh[0].reserve((size_t)total * 8));
    ABACUS_TRY(g_ctx.mesh[1].reserve((size_t)total * 8));
    HIP_TRY(hipMemcpyAsync(g_ctx.mesh[0].p, delta_c64, (size_t)total * 8, hipMemcpyHostToDevice, stream()));
    ABACUS_LAUNCH("helper_delta_mu2", helper_delta_mu2, dim3(helper_grid(total)), dim3(256), 0, g_ctx.mesh[0].as<float2>(),
                  g_ctx.mesh[1].as<float2>(), n1d);
    HIP_TRY(hipMemcpyAsync(out_c64, g_ctx.mesh[1].p, (size_t)total * 8, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_raw_power(const void *field_c64, const void *field2_c64, int64_t total, float *out) {
    ABACUS_ENTER();
    if (!field_c64 || !out || total < 0) return fail("abacus_raw_power: bad arguments");
    if (total == 0) return 0;
    ABACUS_TRY(g_ctx.mesh[0].reserve((size_t)total * 8));
    if (field2_c64) ABACUS_TRY(g_ctx.mesh[1].reserve((size_t)total * 8));
    ABACUS_TRY(g_ctx.helper_in.reserve((size_t)total * 4));
    HIP_TRY(hipMemcpyAsync(g_ctx.mesh[0].p, field_c64, (size_t)total * 8, hipMemcpyHostToDevice, stream()));
    if (field2_c64) HIP_TRY(hipMemcpyAsync(g_ctx.mesh[1].p, field2_c64, (size_t)total * 8, hipMemcpyHostToDevice, stream()));
    ABACUS_LAUNCH("helper_raw_power", helper_raw_power, dim3(helper_grid(total)), dim3(256), 0, (const float2 *)g_ctx.mesh[0].as<float2>(),
                  (const float2 *)(field2_c64 ? g_ctx.mesh[1].as<float2>() : nullptr), g_ctx.helper_in.as<float>(), total);
    HIP_TRY(hipMemcpyAsync(out, g_ctx.helper_in.p, (size_t)total * 4, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_shift_field_fft(void *field_c64, const void *shift_c64, int n1d, double Lbox, double d) {
    ABACUS_ENTER();
    if (!field_c64 || !shift_c64 || n1d < 2 || !(Lbox > 0)) return fail("abacus_shift_field_fft: bad arguments");
    const int64_t total = (int64_t)n1d * n1d * (n1d / 2 + 1);
    ABACUS_TRY(g_ctx.mesh[0].reserve((size_t)total * 8));
    ABACUS_TRY(g_ctx.mesh[1].reserve((size_t)total * 8));
    HIP_TRY(hipMemcpyAsync(g_ctx.mesh[0].p, field_c64, (size_t)total * 8, hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipMemcpyAsync(g_ctx.mesh[1].p, shift_c64, (size_t)total * 8, hipMemcpyHostToDevice, stream()));
    const float dk = (float)(2.0 * M_PI / Lbox), halfd = (float)(0.5 * (double)(float)d), norm = (float)(0.5 / ((double)n1d * n1d * n1d));
    ABACUS_LAUNCH("helper_shift_field", helper_shift_field, dim3(helper_grid(total)), dim3(256), 0, g_ctx.mesh[0].as<float2>(),
                  (const float2 *)g_ctx.mesh[1].as<float2>(), n1d, dk, halfd, norm);
    HIP_TRY(hipMemcpyAsync(field_c64, g_ctx.mesh[0].p, (size_t)total * 8, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_expand_poles_to_3d(const double *k_ell, const double *P_ell, int nk, int n1d, double Lbox, const int64_t *poles,
                              int Np, float *out) {
    ABACUS_ENTER();
    if (!k_ell || !P_ell || !poles || !out || nk < 2 || n1d < 2 || Np < 1 || Np > MAX_POLES)
        return fail("abacus_expand_poles_to_3d: bad arguments");
    if (std::fabs((k_ell[1] - k_ell[0]) - (k_ell[nk - 1] - k_ell[nk - 2])) >= 1.0e-6)
        return fail("abacus_expand_poles_to_3d: k_ell must be equidistant (:475)");
    BinArgs b;
    memset(&b, 0, sizeof b);
    std::vector<int> slot(Np, -1);
    for (int q = 0; q < Np; q++)
        if (poles[q] != 0) {
            ABACUS_TRY(pole_coefs((int)poles[q], b.polecoef[b.Np]));
            b.poledeg[b.Np] = (int)poles[q] / 2;
            slot[q] = b.Np++;
        }
    std::vector<float> tab((size_t)nk * (Np + 1));
    for (int m = 0; m < nk; m++) tab[m] = (float)k_ell[m];
    for (int q = 0; q < Np; q++)
        for (int m = 0; m < nk; m++) tab[(size_t)(q + 1) * nk + m] = (float)P_ell[(size_t)q * nk + m];
    const int64_t total = (int64_t)n1d * n1d * (n1d / 2 + 1);
    ABACUS_TRY(g_ctx.helper_in.reserve((size_t)total * 4));
    ABACUS_TRY(g_ctx.helper_tab.reserve(tab.size() * 4 + (size_t)Np * 4 + 64));
    float *d_tab = g_ctx.helper_tab.as<float>();
    int *d_slot = reinterpret_cast<int *>(d_tab + tab.size());
    HIP_TRY(hipMemcpyAsync(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipMemcpyAsync(d_slot, slot.data(), (size_t)Np * 4, hipMemcpyHostToDevice, stream()));
    ABACUS_LAUNCH("helper_expand_poles", helper_expand_poles, dim3(helper_grid(total)), dim3(256), 0,
                  g_ctx.helper_in.as<float>(), n1d, (float)(2.0 * M_PI / Lbox), (const float *)d_tab, (const float *)(d_tab + nk),
                  nk, b, Np, (const int *)d_slot);
    HIP_TRY(hipMemcpyAsync(out, g_ctx.helper_in.p, (size_t)total * 4, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

double abacus_power_geometry_ms(void) { return xbin_last_build_ms(); }
double abacus_power_last_batches(void) { return g_last_batches; }
int abacus_power_xbin_generation(void) { return xbin_last_gen(); }

}  // extern "C"

namespace abacus {
int power_trim_caches() {
    HIP_TRY(hipStreamSynchronize(stream()));
    for (auto &m : g_ctx.mesh) ABACUS_TRY(m.release());
    ABACUS_TRY(fft_trim_scratch());
    return 0;
}
}  // namespace abacus

extern "C" {

int abacus_power_release(void) {
    ABACUS_TRY(xbin_release());
    for (auto &kv : g_ctx.plans) (void)hipfftDestroy(kv.second);
    g_ctx.plans.clear();
    for (auto &kv : g_ctx.c2r_plans) (void)hipfftDestroy(kv.second);
    g_ctx.c2r_plans.clear();
    for (auto &kv : g_ctx.plans64) (void)hipfftDestroy(kv.second);
    g_ctx.plans64.clear();
    ABACUS_TRY(g_ctx.helper_in.release());
    ABACUS_TRY(g_ctx.helper_tab.release());
    for (auto &m : g_ctx.mesh) ABACUS_TRY(m.release());
    for (int q = 0; q < PowerCtx::NFIELD; q++) {
        ABACUS_TRY(g_ctx.field[q][0].release());
        ABACUS_TRY(g_ctx.field[q][1].release());
        g_ctx.finfo[q] = PowerCtx::FieldInfo();
    }
    for (DevBuf *b : {&g_ctx.W, &g_ctx.phase, &g_ctx.edges, &g_ctx.accum, &g_ctx.pos, &g_ctx.pos2, &g_ctx.w, &g_ctx.w2})
        ABACUS_TRY(b->release());
    g_ctx.phase_n = 0;
    if (g_copy_stream) {                  // the batched host upload's stream and events (field_fft_dev)
        (void)hipStreamSynchronize(g_copy_stream);
        for (hipEvent_t &e : g_copy_ev) {
            if (e) (void)hipEventDestroy(e);
            e = nullptr;
        }
        (void)hipStreamDestroy(g_copy_stream);
        g_copy_stream = nullptr;
    }
    ABACUS_TRY(fft_native_release());
    ABACUS_TRY(tsc_release_work());
    return scratch_trim_idle();
}

}  // extern "C"

// ---- float64 meshes: get_field / get_field_fft / calc_power with dtype=np.float64 (analysis/power_spectrum.py:808-857, 1001-1070,
// 1131-1319) --------------------------------------------------------------------------------------------------------------------
// What dtype changes in the reference: the mesh, its normalisation, the transform and 1/M scaling are float64 (complex128
// spectrum), the compensation divides by the float32 window products; an INTERLACED field ignores it (get_interlaced_field_fft is
// called without dtype, :1048-1052: float32 meshes, complex64 spectrum) and bin_kmu is called with its default float32
// arithmetic whatever the mesh (:787-789), its float64 weights entering float32 accumulators.  Here: float64 deposit
// (tsc.hip), the mixed-radix double-precision transform of gfft.hip, raw power in float64, then the usual binning (float64
// accumulators over the weights rounded to float32).
namespace {
int pitch_r64(int n) { return (n + 2 + 15) / 16 * 16; }          // doubles per z row: rows start on 128-B lines

// spectrum *= inv_size; /= (W[i] W[j]) W[k] (float32 products like the reference's broadcast, :1063-1069)
__global__ void f64_scale_compensate(double2 *__restrict__ f, int n, int pitch_c, double inv_size, const float *__restrict__ W) {
    const int kzlen = n / 2 + 1;
    const int64_t total = (int64_t)n * n * kzlen;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(q % kzlen);
        const int64_t row = q / kzlen;
        double2 v = f[row * pitch_c + k];
        v.x *= inv_size, v.y *= inv_size;
        if (W) {
            const float w = (W[row / n] * W[row % n]) * W[k];
            v.x /= (double)w, v.y /= (double)w;
        }
        f[row * pitch_c + k] = v;
    }
}
// get_raw_power (:707-727) of padded float64 spectra into contiguous float32 weights (n, n, kzlen)
__global__ void f64_raw_power(const double2 *__restrict__ a, const double2 *__restrict__ b, int n, int pitch_c, float *__restrict__ out) {
    const int kzlen = n / 2 + 1;
    const int64_t total = (int64_t)n * n * kzlen;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(q % kzlen);
        const int64_t row = q / kzlen;
        const double2 u = a[row * pitch_c + k];
        double p;
        if (b) {
            const double2 v = b[row * pitch_c + k];
            p = u.x * v.x + u.y * v.y;
        } else {
            const double m = hypot(u.x, u.y);
            p = m * m;
        }
        out[q] = (float)p;
    }
}

// deposit + normalise + transform of one particle set into g_ctx.mesh[slot] (float64, padded in-place R2C layout)
int field64_dev(void *pos, int pos_f64, int64_t n, const void *w, double L, int nmesh, int paste, double offset, int slot, bool transform,
                const float *W_dev) {
    if (n <= 0) return fail("power: empty particle set");
    const int pr = pitch_r64(nmesh);
    ABACUS_TRY(g_ctx.mesh[slot].reserve((size_t)nmesh * nmesh * pr * sizeof(double)));
    double *mesh = g_ctx.mesh[slot].as<double>();
    const double M = (double)nmesh * nmesh * nmesh;
    ABACUS_TRY(tsc_deposit_f64mesh(pos, pos_f64, n, w, mesh, nmesh, pr, L, offset, paste == 0, M / (double)n, paste, 1.0));
    if (!transform) return 0;
    if (gfft_supported(nmesh, 1)) {
        ABACUS_TRY(gfft_r2c_inplace_f64(mesh, nmesh, pr));
    } else {   // odd sizes, prime factors above 13: hipFFT's double-precision transform on the same padded in-place layout
        auto it = g_ctx.plans64.find(nmesh);
        if (it == g_ctx.plans64.end()) {
            hipfftHandle h;
            int dims[3] = {nmesh, nmesh, nmesh};
            int inembed[3] = {nmesh, nmesh, pr}, onembed[3] = {nmesh, nmesh, pr / 2};
            ABACUS_TRY(fft_check(plan_with_retry([&] { return hipfftPlanMany(&h, 3, dims, inembed, 1, 1, onembed, 1, 1, HIPFFT_D2Z, 1); }), "hipfftPlanMany (D2Z)"));
            it = g_ctx.plans64.emplace(nmesh, h).first;
        }
        ABACUS_TRY(fft_check(hipfftSetStream(it->second, stream()), "hipfftSetStream"));
        prof_begin("hipfft_d2z");
        const hipfftResult r = hipfftExecD2Z(it->second, (hipfftDoubleReal *)mesh, (hipfftDoubleComplex *)mesh);
        prof_end("hipfft_d2z");
        ABACUS_TRY(fft_check(r, "hipfftExecD2Z"));
    }
    const int64_t total = (int64_t)nmesh * nmesh * (nmesh / 2 + 1);
    ABACUS_LAUNCH("f64_scale_compensate", f64_scale_compensate, dim3(helper_grid(total)), dim3(256), 0, reinterpret_cast<double2 *>(mesh), nmesh,
                  pr / 2, 1.0 / M, W_dev);
    return 0;
}

// host particles (float32 or float64) -> device copies in g_ctx.pos / g_ctx.w (slot 0) or pos2 / w2
int stage_particles_any(const void *pos, int pos_f64, int64_t n, const void *w, DevBuf &dp, DevBuf &dw, void **pd, void **wd) {
    const size_t es = pos_f64 ? 8 : 4;
    ABACUS_TRY(dp.reserve((size_t)std::max<int64_t>(n, 1) * 3 * es));
    HIP_TRY(hipMemcpyAsync(dp.p, pos, (size_t)n * 3 * es, hipMemcpyHostToDevice, stream()));
    *pd = dp.p, *wd = nullptr;
    if (w) {
        ABACUS_TRY(dw.reserve((size_t)std::max<int64_t>(n, 1) * es));
        HIP_TRY(hipMemcpyAsync(dw.p, w, (size_t)n * es, hipMemcpyHostToDevice, stream()));
        *wd = dw.p;
    }
    return 0;
}
}  // namespace

extern "C" {

int abacus_field_f64(void *pos, int pos_f64, int64_t n, const void *w, double Lbox, int nmesh, int paste, double offset, double *field) {
    ABACUS_ENTER();
    ABACUS_TRY(check_common(nmesh, paste));
    if (!field || !pos) return fail("abacus_field_f64: null argument");
    void *pd, *wd;
    ABACUS_TRY(stage_particles_any(pos, pos_f64, n, w, g_ctx.pos, g_ctx.w, &pd, &wd));
    tsc_wrapped_reset();
    ABACUS_TRY(field64_dev(pd, pos_f64, n, wd, Lbox, nmesh, paste, offset, 0, false, nullptr));
    const int pr = pitch_r64(nmesh);
    HIP_TRY(hipMemcpy2DAsync(field, (size_t)nmesh * 8, g_ctx.mesh[0].p, (size_t)pr * 8, (size_t)nmesh * 8, (size_t)nmesh * nmesh,
                             hipMemcpyDeviceToHost, stream()));
    if (paste == 0 && tsc_wrapped_seen()) HIP_TRY(hipMemcpyAsync(pos, pd, (size_t)n * 3 * (pos_f64 ? 8 : 4), hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_field_fft_f64(void *pos, int pos_f64, int64_t n, const void *w, double Lbox, int nmesh, int paste, const float *W_host,
                         void *out_c128) {
    ABACUS_ENTER();
    ABACUS_TRY(check_common(nmesh, paste));
    if (!out_c128 || !pos) return fail("abacus_field_fft_f64: null argument");
    const float *W_dev;
    ABACUS_TRY(upload_W(W_host, nmesh, &W_dev));
    void *pd, *wd;
    ABACUS_TRY(stage_particles_any(pos, pos_f64, n, w, g_ctx.pos, g_ctx.w, &pd, &wd));
    tsc_wrapped_reset();
    ABACUS_TRY(field64_dev(pd, pos_f64, n, wd, Lbox, nmesh, paste, 0.0, 0, true, W_dev));
    const int pr = pitch_r64(nmesh), kzlen = nmesh / 2 + 1;
    HIP_TRY(hipMemcpy2DAsync(out_c128, (size_t)kzlen * 16, g_ctx.mesh[0].p, (size_t)pr * 8, (size_t)kzlen * 16, (size_t)nmesh * nmesh,
                             hipMemcpyDeviceToHost, stream()));
    if (paste == 0 && tsc_wrapped_seen()) HIP_TRY(hipMemcpyAsync(pos, pd, (size_t)n * 3 * (pos_f64 ? 8 : 4), hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_power_f64(void *pos, int pos_f64, int64_t n, const void *w, void *pos2, int64_t n2, const void *w2, double Lbox, int nmesh,
                     int paste, const float *W_host, const double *kedges, int Nk, const double *muedges, int Nmu, const int64_t *poles,
                     int Np, float *power, int64_t *N_mode, float *binned_poles, int64_t *N_mode_poles, float *k_avg) {
    ABACUS_ENTER();
    ABACUS_TRY(check_common(nmesh, paste));
    if (!pos) return fail("abacus_power_f64: null positions");
    const float *W_dev;
    ABACUS_TRY(upload_W(W_host, nmesh, &W_dev));
    void *pd, *wd, *pd2 = nullptr, *wd2 = nullptr;
    ABACUS_TRY(stage_particles_any(pos, pos_f64, n, w, g_ctx.pos, g_ctx.w, &pd, &wd));
    tsc_wrapped_reset();
    ABACUS_TRY(field64_dev(pd, pos_f64, n, wd, Lbox, nmesh, paste, 0.0, 0, true, W_dev));
    const bool wrapped1 = paste == 0 && tsc_wrapped_seen();
    bool wrapped2 = false;
    if (pos2) {
        ABACUS_TRY(stage_particles_any(pos2, pos_f64, n2, w2, g_ctx.pos2, g_ctx.w2, &pd2, &wd2));
        tsc_wrapped_reset();
        ABACUS_TRY(field64_dev(pd2, pos_f64, n2, wd2, Lbox, nmesh, paste, 0.0, 1, true, W_dev));
        wrapped2 = paste == 0 && tsc_wrapped_seen();
    }
    const int pr = pitch_r64(nmesh), kzlen = nmesh / 2 + 1;
    const int64_t total = (int64_t)nmesh * nmesh * kzlen;
    ABACUS_TRY(g_ctx.helper_in.reserve((size_t)total * 4));
    ABACUS_LAUNCH("f64_raw_power", f64_raw_power, dim3(helper_grid(total)), dim3(256), 0, (const double2 *)g_ctx.mesh[0].as<double2>(),
                  (const double2 *)(pos2 ? g_ctx.mesh[1].as<double2>() : nullptr), nmesh, pr / 2, g_ctx.helper_in.as<float>());
    if (wrapped1) HIP_TRY(hipMemcpyAsync(pos, pd, (size_t)n * 3 * (pos_f64 ? 8 : 4), hipMemcpyDeviceToHost, stream()));
    if (wrapped2) HIP_TRY(hipMemcpyAsync(pos2, pd2, (size_t)n2 * 3 * (pos_f64 ? 8 : 4), hipMemcpyDeviceToHost, stream()));
    // bin_kmu on the raw power (its default float32 edge arithmetic, :787-789), times L^3 (:792-795)
    return bin_real_grid(g_ctx.helper_in.as<float>(), nmesh, kzlen, 1.0f, Lbox, 2.0 * M_PI / Lbox, Lbox * Lbox * Lbox, kedges, Nk, muedges, Nmu,
                         poles, Np, power, N_mode, binned_poles, N_mode_poles, k_avg);
}

}  // extern "C"

