// Last FFT pass fused with the (k, mu) / multipole binning: the x pass of the fused 3-D R2C
// transform (fft.hip: two independent n/2-point column transforms per axis after the radix-2 stages done in the z pass)
// no longer writes the spectrum back - a workgroup transforms a tile of H = n/2 x-rows by C kz-columns in LDS and bins
// its |delta_k|^2 straight from LDS into the workgroup's float64 LDS histogram.  Replaces, for the auto power spectrum of
// one non-interlaced field, fft_cols<H, 16> (x) + spectrum_bin: one 4M-byte read instead of 4M read + 4M write + 4M read
// (M = mesh cells), and the per-mode bin arithmetic is shared by the two modes +kx / -kx, which sit in the same LDS
// column (same |k|, same mu: abacusnbody/analysis/power_spectrum.py:233-244 depends on i only through i2).
//
// Reference chain restated: _normalize (:1073-1078, times f32(1/M)), compensation divide (:1063-1069),
// get_raw_power |delta_k|^2 (:707-727), bin_kmu (:150-300): edges squared in units of dk as float32, skip below
// edges[0], stop at edges[-1], bin b holds edges[b] < k^2 <= edges[b+1] (bin 0 closed below), mu^2 = f32(k^2) / kmag2,
// weight 1 on the kz = 0 plane and 2 elsewhere, k_avg from sqrt(kmag2), multipoles (2l+1) P_l(mu) summed per k bin.
// The reference's running bin search along kz is stateless in exact terms (|k| and mu are monotone along a row), so a
// traversal along kx with its own running search (|k| rises, mu falls with |kx|) lands every mode in the same bin.
// Sums are float64 (the reference's float32 per-thread accumulators are not reproducible across thread counts).
//
// Tile = (x-half xh, y-row yr, column tile ct): rows r = 0..H-1 hold x = xh*H + r ... after the transform position f holds
// x-frequency i = 2f + xh; the row yr holds y-frequency j = 2 (yr mod H) + (yr div H) (fft.hip's permuted order).
// One wave owns a column through the butterfly passes AND its binning (no workgroup barrier between them); lane l
// walks 8 (H = 1024) consecutive values of |i|, pairing position f with its mirror.
#include <cmath>
#include <cstdlib>

#include "common.hpp"
#include "bin_device.hpp"

using namespace abacus;

namespace abacus {
const float2 *fft_twiddles(int n);   // exp(-2 pi i m / n), m < n, device table (fft.hip)
int fft_num_cus();
}  // namespace abacus

namespace {

#include "fft_device.hpp"

constexpr int XB_THREADS = 512;

struct XBinGeom {
    int n, kzlen, pitch_c;      // mesh size, n/2 + 1, complex row pitch
    float inv_size;             // f32(1/M)
    const float *W;             // (n,) compensation window or nullptr
    int dbg;                    // ablation: 1 skip the transform, 2 skip the binning, 4 no histogram atomics, 8 no LDS reads in the binning
};

template <int H, int C, int NP, bool COMP>
__global__ __launch_bounds__(XB_THREADS) void fft_x_bin(const float2 *__restrict__ data, XBinGeom g, BinArgs b,
                                                         const float2 *__restrict__ twH) {
    constexpr int CP = colpitch_of<H>();
    constexpr int NLD = (H * (C / 2)) / XB_THREADS;       // 16-B loads per thread and tile
    static_assert((H * (C / 2)) % XB_THREADS == 0 && wave_local(H), "tile shape");
    constexpr int NPC = NP > 0 ? NP : 1;
    constexpr int RUN = (H / 2) / 64;                     // values of |i| per lane
    extern __shared__ __align__(16) unsigned char smem[];
    const int nb = b.Nk * b.Nmu;
    // LDS: [twiddles H][tile C x CP][sum f64 nb][ksum f64 nb][mu^2 and mu^4 moments f64 2*Nk][cnt u32 nb][kedges2][muedges2][W n]
    float2 *tw = reinterpret_cast<float2 *>(smem);
    float2 *lds = tw + H;
    double *h_sum = reinterpret_cast<double *>(lds + C * CP + 1);   // CP odd, C even: +1 keeps 16-B alignment irrelevant, 8-B holds
    double *h_ksum = h_sum + nb;
    double *h_m2 = h_ksum + nb, *h_m4 = h_m2 + b.Nk;   // per k bin: sum w P mu^2, sum w P mu^4 (NP > 0)
    unsigned int *h_cnt = reinterpret_cast<unsigned int *>(h_m4 + b.Nk);
    float *ke = reinterpret_cast<float *>(h_cnt + nb);
    float *me = ke + (b.Nk + 1);
    float *Wl = me + (b.Nmu + 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int q = tid; q < H; q += XB_THREADS) tw[q] = twH[q];
    for (int q = tid; q < nb; q += XB_THREADS) {
        h_sum[q] = 0.0;
        h_ksum[q] = 0.0;
        h_cnt[q] = 0u;
    }
    for (int q = tid; q < 2 * b.Nk; q += XB_THREADS) h_m2[q] = 0.0;
    for (int q = tid; q <= b.Nk; q += XB_THREADS) ke[q] = b.kedges2[q];
    for (int q = tid; q <= b.Nmu; q += XB_THREADS) me[q] = b.muedges2[q];
    if (COMP)
        for (int q = tid; q < g.n; q += XB_THREADS) Wl[q] = g.W[q];
    float pc[NPC][3];
#pragma unroll
    for (int q = 0; q < NPC; q++)
#pragma unroll
        for (int m = 0; m < 3; m++) pc[q][m] = (q < NP && m <= b.poledeg[q]) ? b.polecoef[q][m] : 0.f;
    const float klo = b.kedges2[0], khi = b.kedges2[b.Nk];
    const int n = g.n, Nmu = b.Nmu, Nk = b.Nk;
    const int ntile_c = (g.kzlen + C - 1) / C;
    const int64_t S = (int64_t)n * g.pitch_c;           // x stride (complex elements)
    const int n_outer = 2 * n;                          // (xh, yr)

    v4f regs[NLD];
    auto tile_ptr = [&](int o, int ct) {                // o = xh * n + yr
        const int xh = o >= n ? 1 : 0, yr = o - xh * n;
        return data + (int64_t)xh * H * S + (int64_t)yr * g.pitch_c + ct * C;
    };
    auto prefetch = [&](const float2 *p) {
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int e = q * XB_THREADS + tid;
            const int c2 = (e % (C / 2)) * 2, y = e / (C / 2);
            gload16_async(regs[q], p + (int64_t)y * S + c2);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < NLD; q++) touch(regs[q]);
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int e = q * XB_THREADS + tid;
            const int c2 = (e % (C / 2)) * 2, y = e / (C / 2);
            lds[c2 * CP + padq(y)] = make_float2(regs[q].x, regs[q].y);
            lds[(c2 + 1) * CP + padq(y)] = make_float2(regs[q].z, regs[q].w);
        }
    };
    // tile order as in fft_cols: the workgroups of one XCD (b % 8) walk the column tiles of the same outer indices
    const bool xmap = (gridDim.x % 8 == 0) && (n_outer % 8 == 0);
    const int grp = xmap ? (int)(blockIdx.x & 7) : 0, ostep = xmap ? 8 : 1;
    const unsigned int qstep = xmap ? (gridDim.x >> 3) : gridDim.x, q0 = xmap ? (blockIdx.x >> 3) : blockIdx.x;
    const int n_og = n_outer / ostep;
    const int dg = (int)(qstep / (unsigned int)ntile_c), dc = (int)(qstep % (unsigned int)ntile_c);
    int og = (int)(q0 / (unsigned int)ntile_c), ct = (int)(q0 % (unsigned int)ntile_c);
    __syncthreads();                                     // tables and the zeroed histogram
    if (og < n_og) {
        prefetch(tile_ptr(og * ostep + grp, ct));
        wait_vmcnt<0>();
        stage();
        for (;;) {
            __syncthreads();
            const int o_cur = og * ostep + grp, ct_cur = ct;
            og += dg, ct += dc;
            if (ct >= ntile_c) ct -= ntile_c, og++;
            const bool has_next = og < n_og;
            if (has_next) prefetch(tile_ptr(og * ostep + grp, ct));
            // ---- this wave's columns: transform, then bin ----
            const int xh = o_cur >= n ? 1 : 0, yr = o_cur - xh * n;
            const int j = ((yr & (H - 1)) << 1) | (yr >= H ? 1 : 0);
            const int jj = j < n / 2 ? j : j - n;
#pragma unroll 1
            for (int c = wave; c < C; c += XB_THREADS / 64) {
                float2 *col = lds + c * CP;
                if (!(g.dbg & 1)) {
                    PassesW<H, H>::run(col, tw, lane);
                    wave_sync();
                }
                const int k = ct_cur * C + c;
                if (k >= g.kzlen || (g.dbg & 2)) continue;
                const int r2 = jj * jj + k * k;
                const float k2f = (float)(k * k);
                const float wk = k == 0 ? 1.f : 2.f;
                const int ck = k == 0 ? 1 : 2;
                // lane l: a in [RUN*l, RUN*l + RUN) (+ a = H/2 for the last lane of the even half); |i| = 2a + xh
                const int a0 = lane * RUN;
                const int a1 = a0 + RUN + ((lane == 63 && xh == 0) ? 1 : 0);
                int cur = -1, cur_bk = 0, cnt = 0;
                // the multipoles are accumulated as MOMENTS - sum w P mu^2 and sum w P mu^4 per k bin, in the run and in the
                // LDS histogram alike; the Legendre combination (2l+1) P_l(mu) = c0 + c1 mu^2 + c2 mu^4 is linear in them and
                // is formed once per workgroup at the end (the flush of a run is 3 + 2 atomics and no float64 arithmetic:
                // its issue slots are paid by the whole wave whenever one lane changes bin)
                float sp = 0.f, sk = 0.f, s2 = 0.f, s4 = 0.f;
                auto flush = [&]() {
                    if (cnt && !(g.dbg & 4)) {   // 4: ablation, no histogram atomics
                        atomicAdd(&h_cnt[cur], (unsigned int)cnt);
                        atomicAdd(&h_sum[cur], (double)sp);
                        atomicAdd(&h_ksum[cur], (double)sk);
                        if (NP > 0) {
                            atomicAdd(&h_m2[cur_bk], (double)s2);
                            atomicAdd(&h_m4[cur_bk], (double)s4);
                        }
                    }
                    cnt = 0;
                    sp = sk = s2 = s4 = 0.f;
                };
                int a = a0;
                int iabs = 2 * a + xh;
                // skip below the first edge (:246): kmag2 rises with a
                while (a < a1 && (float)(r2 + iabs * iabs) < klo) a++, iabs += 2;
                if (a < a1 && (float)(r2 + iabs * iabs) < khi) {
                    const float kmag2a = (float)(r2 + iabs * iabs);
                    int bk = lower_bin(ke, Nk - 1, kmag2a);
                    float ke_hi = ke[bk + 1];
                    const float mu2a = kmag2a > 0.f ? k2f * (1.0f / kmag2a) : 0.f;
                    int bmu = lower_bin(me, Nmu - 1, mu2a);
                    float me_lo = me[bmu], me_hi = me[bmu + 1];
#pragma unroll 1
                    for (; a < a1; a++, iabs += 2) {
                        const float kmag2 = (float)(r2 + iabs * iabs);   // dtype(i2 + j2 + k**2) (:239), exact integer
                        if (kmag2 >= khi) break;                         // (:249): everything further out too
                        while (kmag2 > ke_hi) {                          // (:252-253)
                            bk++;
                            ke_hi = ke[bk + 1];
                        }
                        float mu2 = kmag2 > 0.f ? k2f * __builtin_amdgcn_rcpf(kmag2) : 0.f;
                        while (bmu > 0 && !(mu2 > me_lo)) {              // mu falls with |i|: the bin search runs backwards
                            bmu--;
                            me_hi = me_lo;
                            me_lo = me[bmu];
                        }
                        const float tol = 6e-7f * mu2;
                        if (fabsf(mu2 - me_hi) <= tol || fabsf(mu2 - me_lo) <= tol) {
                            mu2 = kmag2 > 0.f ? k2f * (1.0f / kmag2) : 0.f;   // IEEE division, as the reference rounds it (:240-241)
                            bmu = lower_bin(me, Nmu - 1, mu2);
                            me_lo = me[bmu];
                            me_hi = me[bmu + 1];
                        }
                        const int tb = bk * Nmu + bmu;
                        if (tb != cur) {
                            flush();
                            cur = tb;
                            cur_bk = bk;
                        }
                        // the two modes with this |i|: positions fA = a and its mirror fB (one mode for i = 0 and i = n/2)
                        const int fA = a;
                        const int fB = xh ? H - 1 - a : (H - a) & (H - 1);
                        float2 vA = (g.dbg & 8) ? make_float2(1.f, (float)a) : col[padq(fA)];   // 8: ablation, no LDS reads
                        vA.x *= g.inv_size, vA.y *= g.inv_size;                // _normalize (:1058-1060)
                        if (COMP) {
                            const float scl = 1.0f / ((Wl[2 * fA + xh] * Wl[j]) * Wl[k]);   // (:1065-1069)
                            vA.x *= scl, vA.y *= scl;
                        }
                        float p = vA.x * vA.x + vA.y * vA.y;                   // get_raw_power (:726)
                        int mult = 1;
                        if (fB != fA) {
                            float2 vB = (g.dbg & 8) ? make_float2(1.f, (float)a) : col[padq(fB)];
                            vB.x *= g.inv_size, vB.y *= g.inv_size;
                            if (COMP) {
                                const float scl = 1.0f / ((Wl[2 * fB + xh] * Wl[j]) * Wl[k]);
                                vB.x *= scl, vB.y *= scl;
                            }
                            p += vB.x * vB.x + vB.y * vB.y;
                            mult = 2;
                        }
                        cnt += ck * mult;
                        const float wp = wk * p;
                        sp += wp;
                        sk += (wk * (float)mult) * __builtin_amdgcn_sqrtf(kmag2);
                        if (NP > 0) {
                            const float m2 = wp * mu2;
                            s2 += m2;
                            s4 += m2 * mu2;
                        }
                    }
                }
                flush();
            }
            if (!has_next) break;
            __syncthreads();     // every wave is done with the tile
            wait_vmcnt<0>();     // no stores in this kernel: the prefetch is all that is outstanding
            stage();
        }
    }
    __syncthreads();
    for (int q = tid; q < nb; q += XB_THREADS)
        if (h_cnt[q]) {
            atomicAdd(&b.g_cnt[q], (unsigned long long)h_cnt[q]);
            atomicAdd(&b.g_sum[q], h_sum[q]);
            atomicAdd(&b.g_ksum[q], h_ksum[q]);
        }
    if (NP > 0)
        for (int bk = tid; bk < Nk; bk += XB_THREADS) {
            double s0 = 0.0;   // sum w P of the k bin over its mu bins
            for (int m = 0; m < Nmu; m++) s0 += h_sum[bk * Nmu + m];
            const double m2 = h_m2[bk], m4 = h_m4[bk];
#pragma unroll
            for (int q = 0; q < NPC; q++)
                if (q < NP) {
                    const double v = (double)pc[q][0] * s0 + (double)pc[q][1] * m2 + (double)pc[q][2] * m4;
                    if (v != 0.0) atomicAdd(&b.g_pole[q * Nk + bk], v);
                }
        }
}

template <int H, int C>
size_t xbin_lds_bytes(int n, int Nk, int Nmu, int Np, bool comp) {
    const size_t nb = (size_t)Nk * Nmu;
    return (size_t)(H + C * colpitch_of<H>() + 1) * sizeof(float2) + nb * (8 + 8 + 4) + (size_t)2 * Nk * 8 +
           (size_t)(Nk + 1 + Nmu + 1) * 4 + (comp ? (size_t)n * 4 : 0) + 16;
}

template <int H, int C, int NP, bool COMP>
int launch_xbin(const float2 *data, const XBinGeom &g, const BinArgs &b, size_t lds) {
    auto kern = fft_x_bin<H, C, NP, COMP>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 1;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, XB_THREADS, lds));
    const int64_t ntiles = (int64_t)2 * g.n * ((g.kzlen + C - 1) / C);
    const unsigned int grid = (unsigned int)std::min<int64_t>(ntiles, (int64_t)fft_num_cus() * std::max(per_cu, 1));
    const float2 *tw = fft_twiddles(H);
    if (!tw) return -1;
    ABACUS_LAUNCH("fft_x_bin", kern, dim3(grid), dim3(XB_THREADS), lds, data, g, b, tw);
    return 0;
}

template <int H, int C>
int dispatch_xbin(const float2 *data, const XBinGeom &g, const BinArgs &b, size_t lds) {
#define XB(NP)                                                         \
    (g.W ? launch_xbin<H, C, NP, true>(data, g, b, lds) : launch_xbin<H, C, NP, false>(data, g, b, lds))
    switch (b.Np) {
        case 0: return XB(0);
        case 1: return XB(1);
        case 2: return XB(2);
    }
#undef XB
    return fail("fft_x_bin: %d multipoles", b.Np);
}

}  // namespace

namespace abacus {

// can the fused last pass serve this mesh / histogram?  (n/2-point wave-local transforms: n = 1024, 2048; at most two
// ell != 0 multipoles of degree <= 4; tile + histogram within the 160 KiB LDS)
bool xbin_supported(int n, int Nk, int Nmu, const BinArgs &b, bool comp) {
    if (b.Np > 2) return false;
    for (int q = 0; q < b.Np; q++)
        if (b.poledeg[q] > 2) return false;
    const size_t cap = 160 * 1024;
    if (n == 2048) return xbin_lds_bytes<1024, 8>(n, Nk, Nmu, b.Np, comp) <= cap;
    if (n == 1024) return xbin_lds_bytes<512, 16>(n, Nk, Nmu, b.Np, comp) <= cap;
    return false;
}

// `mesh` holds the fused transform after its z and y passes (fft_native_r2c_fused_zy); bins |delta_k|^2 of every mode
// into the accumulators of `b` (zeroed by the caller)
int fft_x_bin_run(const float *mesh, int n, int pitch_r, float inv_size, const float *W_dev, const BinArgs &b, int dbg) {
    XBinGeom g;
    g.n = n, g.kzlen = n / 2 + 1, g.pitch_c = pitch_r / 2, g.inv_size = inv_size, g.W = W_dev, g.dbg = dbg;
    const float2 *data = reinterpret_cast<const float2 *>(mesh);
    if (n == 2048) {
        if (((g.kzlen + 7) / 8) * 8 > g.pitch_c) return fail("fft_x_bin: row pitch too small");
        return dispatch_xbin<1024, 8>(data, g, b, xbin_lds_bytes<1024, 8>(n, b.Nk, b.Nmu, b.Np, W_dev != nullptr));
    }
    if (n == 1024) {
        if (((g.kzlen + 15) / 16) * 16 > g.pitch_c) return fail("fft_x_bin: row pitch too small");
        return dispatch_xbin<512, 16>(data, g, b, xbin_lds_bytes<512, 16>(n, b.Nk, b.Nmu, b.Np, W_dev != nullptr));
    }
    return fail("fft_x_bin: unsupported mesh %d", n);
}

}  // namespace abacus
