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
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

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
    // fft_x_bin2 only - where the rows live.  Row i (0 <= i < n/2) of half s of x (s = 0: the sums of the first radix-2
    // stage = even frequencies, s = 1: the twiddled differences = odd frequencies) of y-row y0 <= y < y0 + ny sits at
    //   data[(i >> lgh) * ps + (s * h + (i & (h - 1))) * xs + (y - y0) * ys + k],   h = 1 << lgh
    // full mesh (x, y, k): h = n/2, xs = n * pitch_c, ys = pitch_c, ny = n; y-slab (y_local, x, k): h = n/2, xs = pitch_c,
    // ys = n * pitch_c; receive buffer of the pencil transpose of a W-rank run, (peer, 2 h, y_local, k) with h = n / (2 W):
    // xs = ny * pitch_c, ys = pitch_c, ps = 2 h * xs - the folded slabs of slab_power.py, every peer's block holding its
    // h sum rows, then its h difference rows
    int64_t xs, ys, ps;
    int lgh;
    int ny, y0;
    int put_geom;               // copy the cached N_mode / sum |k| into the accumulators (one rank of a slab run does)
    // interlaced (fft_x_bin2<.., INTER>): the half-cell-shifted field's mesh in the same layout, the table of the exact
    // phases exp(i pi m / n), m < 2n, and f32(0.5 / M) (analysis/power_spectrum.py:932-947, 996-998)
    const float2 *data2 = nullptr;
    const float2 *phase = nullptr;      // nullptr with data2: the CROSS form (data2 = the second field, no shift)
    float half_inv_size = 0.f;
    // QUAD (interlaced CROSS power, four fields): data / data2 = the first catalogue's unshifted and shifted meshes, data3 / data4
    // the second catalogue's, all in one layout
    const float2 *data3 = nullptr, *data4 = nullptr;
    // compact pencil transpose (fft.hip, slab_layout): row yr of the y-slab starts row_off[y0 + yr] elements into a plane block
    // (xs = elements per plane of the blocks this rank receives) instead of yr * ys
    const unsigned int *row_off = nullptr;
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


// =====================================================================================================================
// Second generation of the binning half: everything about a mode's bin that does not depend on the
// data is taken out of the hot loop.  For a mesh n and a set of float32 edges (power_spectrum.py:217-218) the bin of a mode
// is a function of two INTEGERS - kmag2 = i^2 + j^2 + k^2 (exact in float32 below 2^24) and k:
//   * k bin: the reference's float tests `kmag2 < edges[0]`, `kmag2 >= edges[-1]`, `kmag2 > edges[b+1]` (:246-253) are
//     tests of the integer against integer thresholds T[e] (floor / ceil of the edges): the extended bin
//     eb = #{e in 0..Nk : kmag2 > T[e]} is 0 below the first edge, b + 1 in bin b, Nk + 1 beyond the last edge;
//   * mu bin: mu2 = f32(k^2) * (1 / f32(kmag2)) is non-increasing in kmag2 for a fixed k, so `mu2 > muedges2[m]` (:255)
//     holds for kmag2 <= U[k][m], one integer per (k, inner edge), found on the host by bisection with the same two
//     float32 roundings.
// eb comes from ONE LDS word per mode: the bit pattern of f32(kmag2), shifted, indexes 2^m cells per octave; a cell holds
// the eb of its smallest integer and the threshold that ends that bin, `eb = word >> 22 + (kmag2 > (word & 0x3fffff))`,
// exact when no cell spans two edges - checked per cell on the host, m = 6..10, and then for EVERY mode of the mesh
// against the float tests by xbin_geometry on the device.  That kernel also yields N_mode and sum |k| of every bin, which
// depend on (n, edges) alone: the descriptor is cached per (n, edges) and the hot kernel accumulates only w * P and its
// two mu moments - no counts, no k sums, no square root, no walk along the edges.
#include "xdesc_device.hpp"

// N_mode, sum |k| and the validation of the descriptor: one wave per (j, k) column, lanes over |i| (both signs at once)
__global__ __launch_bounds__(256) void xbin_geometry(int n, int Nk, int Nmu, const float *__restrict__ ke,
                                                     const float *__restrict__ me, XDesc d,
                                                     unsigned long long *__restrict__ cnt, double *__restrict__ ksum,
                                                     int *__restrict__ flag) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int nb = Nk * Nmu;
    double *hk = reinterpret_cast<double *>(smem);
    unsigned int *hc = reinterpret_cast<unsigned int *>(hk + nb);
    for (int q = threadIdx.x; q < nb; q += 256) hk[q] = 0.0, hc[q] = 0u;
    __syncthreads();
    const int kzlen = n / 2 + 1, lane = threadIdx.x & 63;
    const int64_t ncol = (int64_t)n * kzlen;
    const float klo = ke[0], khi = ke[Nk];
    int bad = 0;
    for (int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); col < ncol; col += (int64_t)gridDim.x * 4) {
        const int j = (int)(col / kzlen), k = (int)(col - (int64_t)j * kzlen);
        const int jj = j < n / 2 ? j : j - n;
        const int r2 = jj * jj + k * k;
        const float k2f = (float)(k * k);
        const int ck = k == 0 ? 1 : 2;
        for (int i = lane; i <= n / 2; i += 64) {
            const int v = r2 + i * i;
            const float vf = (float)v;
            int tb = -1;
            if (!(vf < klo) && !(vf >= khi)) {                       // (:246-250)
                const int bk = lower_bin(ke, Nk - 1, vf);
                const float mu2 = v > 0 ? k2f * (1.0f / vf) : 0.f;   // (:240-244)
                tb = bk * Nmu + lower_bin(me, Nmu - 1, mu2);
            }
            const int eb = xd_eb(d.lut - d.off, d.sh, v, fmaxf(vf, 1.f));
            int bm = 0;
            for (int m = 0; m < Nmu - 1; m++) bm += v <= d.U[k * d.ustride + m] ? 1 : 0;
            const int fast = (unsigned int)(eb - 1) < (unsigned int)Nk ? (eb - 1) * Nmu + bm : -1;
            bad |= fast != tb;
            if (tb >= 0) {
                const int mult = (i == 0 || i == n / 2) ? 1 : 2;
                atomicAdd(&hc[tb], (unsigned int)(ck * mult));
                atomicAdd(&hk[tb], (double)((float)(ck * mult) * sqrtf(vf)));
            }
        }
    }
    if (bad) atomicOr(flag, 1);
    __syncthreads();
    for (int q = threadIdx.x; q < nb; q += 256)
        if (hc[q]) {
            atomicAdd(&cnt[q], (unsigned long long)hc[q]);
            atomicAdd(&ksum[q], hk[q]);
        }
}

// The hot kernel: the transform half is fft_x_bin's; the binning half reads the 16 values of a lane's 8 pairs, then per pair
// ~25 vector instructions of |delta|^2, cell look-up and threshold compares.  RUNS: a lane keeps the sums of its current
// bin in registers and adds them to the LDS histogram when the bin changes (fine bins: every step or two; coarse bins:
// once per column); !RUNS: one LDS atomic per pair and accumulator.  LDS histogram rows eb = 0 and Nk + 1 take what
// lies outside the edges and are never read.
// The rows of a tile may come straight from the receive buffer of a multi-GPU pencil transpose (XBinGeom: every peer's block
// holds a run of h rows of either half): no unpack pass, no second copy of the slab.
// INTER: the interlaced pair of fields in one pass.  A tile of the unshifted field is transformed first and every wave keeps
// the values of its column(s) in registers; the same tile of the shifted field follows through the same LDS, and the bin
// takes |(a + a' exp(i pi m / n)) f32(0.5 / M)|^2, m = i + j + k - what spectrum_bin<INTER> computes from two spectra in
// HBM, without the two x passes writing them (2 x 8M bytes) and the binning reading them back (8M).
// CROSS (with INTER): the second tile is ANOTHER field's (pos2 of calc_power, not interlaced): the bin takes
// Re(conj(a) b) f32(1 / M)^2 - get_raw_power's cross form (power_spectrum.py:722-726) - instead of the interlaced combination.
// QUAD (with INTER): the cross power of two INTERLACED fields - calc_power(pos, pos2=..., interlaced=True), the reference's
// defaults with a second catalogue (power_spectrum.py:1200-1260 -> get_interlaced_field_fft twice, then get_raw_power's cross
// form): four tiles follow each other through the same LDS.  a: kept; a': K = a + a' e^{i pi m / n} kept in its place; b: the
// lane keeps ONE float per pair of modes, P = w Re(conj(K) b) (the product is linear in the second field, so the pair b, b' never
// has to be held); b': P += w Re(conj(K) b' e^{i pi m / n}), binned with f32(0.5 / M)^2.  Four x-pass write-backs and
// spectrum_bin's four reads less than the three-pass form.
template <int H, int C, int NP, bool COMP, int MU, bool RUNS, bool INTER = false, bool CROSS = false, bool QUAD = false>
__global__ __launch_bounds__(XB_THREADS) void fft_x_bin2(const float2 *__restrict__ data, XBinGeom g, BinArgs b, XDesc d,
                                                          const float2 *__restrict__ twH) {
    constexpr int CP = colpitch_of<H>();
    constexpr int NLD = (H * (C / 2)) / XB_THREADS;
    static_assert((H * (C / 2)) % XB_THREADS == 0 && wave_local(H), "tile shape");
    constexpr int RUN = (H / 2) / 64;                     // values of |i| per lane (8, 4): a run shares its pad term
    static_assert(RUN == 8 || RUN == 4, "run length");
    constexpr int NPC = NP > 0 ? NP : 1;
    extern __shared__ __align__(16) unsigned char smem[];
    const int Nk = b.Nk, Nmu = b.Nmu;
    const int nrow = Nk + 2, nbx = nrow * Nmu;
    // LDS: [tile C x CP][sum f64 (Nk+2)*Nmu][mu^2, mu^4 moments f64 2*(Nk+2)][cell words ncell][U kzlen*(MU-1)][W n]
    float2 *lds = reinterpret_cast<float2 *>(smem);
    double *h_sum = reinterpret_cast<double *>(lds + C * CP + 1);
    double *h_m2 = h_sum + nbx, *h_m4 = h_m2 + nrow;
    unsigned int *lut = reinterpret_cast<unsigned int *>(h_m4 + nrow);
    int *Ul = reinterpret_cast<int *>(lut + d.ncell);                 // (kzlen, MU - 1)
    float *Wl = reinterpret_cast<float *>(Ul + g.kzlen * (MU - 1));
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // twiddles as lane constants: first pass (fused with the staging) j = tid / (C/2); second pass j = lane % (H/64)
    float2 tw1[8], tw2[8];
#pragma unroll
    for (int r = 1; r < 8; r++) {
        tw1[r] = INTER ? make_float2(0.f, 0.f) : twH[(tid / (C / 2)) * r];
        tw2[r] = (INTER && H == 1024) ? make_float2(0.f, 0.f) : twH[(lane % (H / 64)) * r * 8];
    }
    tw1[0] = tw2[0] = make_float2(1.f, 0.f);
    for (int q = tid; q < nbx; q += XB_THREADS) h_sum[q] = 0.0;
    for (int q = tid; q < 2 * nrow; q += XB_THREADS) h_m2[q] = 0.0;
    for (int q = tid; q < d.ncell; q += XB_THREADS) lut[q] = d.lut[q];
    for (int q = tid; q < g.kzlen * (MU - 1); q += XB_THREADS) Ul[q] = d.U[(q / (MU > 1 ? MU - 1 : 1)) * XD_USTRIDE + q % (MU > 1 ? MU - 1 : 1)];
    if (COMP)
        for (int q = tid; q < g.n; q += XB_THREADS) Wl[q] = g.W[q];
    const int n = g.n;
    const int ntile_c = (g.kzlen + C - 1) / C;
    const int64_t S = g.xs;
    const int n_outer = 2 * g.ny;
    static_assert(!CROSS || INTER, "the cross form runs the two-tile schedule");
    static_assert(!QUAD || (INTER && !CROSS), "the four-field form is the interlaced schedule run twice");
    const float inv2 = (INTER && !CROSS) ? g.half_inv_size * g.half_inv_size : g.inv_size * g.inv_size;
    const int sh = d.sh;
    const unsigned int *lut0 = lut - d.off;

    v4f regs[NLD];
    const int lgh = g.lgh, hmask = (1 << lgh) - 1;
    auto tile_ptr = [&](int o, int ct, const float2 *base) {
        const int xh = o >= g.ny ? 1 : 0, yr = o - xh * g.ny;
        return base + ((int64_t)xh << lgh) * S + (g.row_off ? (int64_t)g.row_off[g.y0 + yr] : (int64_t)yr * g.ys) + ct * C;
    };
    auto prefetch = [&](const float2 *p) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int e = q * XB_THREADS + tid;
            const int c2 = (e % (C / 2)) * 2, y = e / (C / 2);
            if (!(g.dbg & 32)) gload16_async(regs[q], p + (int64_t)(y >> lgh) * g.ps + (int64_t)(y & hmask) * S + c2);   // 32: ablation, no loads
        }
    };
    // registers -> LDS THROUGH the first radix-8 pass: load q of a thread is row tid / (C/2) + q * H/8 of its column pair,
    // i.e. the eight loads are the inputs r = 0..7 of butterfly j = tid / (C/2) of the first DIF pass (sub-length H):
    // the staged tile is never written raw and read back (one LDS round trip of the tile less)
    static_assert((H * (C / 2)) / XB_THREADS == 8 && H / 8 == XB_THREADS / (C / 2), "the loads of a thread form one radix-8 butterfly per column");
    auto stage = [&]() {
        const int c2 = (tid % (C / 2)) * 2, jb = tid / (C / 2);
        float2 u[8], w[8];
#pragma unroll
        for (int q = 0; q < NLD; q++) touch(regs[q]);
#pragma unroll
        for (int q = 0; q < 8; q++) u[q] = make_float2(regs[q].x, regs[q].y), w[q] = make_float2(regs[q].z, regs[q].w);
        dft<8>(u);
        dft<8>(w);
#pragma unroll
        for (int r = 1; r < 8; r++) {
            // INTER: the kept columns of the first field need the registers - the staging twiddles come from the (cached) table
            const float2 t = INTER ? twH[(tid / (C / 2)) * r] : tw1[r];
            u[r] = cmul(u[r], t);
            w[r] = cmul(w[r], t);
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            lds[c2 * CP + padq_strided<H / 8>(jb, r)] = u[r];
            lds[(c2 + 1) * CP + padq_strided<H / 8>(jb, r)] = w[r];
        }
    };
    const bool xmap = (gridDim.x % 8 == 0) && (n_outer % 8 == 0);
    const int grp = xmap ? (int)(blockIdx.x & 7) : 0, ostep = xmap ? 8 : 1;
    const unsigned int qstep = xmap ? (gridDim.x >> 3) : gridDim.x, q0 = xmap ? (blockIdx.x >> 3) : blockIdx.x;
    const int n_og = n_outer / ostep;
    const int dg = (int)(qstep / (unsigned int)ntile_c), dc = (int)(qstep % (unsigned int)ntile_c);
    int og = (int)(q0 / (unsigned int)ntile_c), ct = (int)(q0 % (unsigned int)ntile_c);
    // lane-only LDS offsets of the RUN pairs (a = RUN lane + s and its mirror)
    const int a0 = lane * RUN;
    const int offA = padq(a0);
    // transform (the passes behind the staged one) and bin the C columns of the staged tile: half xh of x, y-row yr, column tile
    // columns per wave, and what a wave keeps of the first field's columns between the two halves of an interlaced tile
    constexpr int NCW = C / (XB_THREADS / 64);
    static_assert(NCW * (XB_THREADS / 64) == C, "whole columns per wave");
    float2 keepA[INTER ? NCW : 1][RUN], keepB[INTER ? NCW : 1][RUN], keepQ[INTER ? NCW : 1];
    float part[QUAD ? NCW : 1][RUN], partQ[QUAD ? NCW : 1];
    // mode 0: transform and bin (one field); 1: transform and keep (first field of an interlaced pair); 2: transform, combine
    // with what was kept, bin; QUAD: 3: combine with what was kept and keep the sum, 4: first half of the cross product, kept
    // per pair of modes, 5: second half, bin
    auto process = [&](const int xh, const int yr, const int ct_cur, const int mode) {
        const int j = ((yr & (H - 1)) << 1) | (yr >= H ? 1 : 0);
        const int jj = j < n / 2 ? j : j - n;
        // mirrors: xh = 1: H-1-a; xh = 0: H-a (a >= 1), a = 0 is its own mirror (i = 0, one mode)
        const int offB = xh ? padq(H - 1 - a0) : padq(H - 1 - a0) + 1;      // minus s (xh = 0: s >= 1)
        const int offB0 = xh ? offB : padq((H - a0) & (H - 1));
        const float mB0 = (!xh && lane == 0) ? 0.f : 1.f;
#pragma unroll
        for (int ci = 0; ci < NCW; ci++) {
            const int c = wave + ci * (XB_THREADS / 64);
            float2 *col = lds + c * CP;
            const int k = ct_cur * C + c;
            const int r2 = jj * jj + k * k;
            // padding columns of the last tile and columns that lie beyond the last edge as a whole: neither transformed nor binned
            if (!(g.dbg & 16) && (k >= g.kzlen || r2 > d.vtop)) continue;
            if (!(g.dbg & 1)) {
                if (INTER && H == 1024) {   // one column per wave and 32 registers of kept values: the lane's twiddles only while the pass runs
                    float2 twl[8];
                    twl[0] = make_float2(1.f, 0.f);
#pragma unroll
                    for (int r = 1; r < 8; r++) twl[r] = twH[(lane % (H / 64)) * r * 8];
                    dif_pass_w_regtw<H, H / 8, 8>(col, twl, lane);
                } else
                dif_pass_w_regtw<H, H / 8, 8>(col, tw2, lane);          // the passes behind the one stage() performed
                PassesW<H, H / 64>::run(col, nullptr, lane);            // last pass: no twiddles
                wave_sync();
            }
            if (k >= g.kzlen || (g.dbg & 2)) continue;
            if (r2 > d.vtop) continue;                     // the whole column lies beyond the last edge
            int Uk[MU > 1 ? MU - 1 : 1];
#pragma unroll
            for (int m = 0; m < MU - 1; m++) Uk[m] = Ul[k * (MU - 1) + m];
            const float k2f = (float)(k * k);
            const float scale = (k == 0 ? 1.f : 2.f) * inv2;   // weight (:258-262) times f32(1/M)^2 (:1058-1060)
            float wjk = 1.f;
            if (COMP) wjk = Wl[j] * Wl[k];
            float2 vA[RUN], vB[RUN];
            if (g.dbg & 8) {
#pragma unroll
                for (int s = 0; s < RUN; s++) vA[s] = vB[s] = make_float2(1.f, (float)(a0 + s));
            } else {
#pragma unroll
                for (int s = 0; s < RUN; s++) vA[s] = col[offA + s];
                vB[0] = col[offB0];
#pragma unroll
                for (int s = 1; s < RUN; s++) vB[s] = col[offB - s];
            }
            if (INTER && mode == 1) {          // first field: keep the column, bin nothing
#pragma unroll
                for (int s = 0; s < RUN; s++) keepA[INTER ? ci : 0][s] = vA[s], keepB[INTER ? ci : 0][s] = vB[s];
                keepQ[INTER ? ci : 0] = (!xh && lane == 63) ? col[padq(H / 2)] : make_float2(0.f, 0.f);
                continue;
            }
            const int i0 = 2 * a0 + xh;
            float psum[QUAD ? RUN : 1], pq = 0.f;
            if constexpr (QUAD) {
                float2 q = (!xh && lane == 63) ? col[padq(H / 2)] : make_float2(0.f, 0.f);   // i = n/2: one mode, last lane of the even half
                if (mode == 3 || mode == 5) {       // the shifted deposits: times e^{i pi m / n}, m = i + j + k (INTER's phases)
                    const int mjk = jj + k;
                    int mp = mjk + i0, mm = mjk - i0;
                    mp += mp < 0 ? 2 * n : 0, mm += mm < 0 ? 2 * n : 0;
                    float2 pp = g.phase[mp], pm = g.phase[mm];
                    const float2 rot = g.phase[2];
#pragma unroll
                    for (int s = 0; s < RUN; s++) {
                        vA[s] = make_float2(vA[s].x * pp.x - vA[s].y * pp.y, vA[s].x * pp.y + vA[s].y * pp.x);
                        vB[s] = make_float2(vB[s].x * pm.x - vB[s].y * pm.y, vB[s].x * pm.y + vB[s].y * pm.x);
                        pp = make_float2(pp.x * rot.x - pp.y * rot.y, pp.x * rot.y + pp.y * rot.x);
                        pm = make_float2(pm.x * rot.x + pm.y * rot.y, pm.y * rot.x - pm.x * rot.y);
                    }
                    int m = mjk - H;               // i = n/2 folds to -n/2 (shift_field_fft, power_spectrum.py:940-942)
                    m += m < 0 ? 2 * n : 0;
                    const float2 ph = g.phase[m];
                    q = make_float2(q.x * ph.x - q.y * ph.y, q.x * ph.y + q.y * ph.x);
                }
                if (mode == 3) {                    // K = a + a' e^{..}
#pragma unroll
                    for (int s = 0; s < RUN; s++) {
                        keepA[QUAD ? ci : 0][s].x += vA[s].x, keepA[QUAD ? ci : 0][s].y += vA[s].y;
                        keepB[QUAD ? ci : 0][s].x += vB[s].x, keepB[QUAD ? ci : 0][s].y += vB[s].y;
                    }
                    keepQ[QUAD ? ci : 0].x += q.x, keepQ[QUAD ? ci : 0].y += q.y;
                    continue;
                }
#pragma unroll
                for (int s = 0; s < RUN; s++) {    // w Re(conj(K) v) of the pair (i, -i): same |k|, same mu, one bin
                    const float2 a = keepA[QUAD ? ci : 0][s], bq = keepB[QUAD ? ci : 0][s];
                    float pA = a.x * vA[s].x + a.y * vA[s].y, pB = bq.x * vB[s].x + bq.y * vB[s].y;
                    if (COMP) {
                        const int fA = a0 + s, fB = xh ? H - 1 - fA : (H - fA) & (H - 1);
                        const float sA = __builtin_amdgcn_rcpf(Wl[2 * fA + xh] * wjk), sB = __builtin_amdgcn_rcpf(Wl[2 * fB + xh] * wjk);
                        pA *= sA * sA, pB *= sB * sB;
                    }
                    if (s == 0) pB *= mB0;
                    psum[s] = pA + pB;
                }
                pq = keepQ[QUAD ? ci : 0].x * q.x + keepQ[QUAD ? ci : 0].y * q.y;
                if (COMP) {
                    const float sc = __builtin_amdgcn_rcpf(Wl[H] * wjk);
                    pq *= sc * sc;
                }
                if (mode == 4) {
#pragma unroll
                    for (int s = 0; s < RUN; s++) part[QUAD ? ci : 0][s] = psum[s];
                    partQ[QUAD ? ci : 0] = pq;
                    continue;
                }
#pragma unroll
                for (int s = 0; s < RUN; s++) psum[s] += part[QUAD ? ci : 0][s];
                pq += partQ[QUAD ? ci : 0];
            } else if constexpr (CROSS) {              // second field b: Re(conj(a) b), left in .x (the power below takes it from there)
#pragma unroll
                for (int s = 0; s < RUN; s++) {
                    const float2 a = keepA[INTER ? ci : 0][s], bq = keepB[INTER ? ci : 0][s];
                    vA[s].x = a.x * vA[s].x + a.y * vA[s].y;
                    vB[s].x = bq.x * vB[s].x + bq.y * vB[s].y;
                }
            } else if (INTER) {                 // second field: a + a' exp(i pi m / n), m = i + j + k folded into [0, 2n)
                // the phases of a lane's run from the table at its first pair, then by the rotation exp(+-2 pi i / n) per step
                // (i advances by 2): two table reads per lane and column instead of 2 RUN gathers over a 16-KB table
                const int mjk = jj + k;
                int mp = mjk + i0, mm = mjk - i0;
                mp += mp < 0 ? 2 * n : 0, mm += mm < 0 ? 2 * n : 0;
                float2 pp = g.phase[mp], pm = g.phase[mm];
                const float2 rot = g.phase[2];
#pragma unroll
                for (int s = 0; s < RUN; s++) {
                    const float2 a = keepA[INTER ? ci : 0][s], bq = keepB[INTER ? ci : 0][s];
                    vA[s] = make_float2(a.x + (vA[s].x * pp.x - vA[s].y * pp.y), a.y + (vA[s].x * pp.y + vA[s].y * pp.x));
                    vB[s] = make_float2(bq.x + (vB[s].x * pm.x - vB[s].y * pm.y), bq.y + (vB[s].x * pm.y + vB[s].y * pm.x));
                    pp = make_float2(pp.x * rot.x - pp.y * rot.y, pp.x * rot.y + pp.y * rot.x);
                    pm = make_float2(pm.x * rot.x + pm.y * rot.y, pm.y * rot.x - pm.x * rot.y);
                }
            }
            int v = r2 + i0 * i0, inc = 4 * i0 + 4;
            int cur = 0, curk = 0;
            float sp = 0.f, s2 = 0.f, s4 = 0.f;
            auto flush = [&]() {
                if (!(g.dbg & 4)) {
                    atomicAdd(&h_sum[cur], (double)sp);
                    if (NP > 0) {
                        atomicAdd(&h_m2[curk], (double)s2);
                        atomicAdd(&h_m4[curk], (double)s4);
                    }
                }
            };
            // where a mode goes: independent of the data - the RUN look-ups of a lane's run are issued together, ahead of the
            // run-length accumulation below (one after the other, each behind its own LDS round trip, they were a third of the
            // binning's time at two waves per SIMD)
            auto locate = [&](int vv, int &tb, int &eb, float &mu2) {
                const float vf1 = fmaxf((float)vv, 1.f);            // kmag2 = 0: the cell of 1, mu2 = 0 (:243)
                eb = xd_eb(lut0, sh, vv, vf1);
                int bmu = 0;
#pragma unroll
                for (int m = 0; m < MU - 1; m++) bmu += vv <= Uk[m] ? 1 : 0;
                tb = (int)__umul24(eb, Nmu) + bmu;
                mu2 = NP > 0 ? k2f * __builtin_amdgcn_rcpf(vf1) : 0.f;
            };
            auto bin = [&](int tb, int eb, float mu2, float p, bool first) {
                const float pw = p * scale;
                float t2 = 0.f, t4 = 0.f;
                if (NP > 0) {
                    t2 = pw * mu2;
                    t4 = t2 * mu2;
                }
                if (RUNS) {
                    if (first) {
                        cur = tb, curk = eb;
                    } else if (tb != cur) {
                        flush();
                        cur = tb, curk = eb;
                        sp = s2 = s4 = 0.f;
                    }
                    sp += pw, s2 += t2, s4 += t4;
                } else if ((unsigned int)(eb - 1) < (unsigned int)Nk && !(g.dbg & 4)) {
                    atomicAdd(&h_sum[tb], (double)pw);
                    if (NP > 0) {
                        atomicAdd(&h_m2[eb], (double)t2);
                        atomicAdd(&h_m4[eb], (double)t4);
                    }
                }
            };
            constexpr bool HOIST = !INTER;       // (the pair forms hold a kept column in registers: no room for the look-ups of a run)
            int tbs[HOIST ? RUN : 1], ebs[HOIST ? RUN : 1];
            float mu2s[HOIST ? RUN : 1];
            if constexpr (HOIST) {
#pragma unroll
                for (int s = 0; s < RUN; s++) {
                    locate(v, tbs[s], ebs[s], mu2s[s]);
                    v += inc, inc += 8;
                }
            }
#pragma unroll
            for (int s = 0; s < RUN; s++) {
                float pA = QUAD ? psum[QUAD ? s : 0] : CROSS ? vA[s].x : vA[s].x * vA[s].x + vA[s].y * vA[s].y;      // get_raw_power (:726)
                float pB = QUAD ? 0.f : CROSS ? vB[s].x : vB[s].x * vB[s].x + vB[s].y * vB[s].y;
                if (COMP && !QUAD) {                                   // (:1065-1069)
                    const int fA = a0 + s, fB = xh ? H - 1 - fA : (H - fA) & (H - 1);
                    const float sA = __builtin_amdgcn_rcpf(Wl[2 * fA + xh] * wjk), sB = __builtin_amdgcn_rcpf(Wl[2 * fB + xh] * wjk);
                    pA *= sA * sA, pB *= sB * sB;
                }
                if (s == 0) pB *= mB0;
                if constexpr (HOIST) {
                    bin(tbs[s], ebs[s], mu2s[s], pA + pB, s == 0);
                } else {
                    int tb1, eb1;
                    float mu21;
                    locate(v, tb1, eb1, mu21);
                    v += inc, inc += 8;
                    bin(tb1, eb1, mu21, pA + pB, s == 0);
                }
            }
            if (!xh && lane == 63) {       // i = n/2 (a = H/2): one mode, last lane of the even half
                float2 q = (g.dbg & 8) ? make_float2(1.f, 1.f) : col[padq(H / 2)];
                float p;
                if constexpr (QUAD) {
                    p = pq;
                } else if constexpr (CROSS) {
                    const float2 a = keepQ[INTER ? ci : 0];
                    p = a.x * q.x + a.y * q.y;
                } else {
                    if (INTER) {
                        int m = jj + k - H;            // i = n/2 folds to -n/2 (shift_field_fft, power_spectrum.py:940-942)
                        m += m < 0 ? 2 * n : 0;
                        const float2 ph = g.phase[m], a = keepQ[INTER ? ci : 0];
                        q = make_float2(a.x + (q.x * ph.x - q.y * ph.y), a.y + (q.x * ph.y + q.y * ph.x));
                    }
                    p = q.x * q.x + q.y * q.y;
                }
                if (COMP && !QUAD) {
                    const float sc = __builtin_amdgcn_rcpf(Wl[H] * wjk);
                    p *= sc * sc;
                }
                int tbq, ebq;
                float mu2q;
                locate(r2 + H * H, tbq, ebq, mu2q);
                bin(tbq, ebq, mu2q, p, false);
            }
            if (RUNS && (unsigned int)(curk - 1) < (unsigned int)Nk) flush();
        }
    };
    __syncthreads();
    // DEAD TILES: with ky^2 + kz^2 beyond the last edge already at the tile's first column no mode of the tile is binned,
    // whatever kx - the corners of the (ky, kz) plane outside the circle of radius k_max: 21 % of the tiles when the bins end
    // at the Nyquist frequency.  They are neither loaded nor transformed (N_mode and k_avg come from the cached geometry).
    auto dead = [&](int o, int ctile) {
        const int xh = o >= g.ny ? 1 : 0, yr = g.y0 + o - xh * g.ny;
        const int j = ((yr & (H - 1)) << 1) | (yr >= H ? 1 : 0), jj = j < n / 2 ? j : j - n, k0 = ctile * C;
        return !(g.dbg & 16) && jj * jj + k0 * k0 > d.vtop;
    };
    auto step = [&]() {
        do {
            og += dg, ct += dc;
            if (ct >= ntile_c) ct -= ntile_c, og++;
        } while (og < n_og && dead(og * ostep + grp, ct));
    };
    if (og < n_og && dead(og * ostep + grp, ct)) step();
    if (og < n_og) {
        prefetch(tile_ptr(og * ostep + grp, ct, data));
        wait_vmcnt<0>();
        stage();
        for (;;) {
            __syncthreads();
            const int o_cur = og * ostep + grp, ct_cur = ct;
            const int xh = o_cur >= g.ny ? 1 : 0;
            if (INTER) {         // the shifted field's tile follows the unshifted one through the same LDS
                prefetch(tile_ptr(o_cur, ct_cur, g.data2));
                process(xh, g.y0 + o_cur - xh * g.ny, ct_cur, 1);
                __syncthreads();
                wait_vmcnt<0>();
                stage();
                __syncthreads();
            }
            if (QUAD) {          // ... then the second catalogue's pair
                prefetch(tile_ptr(o_cur, ct_cur, g.data3));
                process(xh, g.y0 + o_cur - xh * g.ny, ct_cur, 3);
                __syncthreads();
                wait_vmcnt<0>();
                stage();
                __syncthreads();
                prefetch(tile_ptr(o_cur, ct_cur, g.data4));
                process(xh, g.y0 + o_cur - xh * g.ny, ct_cur, 4);
                __syncthreads();
                wait_vmcnt<0>();
                stage();
                __syncthreads();
            }
            step();
            const bool has_next = og < n_og;
            if (has_next) prefetch(tile_ptr(og * ostep + grp, ct, data));
            process(xh, g.y0 + o_cur - xh * g.ny, ct_cur, QUAD ? 5 : INTER ? 2 : 0);
            if (!has_next) break;
            __syncthreads();     // every wave is done with the tile
            wait_vmcnt<0>();     // no stores in this kernel: the prefetch is all that is outstanding
            stage();
        }
    }
    __syncthreads();
    for (int q = tid; q < Nk * Nmu; q += XB_THREADS) {
        const double s = h_sum[q + Nmu];           // row eb = bk + 1
        if (s != 0.0) atomicAdd(&b.g_sum[q], s);
        if (blockIdx.x == 0 && g.put_geom) b.g_cnt[q] = d.cnt[q], b.g_ksum[q] = d.ksum[q];
    }
    if (NP > 0) {
        float pc[NPC][3];
#pragma unroll
        for (int q = 0; q < NPC; q++)
#pragma unroll
            for (int m = 0; m < 3; m++) pc[q][m] = (q < NP && m <= b.poledeg[q]) ? b.polecoef[q][m] : 0.f;
        for (int bk = tid; bk < Nk; bk += XB_THREADS) {
            double s0 = 0.0;
            for (int m = 0; m < Nmu; m++) s0 += h_sum[(bk + 1) * Nmu + m];
            const double m2 = h_m2[bk + 1], m4 = h_m4[bk + 1];
#pragma unroll
            for (int q = 0; q < NPC; q++)
                if (q < NP) {
                    const double vq = (double)pc[q][0] * s0 + (double)pc[q][1] * m2 + (double)pc[q][2] * m4;
                    if (vq != 0.0) atomicAdd(&b.g_pole[q * Nk + bk], vq);
                }
        }
    }
}

template <int H, int C>
size_t xbin2_lds_bytes(int n, int Nk, int Nmu, int ncell, bool comp) {
    const int mu = Nmu <= 1 ? 1 : Nmu <= 4 ? 4 : 8;
    return (size_t)(C * colpitch_of<H>() + 1) * sizeof(float2) + (size_t)(Nk + 2) * Nmu * 8 + (size_t)2 * (Nk + 2) * 8 +
           (size_t)ncell * 4 + (size_t)(n / 2 + 1) * (mu - 1) * 4 + (comp ? (size_t)n * 4 : 0) + 16;
}

// ---- descriptor cache (host) ---------------------------------------------------------------------------------------
struct XDescHost {
    int n = 0, Nk = 0, Nmu = 0;
    bool comp = false;         // part of the key: the window table shares the LDS the cell table is sized for (lds_other)
    std::vector<float> e2;     // kedges2 (Nk+1) then muedges2 (Nmu+1)
    bool ok = false;
    int ncell = 0, sh = 0, off = 0, vtop = 0;
    DevBuf buf;                // [cnt u64 nb][ksum f64 nb][lut u32 ncell][U int kzlen*XD_USTRIDE][flag int]
    uint64_t stamp = 0;
    double build_ms = 0;
    XDesc dev() const {
        XDesc d;
        const size_t nb = (size_t)Nk * Nmu;
        unsigned char *p = static_cast<unsigned char *>(buf.p);
        d.cnt = reinterpret_cast<const unsigned long long *>(p);
        d.ksum = reinterpret_cast<const double *>(p + nb * 8);
        d.lut = reinterpret_cast<const unsigned int *>(p + nb * 16);
        d.U = reinterpret_cast<const int *>(p + nb * 16 + (size_t)ncell * 4);
        d.ncell = ncell, d.sh = sh, d.off = off, d.ustride = XD_USTRIDE, d.vtop = vtop;
        return d;
    }
};
std::vector<XDescHost *> g_desc;
uint64_t g_desc_clock = 0;
constexpr size_t XD_MAX_CACHED = 8;

int float_bits(float f) {
    int b;
    memcpy(&b, &f, 4);
    return b;
}
float bits_float(int b) {
    float f;
    memcpy(&f, &b, 4);
    return f;
}

// host half of the descriptor: thresholds, cell table, mu thresholds.  false: these edges do not fit the scheme
bool xdesc_tables(int n, int Nk, int Nmu, const float *ke, const float *me, size_t lds_other, std::vector<unsigned int> &lut,
                  std::vector<int> &U, int &sh, int &off, int &vtop) {
    if (Nk + 1 > 1023 || Nmu > XD_USTRIDE || n > 2048) return false;
    const int64_t vmax = (int64_t)3 * (n / 2) * (n / 2);
    if (vmax >= XD_TMASK) return false;
    for (int q = 0; q <= Nk; q++)
        if (!(ke[q] == ke[q]) || (q && ke[q] < ke[q - 1])) return false;     // NaN or descending edges
    for (int q = 0; q <= Nmu; q++)
        if (!(me[q] == me[q])) return false;
    auto clampi = [](double x) { return (int)std::min(std::max(x, -1.0), 1073741824.0); };
    std::vector<int> T((size_t)Nk + 2);
    T[0] = clampi(std::ceil((double)ke[0]) - 1.0);                              // kmag2 < edges[0]   <=> kmag2 <= T[0]
    for (int e = 1; e < Nk; e++) T[e] = clampi(std::floor((double)ke[e]));      // kmag2 > edges[e]   <=> kmag2 >  T[e]
    T[Nk] = clampi(std::ceil((double)ke[Nk]) - 1.0);                            // kmag2 >= edges[Nk] <=> kmag2 >  T[Nk]
    T[Nk + 1] = 2147483647;
    for (int e = 1; e <= Nk; e++) T[e] = std::max(T[e], T[e - 1]);
    auto eb_true = [&](int64_t v) { return (int)(std::lower_bound(T.begin(), T.begin() + Nk + 1, (int)std::min<int64_t>(v, 2147483646)) - T.begin()); };
    vtop = (int)std::min<int64_t>(T[Nk], vmax);
    if (vtop < 0) vtop = 0;
    bool found = false;
    for (int m = 6; m <= 10 && !found; m++) {
        sh = 23 - m, off = 127 << m;
        const int ncell = (float_bits((float)vmax) >> sh) - off + 1;     // every kmag2 of the mesh has its cell: no clamp
        if (lds_other + (size_t)ncell * 4 > 160 * 1024) break;
        lut.assign(ncell, 0u);
        bool good = true;
        for (int c = 0; c < ncell && good; c++) {
            int64_t vlo = c == 0 ? 0 : (int64_t)std::ceil((double)bits_float((c + off) << sh));
            int64_t vhi = (int64_t)std::floor((double)bits_float((((c + off + 1) << sh)) - 1));
            if (vlo > vhi) vhi = vlo;    // a cell without an integer: never looked up
            const int e0 = eb_true(vlo), e1 = eb_true(vhi);
            good = e1 - e0 <= 1;
            lut[c] = ((unsigned int)e0 << 22) | (unsigned int)std::min(std::max(T[e0], 0), XD_TMASK);
        }
        found = good;
    }
    if (!found) return false;
    const int kzlen = n / 2 + 1;
    U.assign((size_t)kzlen * XD_USTRIDE, -1);
    const int VMAX = 1 << 23;
    for (int k = 0; k < kzlen; k++) {
        const float k2f = (float)(k * k);
        auto above = [&](int v, float edge) { return k2f * (1.0f / (float)v) > edge; };
        for (int m = 0; m < Nmu - 1; m++) {
            const float edge = me[m + 1];
            int u;
            if (0.f > edge) u = 2147483647;          // mu2 = 0 (kmag2 = 0 included) already lies above
            else if (!above(1, edge)) u = -1;
            else if (above(VMAX, edge)) u = 2147483647;
            else {
                int lo = 1, hi = VMAX;               // above(lo), !above(hi)
                while (hi - lo > 1) {
                    const int mid = lo + (hi - lo) / 2;
                    (above(mid, edge) ? lo : hi) = mid;
                }
                u = lo;
            }
            U[(size_t)k * XD_USTRIDE + m] = u;
        }
    }
    return true;
}

// descriptor of (n, edges): cached; built (tables + xbin_geometry over every mode) on first use
int xdesc_get(int n, int Nk, int Nmu, bool comp, const float *h_e2, const float *d_ke, const float *d_me, size_t lds_other, XDescHost **out) {
    const size_t ne = (size_t)Nk + 1 + Nmu + 1;
    for (XDescHost *x : g_desc)
        if (x->n == n && x->Nk == Nk && x->Nmu == Nmu && x->comp == comp && !memcmp(x->e2.data(), h_e2, ne * 4)) {
            x->stamp = ++g_desc_clock;
            *out = x;
            return 0;
        }
    if (g_desc.size() >= XD_MAX_CACHED) {
        size_t old = 0;
        for (size_t q = 1; q < g_desc.size(); q++)
            if (g_desc[q]->stamp < g_desc[old]->stamp) old = q;
        HIP_TRY(hipStreamSynchronize(stream()));
        ABACUS_TRY(g_desc[old]->buf.release());
        delete g_desc[old];
        g_desc.erase(g_desc.begin() + old);
    }
    XDescHost *x = new XDescHost;
    x->n = n, x->Nk = Nk, x->Nmu = Nmu, x->comp = comp, x->e2.assign(h_e2, h_e2 + ne), x->stamp = ++g_desc_clock;
    g_desc.push_back(x);
    *out = x;
    std::vector<unsigned int> lut;
    std::vector<int> U;
    if (!xdesc_tables(n, Nk, Nmu, h_e2, h_e2 + Nk + 1, lds_other, lut, U, x->sh, x->off, x->vtop)) return 0;   // ok stays false
    x->ncell = (int)lut.size();
    const size_t nb = (size_t)Nk * Nmu;
    if (nb * 12 > 96 * 1024) return 0;
    const size_t bytes = nb * 16 + lut.size() * 4 + U.size() * 4 + 16;
    ABACUS_TRY(x->buf.reserve(bytes));
    unsigned char *p = static_cast<unsigned char *>(x->buf.p);
    HIP_TRY(hipMemsetAsync(p, 0, bytes, stream()));
    HIP_TRY(hipMemcpyAsync(p + nb * 16, lut.data(), lut.size() * 4, hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipMemcpyAsync(p + nb * 16 + lut.size() * 4, U.data(), U.size() * 4, hipMemcpyHostToDevice, stream()));
    int *flag = reinterpret_cast<int *>(p + nb * 16 + lut.size() * 4 + U.size() * 4);
    const XDesc d = x->dev();
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(xbin_geometry), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(nb * 12)));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, stream()));
    ABACUS_LAUNCH("xbin_geometry", xbin_geometry, dim3(fft_num_cus() * 4), dim3(256), nb * 12, n, Nk, Nmu, d_ke, d_me, d,
                  const_cast<unsigned long long *>(d.cnt), const_cast<double *>(d.ksum), flag);
    HIP_TRY(hipEventRecord(e1, stream()));
    int bad = 1;
    HIP_TRY(hipMemcpyAsync(&bad, flag, 4, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));     // lut, U are locals
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    HIP_TRY(hipEventDestroy(e0));
    HIP_TRY(hipEventDestroy(e1));
    x->build_ms = ms;
    x->ok = bad == 0;
    return 0;
}

template <int H, int C, int NP, bool COMP, int MU>
int launch_xbin2(const float2 *data, const XBinGeom &g, const BinArgs &b, const XDesc &d, size_t lds) {
    const bool runs = option("pk_xbin_pairs") == 0;
    auto kern = g.data3 ? fft_x_bin2<H, C, NP, COMP, MU, true, true, false, true>
                : g.data2 ? (g.phase ? fft_x_bin2<H, C, NP, COMP, MU, true, true> : fft_x_bin2<H, C, NP, COMP, MU, true, true, true>)
                          : (runs ? fft_x_bin2<H, C, NP, COMP, MU, true> : fft_x_bin2<H, C, NP, COMP, MU, false>);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 1;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, XB_THREADS, lds));
    const int64_t ntiles = (int64_t)2 * g.ny * ((g.kzlen + C - 1) / C);
    const unsigned int grid = (unsigned int)std::min<int64_t>(ntiles, (int64_t)fft_num_cus() * std::max(per_cu, 1));
    const float2 *tw = fft_twiddles(H);
    if (!tw) return -1;
    ABACUS_LAUNCH("fft_x_bin", kern, dim3(grid), dim3(XB_THREADS), lds, data, g, b, d, tw);
    return 0;
}

template <int H, int C>
int dispatch_xbin2(const float2 *data, const XBinGeom &g, const BinArgs &b, const XDesc &d, size_t lds) {
#define XB2(NP, MU)                                                                  \
    (g.W ? launch_xbin2<H, C, NP, true, MU>(data, g, b, d, lds) : launch_xbin2<H, C, NP, false, MU>(data, g, b, d, lds))
#define XB2_MU(NP) (b.Nmu <= 1 ? XB2(NP, 1) : b.Nmu <= 4 ? XB2(NP, 4) : XB2(NP, 8))
    switch (b.Np) {
        case 0: return XB2_MU(0);
        case 1: return XB2_MU(1);
        case 2: return XB2_MU(2);
    }
#undef XB2_MU
#undef XB2
    return fail("fft_x_bin: %d multipoles", b.Np);
}

size_t xbin2_lds_other(int n, int Nk, int Nmu, bool comp) {
    return n == 2048 ? xbin2_lds_bytes<1024, 8>(n, Nk, Nmu, 0, comp) : xbin2_lds_bytes<512, 16>(n, Nk, Nmu, 0, comp);
}

}  // namespace

namespace abacus {

// can the fused last pass serve this mesh / histogram?  (n/2-point wave-local transforms: n = 1024, 2048; at most two
// ell != 0 multipoles of degree <= 4; tile + histogram within the 160 KiB LDS).  Second generation (cached geometry
// descriptor, <= 8 mu bins): builds the descriptor of (n, edges) on first use.
static bool xbin1_supported(int n, int Nk, int Nmu, const BinArgs &b, bool comp) {
    const size_t cap = 160 * 1024;
    if (n == 2048) return xbin_lds_bytes<1024, 8>(n, Nk, Nmu, b.Np, comp) <= cap;
    if (n == 1024) return xbin_lds_bytes<512, 16>(n, Nk, Nmu, b.Np, comp) <= cap;
    return false;
}

static int xbin2_desc(int n, const BinArgs &b, bool comp, XDescHost **x) {
    *x = nullptr;
    if (option("pk_xbin_gen") == 1 || !b.h_edges2 || (n != 1024 && n != 2048) || b.Nmu > XD_USTRIDE) return 0;
    const size_t other = xbin2_lds_other(n, b.Nk, b.Nmu, comp);
    if (other + 64 * 4 > 160 * 1024) return 0;
    ABACUS_TRY(xdesc_get(n, b.Nk, b.Nmu, comp, b.h_edges2, b.kedges2, b.muedges2, other, x));
    if (!(*x)->ok || other + (size_t)(*x)->ncell * 4 > 160 * 1024) *x = nullptr;
    return 0;
}

bool xbin_supported(int n, int Nk, int Nmu, const BinArgs &b, bool comp) {
    if (b.Np > 2) return false;
    for (int q = 0; q < b.Np; q++)
        if (b.poledeg[q] > 2) return false;
    if (n != 1024 && n != 2048) return false;
    if (xbin1_supported(n, Nk, Nmu, b, comp)) return true;
    XDescHost *x = nullptr;
    return xbin2_desc(n, b, comp, &x) == 0 && x != nullptr;
}

// the descriptor of (n, edges of b) for a kernel that keeps `lds_other` bytes of LDS beside the cell table; ok = 0: these edges do
// not fit the scheme (or any other mesh size than the two of the fused power-of-two pass asks with its own lds_other)
int xdesc_lookup(int n, const BinArgs &b, bool comp, size_t lds_other, const unsigned int **lut, const int **U, int *ncell, int *sh, int *off,
                 int *vtop, const unsigned long long **cnt, const double **ksum, int *ok) {
    *ok = 0;
    if (!b.h_edges2 || b.Nmu > XD_USTRIDE || lds_other + 64 * 4 > 160 * 1024) return 0;
    XDescHost *x = nullptr;
    ABACUS_TRY(xdesc_get(n, b.Nk, b.Nmu, comp, b.h_edges2, b.kedges2, b.muedges2, lds_other, &x));
    if (!x->ok || lds_other + (size_t)x->ncell * 4 > 160 * 1024) return 0;
    const XDesc d = x->dev();
    *lut = d.lut, *U = d.U, *ncell = d.ncell, *sh = d.sh, *off = d.off, *vtop = d.vtop, *cnt = d.cnt, *ksum = d.ksum;
    *ok = 1;
    return 0;
}

// milliseconds the geometry pass of the most recently used descriptor took when it was built (bench.py reports it next to
// the step time: it is paid once per (nmesh, edges))
double xbin_last_build_ms() {
    const XDescHost *best = nullptr;
    for (const XDescHost *x : g_desc)
        if (!best || x->stamp > best->stamp) best = x;
    return best ? best->build_ms : 0.0;
}

static int g_last_gen = 0;
int xbin_last_gen() { return g_last_gen; }

int xbin_release() {
    for (XDescHost *x : g_desc) {
        ABACUS_TRY(x->buf.release());
        delete x;
    }
    g_desc.clear();
    return 0;
}

// `mesh` holds the fused transform after its z and y passes (fft_native_r2c_fused_zy); bins |delta_k|^2 of every mode
// into the accumulators of `b` (zeroed by the caller).  layout 0: the whole mesh (x, y, k); 1: the y-slab [y0, y0 +
// ny_local) of a multi-GPU transform unpacked to (y_local, x, k); 2: the same y-slab as the pencil transpose of a
// `world`-rank run delivers it, (peer, 2 h, y_local, k), h = n / (2 world) (folded slabs: each peer sends the sum rows, then
// the difference rows of its h plane pairs; with one rank that IS the whole mesh in place).  Only the cached-geometry kernel
// serves the slab forms, and `put_geom` says whether this rank contributes the mesh-wide N_mode / sum |k| to the
// histogram that is all-reduced afterwards.
int fft_x_bin_run(const float *mesh, int n, int pitch_r, float inv_size, const float *W_dev, const BinArgs &b, int dbg,
                  int y0, int ny_local, int put_geom, int layout, int world, const float *mesh_shifted, const float2 *phase,
                  const unsigned int *row_off, int64_t plane_elems, const float *mesh_b, const float *mesh_b_shifted) {
    XBinGeom g;
    g.n = n, g.kzlen = n / 2 + 1, g.pitch_c = pitch_r / 2, g.inv_size = inv_size, g.W = W_dev, g.dbg = dbg;
    if (mesh_shifted) {
        // phase table: the interlaced pair; none: `mesh_shifted` is a second field, cross power - in the layout of `mesh` either way
        g.data2 = reinterpret_cast<const float2 *>(mesh_shifted), g.phase = phase;
        g.half_inv_size = (float)(0.5 / ((double)n * n * n));
    }
    if (mesh_b || mesh_b_shifted) {     // the interlaced cross power: (mesh, mesh_shifted) x (mesh_b, mesh_b_shifted)
        if (!mesh_shifted || !phase || !mesh_b || !mesh_b_shifted) return fail("fft_x_bin: the four-field form needs both interlaced pairs and the phase table");
        g.data3 = reinterpret_cast<const float2 *>(mesh_b), g.data4 = reinterpret_cast<const float2 *>(mesh_b_shifted);
    }
    const bool slab = layout != 0;
    if (slab && ny_local < 1) return fail("fft_x_bin: empty y-slab");
    g.ny = slab ? ny_local : n, g.y0 = slab ? y0 : 0, g.put_geom = slab ? put_geom : 1;
    int h = n / 2;
    g.ps = 0;
    if (layout == 1) g.xs = g.pitch_c, g.ys = (int64_t)n * g.pitch_c;
    else if (layout == 2) {
        if (world < 1 || (n / 2) % world || ((n / 2 / world) & (n / 2 / world - 1)))
            return fail("fft_x_bin: %d ranks do not fold a mesh of %d into power-of-two runs of planes", world, n);
        h = n / 2 / world;
        g.xs = (int64_t)g.ny * g.pitch_c, g.ys = g.pitch_c, g.ps = 2 * (int64_t)h * g.xs;
        if (row_off) g.row_off = row_off, g.xs = plane_elems, g.ps = 2 * (int64_t)h * g.xs;   // compact transpose
    } else g.xs = (int64_t)n * g.pitch_c, g.ys = g.pitch_c;
    g.lgh = 0;
    while ((1 << g.lgh) < h) g.lgh++;
    const float2 *data = reinterpret_cast<const float2 *>(mesh);
    if (n != 2048 && n != 1024) return fail("fft_x_bin: unsupported mesh %d", n);
    const int C = n == 2048 ? 8 : 16;
    if (((g.kzlen + C - 1) / C) * C > g.pitch_c) return fail("fft_x_bin: row pitch too small");
    const bool comp = W_dev != nullptr;
    XDescHost *x = nullptr;
    ABACUS_TRY(xbin2_desc(n, b, comp, &x));
    g_last_gen = x ? 2 : 1;
    if (x) {
        const XDesc d = x->dev();
        if (n == 2048) return dispatch_xbin2<1024, 8>(data, g, b, d, xbin2_lds_bytes<1024, 8>(n, b.Nk, b.Nmu, d.ncell, comp));
        return dispatch_xbin2<512, 16>(data, g, b, d, xbin2_lds_bytes<512, 16>(n, b.Nk, b.Nmu, d.ncell, comp));
    }
    if (slab || mesh_shifted) return fail("fft_x_bin: no geometry descriptor for this histogram (y-slab / interlaced form)");
    if (!xbin1_supported(n, b.Nk, b.Nmu, b, comp)) return fail("fft_x_bin: histogram does not fit");
    if (n == 2048) return dispatch_xbin<1024, 8>(data, g, b, xbin_lds_bytes<1024, 8>(n, b.Nk, b.Nmu, b.Np, comp));
    return dispatch_xbin<512, 16>(data, g, b, xbin_lds_bytes<512, 16>(n, b.Nk, b.Nmu, b.Np, comp));
}

// the cached-geometry kernel alone (what a y-slab needs)
bool xbin2_supported(int n, const BinArgs &b, bool comp) {
    if (b.Np > 2) return false;
    for (int q = 0; q < b.Np; q++)
        if (b.poledeg[q] > 2) return false;
    XDescHost *x = nullptr;
    return xbin2_desc(n, b, comp, &x) == 0 && x != nullptr;
}

}  // namespace abacus
