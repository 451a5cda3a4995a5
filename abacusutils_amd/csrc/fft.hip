// Hand-written 3-D real-to-complex FFT for the P(k) path on MI355X (gfx950): replaces scipy.fft.rfftn at
// abacusnbody/analysis/power_spectrum.py:980,986,1059 for power-of-two meshes (64 <= n <= 2048); other sizes use
// hipFFT (power.hip).  Unnormalised forward transform, in place on the pitched mesh layout of power.hip.
//
// rocFFT runs this transform as six passes over the mesh (three 1-D FFTs + three transposes, 2.3 TB/s effective);
// here it is three passes, one per axis, each reading and writing the mesh exactly once:
//   fft_z_r2c      rows along z (unit stride): a length-n real FFT as a length-n/2 complex FFT in LDS plus the
//                  even/odd split post-processing; B rows per workgroup, 16-B coalesced loads.
//   fft_cols       columns along y, then along x (strided): a workgroup stages C adjacent columns (C*8 B contiguous
//                  per row: 128 B for n <= 1024, 64 B for n = 2048) x n rows in LDS, transposed so that every column
//                  is contiguous, transforms them in place and writes them back with the same access pattern.
// The LDS transform is an in-place radix-8/4/2 Sande-Tukey (DIF) FFT: no ping-pong buffer (at n = 2048 a tile of
// eight columns is 136 KiB of the 160 KiB LDS); the output is left digit-reversed in LDS and un-permuted for free by
// the write-back.  Twiddles come from a host-computed (float64 -> float32) table held in LDS.
// HBM traffic: 3 x (4M + 4M) bytes for a mesh of M cells = the 24 B/cell of SURVEY.md 8d.
#include <cmath>
#include <cstdlib>
#include <map>
#include <vector>

#include "common.hpp"

using namespace abacus;

namespace {

#include "fft_device.hpp"

// ---- z pass: real rows -> half-spectrum rows, in place ---------------------------------------------------------
// N = n/2.  mesh rows have `pitch_r` floats; row r of the (n*n) rows starts at r*pitch_r.  Persistent workgroups:
// while a tile of B rows is transformed in LDS the next tile is already in flight into registers.
// Loop shape (both kernels): the staging of tile t+1 (registers -> LDS) sits at the END of the body, after the
// write-back of tile t, so it is reached on a single path where the outstanding vector-memory operations are exactly
// "loads of t+1, then stores of t": the compiler then waits with a counted vmcnt that leaves the stores in flight
// (with the staging at the top of the loop the prologue path merges in and every wait degenerates to "all stores
// done", which exposes the full store latency once per tile).
// FUSE (B = 4, full n^3 mesh): a tile is the four rows (x, y), (x, y+n/2), (x+n/2, y), (x+n/2, y+n/2); after their z
// transforms the first radix-2 DIF stage of the y AND of the x transform is applied across them on the way out
// (a = u + v, b = (u - v) W^y, likewise along x).  The y and x passes are then two independent n/2-point column
// transforms each, over rows [0, n/2) and [n/2, n): half the rows per LDS tile, twice the columns - 128-B row
// segments at n = 2048, which the memory system serves at the in-place floor instead of 3.4 TB/s.
// Row r of either half then holds frequency 2 (r mod n/2) + (r div n/2) along that axis.
// ZFold: which plane pairs a fused launch works on.  Pair p (0 <= p < npair) is the planes p and xsep + p of `mesh` and stands
// for the global planes xg0 + p and xg0 + p + n/2 (the twiddle of the x stage is exp(-2 pi i (xg0 + p) / n)).  The whole
// mesh: npair = xsep = n/2, xg0 = 0; a rank of a multi-GPU run owns h pairs of planes n/2 apart (the folded slabs of
// analysis/slab_power.py), stored xsep = h + ghost planes apart.
struct ZFold {
    int npair, xsep, xg0;
    // out of place: the transformed rows go to mesh + dst_off (floats) instead of back where they came from - a pass that
    // reads one buffer and writes another streams at 5.4 TB/s where the in-place read-modify-write reaches 4.9
    // (scripts/ubench/inplace.hip)
    int64_t dst_off = 0;
};
template <int N, int B, int FUSE, int F1 = 0>
__global__ __launch_bounds__(Z_THREADS) void fft_z_r2c(float *__restrict__ mesh, int64_t nrows, int pitch_r,
                                                         const float2 *__restrict__ twN, const float2 *__restrict__ tw2N,
                                                         int dbg, ZFold zf) {
    constexpr int CP = colpitch_of<N>();
    constexpr int NLD = (B * (N / 2) + Z_THREADS - 1) / Z_THREADS;   // 16-B loads per thread and tile
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *tw = reinterpret_cast<float2 *>(smem);
    float2 *tw2 = tw + N;            // exp(-2 pi i k / 2N), k = 0..N (kept in LDS: a global load inside the loop would
    float2 *lds = tw2 + N + 2;       //  make the compiler drain the prefetch with vmcnt(0))
    const int tid = threadIdx.x;
    for (int q = tid; q < N; q += Z_THREADS) tw[q] = twN[q];
    for (int q = tid; q <= N; q += Z_THREADS) tw2[q] = tw2N[q];
    static_assert(!FUSE || B == 4, "the fused first stages work on 2 x 2 rows");
    constexpr int NF = 2 * N;                               // mesh size n
    const int64_t ntiles = FUSE ? (int64_t)zf.npair * N : (nrows + B - 1) / B;
    // first row of a tile, and the row of its r-th member
    auto row_of = [&](int64_t tile, int r) -> int64_t {
        if (!FUSE) return tile * B + r;
        const int64_t x = tile / N, y = tile % N;
        return (x + (r >> 1) * (int64_t)zf.xsep) * NF + y + (r & 1) * N;
    };
    const int pitch_c = pitch_r / 2;
    v4f regs[NLD];
    // WLS: one row per wave - a wave loads, stages and transforms its own row; only the write-back crosses rows
    constexpr bool WLS = wave_local(N) && B * 64 == Z_THREADS;
    auto prefetch = [&](int64_t tile) {
        const int nb = FUSE ? B : (int)min((int64_t)B, nrows - tile * B);
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int e = WLS ? (tid >> 6) * (N / 2) + q * 64 + (tid & 63) : q * Z_THREADS + tid;
            const int r = min(e / (N / 2), nb - 1), m = (e % (N / 2)) * 2;   // rows past the end re-read the last row
            gload16_async(regs[q], mesh + row_of(tile, r) * pitch_r + 2 * m);
        }
    };
    // FUSE1: the eight 16-B loads of a lane are elements 2 lane (+1) + 128 q of its wave's row = two butterflies of the
    // first radix-8 pass (stage_pass1); the passes that follow start at sub-length N/8
    constexpr bool FUSE1 = (F1 & 1) && WLS && N == 1024 && NLD == 8;
    constexpr bool REGTW = (F1 & 2) && WLS && N == 1024;   // twiddles of the second (and a fused first) pass as lane constants
    float2 tw1a[8], tw1b[8], tw2r[8];
    if constexpr (REGTW) {
#pragma unroll
        for (int r = 1; r < 8; r++) {
            tw1a[r] = twN[(2 * (tid & 63)) * r];
            tw1b[r] = twN[(2 * (tid & 63) + 1) * r];
            tw2r[r] = twN[((tid & 63) % (N >= 64 ? N / 64 : 1)) * r * 8];
        }
        tw1a[0] = tw1b[0] = tw2r[0] = make_float2(1.f, 0.f);
    }
    auto stage = [&]() {   // registers -> LDS: two complex (= four consecutive reals) per 16-B load
#pragma unroll
        for (int q = 0; q < NLD; q++) touch(regs[q]);
        if constexpr (FUSE1) {
            float2 *c = lds + (tid >> 6) * CP;
            if constexpr (REGTW) stage_pass1_regtw<N>(regs, c, c, 2 * (tid & 63), 2 * (tid & 63) + 1, tw1a, tw1b);
            else stage_pass1<N, NLD, N / 8>(regs, c, c, 2 * (tid & 63), 2 * (tid & 63) + 1, tw);
            return;
        }
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int e = WLS ? (tid >> 6) * (N / 2) + q * 64 + (tid & 63) : q * Z_THREADS + tid;
            const int r = e / (N / 2), m = (e % (N / 2)) * 2;
            if ((B * (N / 2)) % Z_THREADS == 0 || r < B) {
                float2 *c = lds + r * CP;
                c[padq(m)] = make_float2(regs[q].x, regs[q].y);
                c[padq(m + 1)] = make_float2(regs[q].z, regs[q].w);
            }
        }
    };
    int64_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    prefetch(tile);
    __syncthreads();     // the twiddle table, which a fused first pass reads while staging
    wait_vmcnt<0>();
    stage();
    // stores every thread issues per full tile (threads with one more only wait longer): vmcnt(that) = loads landed
    const int stores_min = FUSE ? 4 * ((pitch_c / 2) / Z_THREADS) : (B * (pitch_c / 2)) / Z_THREADS;
    if constexpr (WLS) __syncthreads();   // the twiddle tables (written by all waves) before the first transform
    for (;;) {
        if constexpr (WLS) wave_sync();   // the row staged by this wave is the row it transforms
        else __syncthreads();
        const int64_t next = tile + gridDim.x;
        const bool has_next = next < ntiles;
        if (has_next) prefetch(next);   // in flight during the transform and the write-back below
        const int64_t row0 = tile * B;
        const int nb = FUSE ? B : (int)min((int64_t)B, nrows - row0);
        if (!(dbg & 1)) {
            if constexpr (wave_local(N)) {   // one row per wave at a time; the write-back below reads across rows
#pragma unroll 1
                for (int r = tid >> 6; r < nb; r += Z_THREADS / 64) {
                    float2 *c = lds + r * CP;
                    if constexpr (REGTW) {
                        if constexpr (!FUSE1) dif_pass_w<N, N, 8, false>(c, tw, tid & 63);
                        dif_pass_w_regtw<N, N / 8, 8>(c, tw2r, tid & 63);
                        PassesW<N, N / 64>::run(c, tw, tid & 63);
                    } else {
                        PassesW<N, FUSE1 ? N / 8 : N>::run(c, tw, tid & 63);
                    }
                }
                __syncthreads();
            } else {
                Passes<N, N, Z_THREADS, B>::run(lds, CP, nb, tw);
            }
        }
        // even/odd split: X_k = E + (-i W_2N^k) O with E = (Z_k + conj Z_{N-k})/2, O = (Z_k - conj Z_{N-k})/2, k = 0..N.
        // A thread forms two adjacent outputs (one 16-B store); the row is written over its whole pitch (zeros behind
        // k = N) so that no partial 128-B line is ever written.
        if (!(dbg & 2) && !FUSE)
            for (int e = tid; e < nb * (pitch_c / 2); e += Z_THREADS) {
                const int r = e / (pitch_c / 2), k0 = (e % (pitch_c / 2)) * 2;
                const float2 *c = lds + r * CP;
                float2 X[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int k = k0 + u;
                    X[u] = make_float2(0.f, 0.f);
                    if (k <= N) {
                        const float2 zk = c[padq(k & (N - 1))];
                        const float2 zn = c[padq((N - k) & (N - 1))];
                        const float2 E = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
                        const float2 O = make_float2(0.5f * (zk.x - zn.x), 0.5f * (zk.y + zn.y));
                        const float2 w = tw2[k];                             // exp(-2 pi i k / 2N) = (cos, -sin)
                        const float2 miw = make_float2(w.y, -w.x);           // -i * w
                        X[u] = cadd(E, cmul(miw, O));
                    }
                }
                *reinterpret_cast<float4 *>(mesh + zf.dst_off + (row0 + r) * pitch_r + 2 * k0) =
                    make_float4(X[0].x, X[0].y, X[1].x, X[1].y);
            }
        if (!(dbg & 2) && FUSE) {
            const int x = (int)(tile / N), y = (int)(tile % N);
            float sy, cy, sx, cx;
            sincospif((float)y / (float)N, &sy, &cy);        // W_n^y = exp(-2 pi i y / n), n = 2N
            sincospif((float)(zf.xg0 + x) / (float)N, &sx, &cx);
            const float2 Wy = make_float2(cy, -sy), Wx = make_float2(cx, -sx);
            for (int k2 = tid; k2 < pitch_c / 2; k2 += Z_THREADS) {
                const int k0 = k2 * 2;
                float2 X[4][2];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float2 *c = lds + r * CP;
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int k = k0 + u;
                        X[r][u] = make_float2(0.f, 0.f);
                        if (k <= N) {
                            const float2 zk = c[padq(k & (N - 1))];
                            const float2 zn = c[padq((N - k) & (N - 1))];
                            const float2 E = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
                            const float2 O = make_float2(0.5f * (zk.x - zn.x), 0.5f * (zk.y + zn.y));
                            const float2 w = tw2[k];
                            X[r][u] = cadd(E, cmul(make_float2(w.y, -w.x), O));
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {   // rows: 0 = (x, y), 1 = (x, y+H), 2 = (x+H, y), 3 = (x+H, y+H)
                    const float2 A = cadd(X[0][u], X[1][u]), Bv = cmul(csub(X[0][u], X[1][u]), Wy);
                    const float2 Cv = cadd(X[2][u], X[3][u]), D = cmul(csub(X[2][u], X[3][u]), Wy);
                    X[0][u] = cadd(A, Cv);
                    X[2][u] = cmul(csub(A, Cv), Wx);
                    X[1][u] = cadd(Bv, D);
                    X[3][u] = cmul(csub(Bv, D), Wx);
                }
#pragma unroll
                for (int r = 0; r < 4; r++)
                    *reinterpret_cast<float4 *>(mesh + zf.dst_off + row_of(tile, r) * pitch_r + 2 * k0) =
                        make_float4(X[r][0].x, X[r][0].y, X[r][1].x, X[r][1].y);
            }
        }
        if (!has_next) break;
        __syncthreads();   // every LDS read of this tile is done
        wait_vmcnt_upto8((dbg & 2) || nb < B ? 0 : stores_min);
        stage();
        tile = next;
    }
}

// ---- strided pass: C adjacent columns x N elements (element stride S complex), in place ----------------------
// tile t -> (outer index o = t / ntile_c, column tile ct = t % ntile_c); first element at o*outer_stride + ct*C.
// Persistent workgroups with the next tile prefetched into registers, as above.
// PACK (y pass of an x-slab in the fused form, outer index o = 2 x + yhalf): the transformed rows are not written back in
// place but straight into the send buffer of the pencil transpose, send[p][x][yl][k] with y-row yr = yhalf N + f,
// p = yr / nyl, yl = yr % nyl - the separate pack pass (one read + one write of the slab) disappears
struct ColsPack {
    float2 *out;
    int lg_nyl, x0;
    int64_t peer_stride, x_stride;     // nxl * nyl * pitch_c, nyl * pitch_c (complex elements)
    // x pass of the fused form in front of a binning that ends at k_max: a tile whose first column already has ky^2 + kz^2
    // beyond skip_cut (in units of the fundamental, > 0 to enable) holds no mode the binning reads - it is not transformed.
    // skip_n: mesh size (the y row of outer index o % skip_n holds frequency 2 (r mod n/2) + (r div n/2), folded)
    float skip_cut = 0.f;
    int skip_n = 0;
    // y pass (outer index o = 2 x + yhalf) in front of such a binning: output rows ky whose 16 columns all have
    // ky^2 + kz^2 beyond wskip_cut are never read again - blocks of RS rows that are dead as a whole are not written back
    // (the count of the stores a thread issues stays uniform over the workgroup, which the counted vmcnt wait needs)
    float wskip_cut = 0.f;
    int64_t dst_off = 0;               // out of place: rows are written dst_off complex elements from where they were read
    // COMPACT pencil transpose (slab_layout below): a row travels with its live columns only.  Output row yr sits row_off[yr]
    // complex elements into its peer's plane block and holds row_len[yr] columns (a multiple of 16, the same for the 16 rows of
    // an aligned group: a wave's rows of one store slot lie in one group, so a skipped slot is skipped by the whole wave and
    // its hand-counted vmcnt stays exact).  Table per row: {its offset in the send buffer for plane 0 of its peer's block
    // (lo, hi), elements per plane of that block, live columns}.  Lane q of a wave loads the entry of the first row the wave
    // stores in slot q - one vector load per tile, issued BEFORE the prefetch of the next tile so that it never sits in
    // vmcnt between the stores - and the store loop takes it from there with readlane
    const uint4 *row_ent = nullptr;
};

template <int N, int C, bool F1 = true, bool PACK = false>
__global__ __launch_bounds__(FFT_THREADS) void fft_cols(float2 *__restrict__ data, int64_t S, int ntile_c,
                                                        int64_t ntiles, int64_t outer_stride, int64_t outer_mod,
                                                        int64_t outer_stride2, const float2 *__restrict__ twN, int dbg,
                                                        ColsPack pk) {
    constexpr int CP = colpitch_of<N>();
    constexpr int NLD = (N * (C / 2) + FFT_THREADS - 1) / FFT_THREADS;
    constexpr bool WHOLE = (N * (C / 2)) % FFT_THREADS == 0;
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *tw = reinterpret_cast<float2 *>(smem);
    float2 *lds = tw + N;
    const int tid = threadIdx.x;
    for (int q = tid; q < N; q += FFT_THREADS) tw[q] = twN[q];
    v4f regs[NLD];
    // tile (outer index o, column tile ct) -> o splits as (o % outer_mod) * outer_stride + (o / outer_mod) * outer_stride2.
    // Tile counts fit 32 bits: all index arithmetic is 32-bit (a 64-bit division costs ~100 VALU instructions and this
    // runs per tile in every lane), only the final element offset is 64-bit
    const int n_outer = (int)(ntiles / ntile_c);
    const bool two_level = outer_mod < (int64_t)n_outer;
    const unsigned int omod = two_level ? (unsigned int)outer_mod : 1u;
    auto tile_ptr = [&](int o, int ct) {
        unsigned int a = (unsigned int)o, b = 0;
        if (two_level) {
            b = a / omod;
            a -= b * omod;
        }
        return data + (int64_t)a * outer_stride + (int64_t)b * outer_stride2 + ct * C;
    };
    auto prefetch = [&](const float2 *g) {
        // lanes walk the C columns of one row first (C*8 B contiguous), two columns per 16-B load
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int e = q * FFT_THREADS + tid;
            const int c2 = (e % (C / 2)) * 2, y = WHOLE ? e / (C / 2) : min(e / (C / 2), N - 1);
            gload16_async(regs[q], g + (int64_t)y * S + c2);
        }
    };
    // FUSE1: load q of a thread is row tid / (C/2) + q * RS of its column pair: whole butterflies of the first radix-8 pass
    constexpr int RS = FFT_THREADS / (C / 2);
    constexpr bool FUSE1 = F1 && wave_local(N) && WHOLE && (RS == N / 8 || RS == N / 16) && NLD * RS == N;
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < NLD; q++) touch(regs[q]);
        if constexpr (FUSE1) {
            const int c2 = (tid % (C / 2)) * 2, j0 = tid / (C / 2);
            stage_pass1<N, NLD, RS>(regs, lds + c2 * CP, lds + (c2 + 1) * CP, j0, j0, tw);
            return;
        }
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int e = q * FFT_THREADS + tid;
            const int c2 = (e % (C / 2)) * 2, y = e / (C / 2);
            if (WHOLE || y < N) {
                lds[c2 * CP + padq(y)] = make_float2(regs[q].x, regs[q].y);
                lds[(c2 + 1) * CP + padq(y)] = make_float2(regs[q].z, regs[q].w);
            }
        }
    };
    // Tile order.  Workgroups b and b + 8 share an XCD (round-robin dispatch; a speed assumption only): the 8 groups
    // take whole outer indices (o = 8 * og + group), so the workgroups of one XCD walk the column tiles of the same few
    // outer indices - for the x pass that is the same set of N pages (rows 17 MB apart at n = 2048) instead of every
    // XCD touching the pages of every outer index in flight.  Falls back to the flat order when the grid or the outer
    // count does not divide.
    const bool xmap = (gridDim.x % 8 == 0) && (n_outer % 8 == 0) && !(dbg & 8);
    const int grp = xmap ? (int)(blockIdx.x & 7) : 0, ostep = xmap ? 8 : 1;
    const unsigned int qstep = xmap ? (gridDim.x >> 3) : gridDim.x, q0 = xmap ? (blockIdx.x >> 3) : blockIdx.x;
    const int n_og = n_outer / ostep;                       // outer indices of this group
    const int dg = (int)(qstep / (unsigned int)ntile_c), dc = (int)(qstep % (unsigned int)ntile_c);
    int og = (int)(q0 / (unsigned int)ntile_c), ct = (int)(q0 % (unsigned int)ntile_c);   // the q-th tile of the group
    auto dead = [&](int o, int ctile) {
        if (!(pk.skip_cut > 0.f)) return false;
        const int yr = o % pk.skip_n, hh = pk.skip_n >> 1;
        const int j = ((yr % hh) << 1) | (yr >= hh ? 1 : 0), jj = j < hh ? j : j - pk.skip_n, k0 = ctile * C;
        return (float)(jj * jj + k0 * k0) > pk.skip_cut;
    };
    auto step = [&]() {
        do {
            og += dg, ct += dc;
            if (ct >= ntile_c) ct -= ntile_c, og++;
        } while (og < n_og && dead(og * ostep + grp, ct));
    };
    if (og < n_og && dead(og * ostep + grp, ct)) step();
    if (og >= n_og) return;
    int o_cur = og * ostep + grp, ct_cur = ct;
    float2 *gcur = tile_ptr(o_cur, ct);
    prefetch(gcur);
    __syncthreads();     // the twiddle table, which a fused first pass reads while staging
    wait_vmcnt<0>();
    stage();
    for (;;) {
        __syncthreads();
        step();
        const bool has_next = og < n_og;
        const int o_next = og * ostep + grp, ct_next = ct;
        float2 *gnext = has_next ? tile_ptr(o_next, ct) : gcur;
        uint4 ent = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (PACK) {
            static_assert(NLD <= 64, "one lane per store slot");
            if (pk.row_ent && (tid & 63) < NLD) {
                const int f0 = ((tid & 63) * FFT_THREADS + (tid & ~63)) / (C / 2);
                ent = pk.row_ent[(o_cur & 1) * N + min(f0, N - 1)];
            }
        }
        if (has_next) prefetch(gnext);
        if (!(dbg & 1)) {
            if constexpr (wave_local(N)) {
#pragma unroll 1
                for (int c = tid >> 6; c < C; c += FFT_THREADS / 64) PassesW<N, FUSE1 ? N / 8 : N>::run(lds + c * CP, tw, tid & 63);
                __syncthreads();
            } else {
                Passes<N, N>::run(lds, CP, C, tw);
            }
        }
        int nstored = NLD;
        if (!(dbg & 2)) {
            float2 *g = gcur;
            const bool wskip = WHOLE && pk.wskip_cut > 0.f;
            const int kz0 = ct_cur * C, half = o_cur & 1;
            // constant trip count: the compiler can then count these stores in its vmcnt bookkeeping
#pragma unroll
            for (int q = 0; q < NLD; q++) {
                const int e = q * FFT_THREADS + tid;
                const int c2 = (e % (C / 2)) * 2, f = e / (C / 2);
                if (wskip) {   // rows [q RS, (q + 1) RS) of this half: frequencies 2 f + half, |.| unimodal - the ends decide
                    const int ja = 2 * (q * RS) + half, jb = 2 * (q * RS + RS - 1) + half;
                    const int fa = ja < pk.skip_n / 2 ? ja : pk.skip_n - ja, fb = jb < pk.skip_n / 2 ? jb : pk.skip_n - jb, m = min(fa, fb);
                    if ((float)(m * m + kz0 * kz0) > pk.wskip_cut) {
                        nstored--;
                        continue;
                    }
                }
                uint2 pl0 = make_uint2(0u, 0u);
                int64_t base0 = 0;
                int f0 = 0;
                if constexpr (PACK) {
                    if (pk.row_ent) {   // compact transpose: this wave's rows of the slot (one aligned group) end before this tile
                        f0 = __builtin_amdgcn_readfirstlane((q * FFT_THREADS + (tid & ~63)) / (C / 2));
                        pl0 = make_uint2((unsigned int)__builtin_amdgcn_readlane((int)ent.z, q), (unsigned int)__builtin_amdgcn_readlane((int)ent.w, q));
                        base0 = (int64_t)(((uint64_t)(unsigned int)__builtin_amdgcn_readlane((int)ent.y, q) << 32) |
                                          (unsigned int)__builtin_amdgcn_readlane((int)ent.x, q));
                        if (ct_cur * C >= (int)pl0.y) {
                            nstored--;
                            continue;
                        }
                    }
                }
                if (WHOLE || f < N) {
                    const int p = padq(wave_local(N) ? f : revpos<N>(f));
                    const float2 a = lds[c2 * CP + p], b = lds[(c2 + 1) * CP + p];
                    float2 *dst = g + pk.dst_off + (int64_t)f * S + c2;
                    if constexpr (PACK) {
                        const int yr = (o_cur & 1) * N + f;
                        if (pk.row_ent)
                            dst = pk.out + base0 + (int64_t)(pk.x0 + (o_cur >> 1)) * (int64_t)pl0.x + (f - f0) * (int)pl0.y + ct_cur * C + c2;   // (the rows of a group lie back to back)
                        else
                            dst = pk.out + (int64_t)(yr >> pk.lg_nyl) * pk.peer_stride + (int64_t)(pk.x0 + (o_cur >> 1)) * pk.x_stride +
                                  (int64_t)(yr & ((1 << pk.lg_nyl) - 1)) * S + ct_cur * C + c2;
                    }
                    *reinterpret_cast<float4 *>(dst) = make_float4(a.x, a.y, b.x, b.y);
                }
            }
        }
        if (!has_next) break;
        __syncthreads();
        // the NLD stores above were issued after the prefetch loads: vmcnt(NLD) = all loads landed, stores in flight
        if (dbg & 2) wait_vmcnt<0>();
        else if (nstored != NLD && NLD <= 16) wait_vmcnt_upto16(__builtin_amdgcn_readfirstlane(nstored));   // (the same in every lane of a wave)
        else wait_vmcnt<(NLD < 60 ? NLD : 0)>();
        stage();
        gcur = gnext, o_cur = o_next, ct_cur = ct_next;
    }
}

int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    return n;
}

struct Tables {
    DevBuf twN, twHalf, tw2;   // exp(-2 pi i m / n), exp(-2 pi i m / (n/2)), exp(-2 pi i k / n) for k <= n/2
};
std::map<int, Tables> g_tables;
DevBuf g_scratch;      // second mesh of the out-of-place z / y passes (fft3d_fused)

int get_tables(int n, Tables **out) {
    auto it = g_tables.find(n);
    if (it == g_tables.end()) {
        Tables t;
        auto fill = [&](DevBuf &buf, int len, int count) -> int {
            std::vector<float2> h((size_t)count);
            for (int m = 0; m < count; m++) {
                const double th = -2.0 * M_PI * (double)m / (double)len;
                h[m] = make_float2((float)cos(th), (float)sin(th));
            }
            ABACUS_TRY(buf.reserve(h.size() * sizeof(float2)));
            HIP_TRY(hipMemcpyAsync(buf.p, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice, stream()));
            HIP_TRY(hipStreamSynchronize(stream()));
            return 0;
        };
        ABACUS_TRY(fill(t.twN, n, n));
        ABACUS_TRY(fill(t.twHalf, n / 2, n / 2));
        ABACUS_TRY(fill(t.tw2, n, n / 2 + 1));
        it = g_tables.emplace(n, t).first;
    }
    *out = &it->second;
    return 0;
}

template <int N, int B, int FUSE, int F1>
int launch_z1(float *mesh, int64_t nrows, int pitch_r, Tables *t, ZFold zf) {
    const size_t lds = (size_t)(2 * N + 2 + B * colpitch_of<N>()) * sizeof(float2);
    auto kern = fft_z_r2c<N, B, FUSE, F1>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t ntiles = FUSE ? (int64_t)zf.npair * N : ceil_div(nrows, B);
    int per_cu = 1;   // persistent grid = exactly the resident workgroups (a larger grid would run in two uneven waves)
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, Z_THREADS, lds));
    const unsigned int grid = (unsigned int)std::min<int64_t>(ntiles, (int64_t)num_cus() * std::max(per_cu, 1));
    ABACUS_LAUNCH("fft_z_r2c", kern, dim3(grid), dim3(Z_THREADS), lds, mesh, nrows, pitch_r, t->twHalf.as<float2>(),
                  t->tw2.as<float2>(), option("dbg_fft"), zf);
    return 0;
}
// (F1 variants of the z pass - first radix-8 pass fused with the staging, twiddles as lane constants - were measured equal at
// N = 1024 in round 3 and are no longer instantiated)
template <int N, int B, int FUSE = 0>
int launch_z(float *mesh, int64_t nrows, int pitch_r, Tables *t, ZFold zf = ZFold{N, N, 0}) {
    return launch_z1<N, B, FUSE, 0>(mesh, nrows, pitch_r, t, zf);
}

template <int N, int C, bool F1, bool PACK = false>
int launch_cols1(const char *name, float2 *data, int64_t S, int ntile_c, int64_t outer, int64_t outer_stride, const float2 *tw,
                 int64_t outer_mod, int64_t outer_stride2, ColsPack pk = ColsPack()) {
    const size_t lds = (size_t)(N + C * colpitch_of<N>()) * sizeof(float2);
    auto kern = fft_cols<N, C, F1, PACK>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t ntiles = outer * ntile_c;
    int per_cu = 1;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, FFT_THREADS, lds));
    const unsigned int grid = (unsigned int)std::min<int64_t>(ntiles, (int64_t)num_cus() * std::max(per_cu, 1));
    ABACUS_LAUNCH(name, kern, dim3(grid), dim3(FFT_THREADS), lds, data, S, ntile_c, ntiles, outer_stride, outer_mod,
                  outer_stride2, tw, option("dbg_fft"), pk);
    return 0;
}
template <int N, int C>
int launch_cols(const char *name, float2 *data, int64_t S, int ntile_c, int64_t outer, int64_t outer_stride, const float2 *tw,
                int64_t outer_mod = (int64_t)1 << 40, int64_t outer_stride2 = 0, ColsPack pk = ColsPack()) {
    return launch_cols1<N, C, false>(name, data, S, ntile_c, outer, outer_stride, tw, outer_mod, outer_stride2, pk);
}

template <int N, int C, int BZ>
int fft3d(float *mesh, int pitch_r, Tables *t, int64_t nx_local) {
    // nx_local x N x N real mesh (nx_local = N for the single-GPU transform)
    const int pitch_c = pitch_r / 2, kzlen = N / 2 + 1;
    ABACUS_TRY((launch_z<N / 2, BZ>(mesh, nx_local * N, pitch_r, t)));
    float2 *data = reinterpret_cast<float2 *>(mesh);
    // y: for every x-plane, columns along y (element stride pitch_c)
    const int ntile_c = (kzlen + C - 1) / C;   // the last tile reads into the row padding (pitch_c >= ntile_c*C)
    if (ntile_c * C > pitch_c) return fail("fft: row pitch too small for the column tiles");
    return launch_cols<N, C>("fft_cols_y", data, pitch_c, ntile_c, nx_local, (int64_t)N * pitch_c, t->twN.as<float2>());
}

template <int N, int C>
int fft_x(float2 *data, int pitch_c, Tables *t, int64_t ny_local, int64_t x_stride, int64_t y_stride) {
    // x: for every y, columns along x (element stride x_stride)
    const int ntile_c = (N / 2 + 1 + C - 1) / C;
    if (ntile_c * C > pitch_c) return fail("fft: row pitch too small for the column tiles");
    return launch_cols<N, C>("fft_cols_x", data, x_stride, ntile_c, ny_local, y_stride, t->twN.as<float2>());
}

}  // namespace

namespace abacus {

bool gfft_supported(int n, int is_double);
int gfft_r2c_inplace_f32(float *mesh, int n, int pitch_r, float xcut);
int gfft_release();

// the tuned power-of-two kernels of this file (also what the slab-decomposed transform needs)
bool fft_native_pow2(int n) { return n >= 64 && n <= 2048 && (n & (n - 1)) == 0; }
// ... or the mixed-radix kernels of gfft.hip (even sizes with factors 2, 3, 5, 7, 11, 13: 72, 96, 384, 550, 768, 1536 ...).
// Per 3-D transform against hipFFT (profiles/r04/gfft_vs_hipfft.txt): 384^3 0.53 vs 0.58 ms, 550^3 1.51 vs 1.90, 768^3 3.5 vs 4.6,
// 1152^3 14.4 vs 16.9, 1536^3 30.4 vs 38.8.  Sizes neither family covers (odd, or a prime factor above 13) go to hipFFT.
bool fft_native_supported(int n) { return fft_native_pow2(n) || gfft_supported(n, 0); }

// z and y passes over `nx_local` consecutive x-planes (the part of the transform that is local to an x-slab)
int fft_native_zy(float *mesh, int n, int pitch_r, int64_t nx_local) {
    Tables *t;
    ABACUS_TRY(get_tables(n, &t));
    switch (n) {
        case 64: return fft3d<64, 16, 16>(mesh, pitch_r, t, nx_local);
        case 128: return fft3d<128, 16, 16>(mesh, pitch_r, t, nx_local);
        case 256: return fft3d<256, 16, 8>(mesh, pitch_r, t, nx_local);
        case 512: return fft3d<512, 16, 8>(mesh, pitch_r, t, nx_local);
        case 1024: return fft3d<1024, 16, 4>(mesh, pitch_r, t, nx_local);
        case 2048: return fft3d<2048, 8, 4>(mesh, pitch_r, t, nx_local);
    }
    return fail("fft: unsupported size %d", n);
}

// x pass over `ny_local` rows of y: element (x, y, k) at data[x*x_stride + y*y_stride + k]
int fft_native_x(float *mesh, int n, int pitch_r, int64_t ny_local, int64_t x_stride, int64_t y_stride) {
    Tables *t;
    ABACUS_TRY(get_tables(n, &t));
    float2 *data = reinterpret_cast<float2 *>(mesh);
    const int pitch_c = pitch_r / 2;
    switch (n) {
        case 64: return fft_x<64, 16>(data, pitch_c, t, ny_local, x_stride, y_stride);
        case 128: return fft_x<128, 16>(data, pitch_c, t, ny_local, x_stride, y_stride);
        case 256: return fft_x<256, 16>(data, pitch_c, t, ny_local, x_stride, y_stride);
        case 512: return fft_x<512, 16>(data, pitch_c, t, ny_local, x_stride, y_stride);
        case 1024: return fft_x<1024, 16>(data, pitch_c, t, ny_local, x_stride, y_stride);
        case 2048: return fft_x<2048, 8>(data, pitch_c, t, ny_local, x_stride, y_stride);
    }
    return fail("fft: unsupported size %d", n);
}

// xcut > 0 (mixed-radix sizes): see fft_native_r2c_fused
int fft_native_r2c_inplace(float *mesh, int n, int pitch_r, float xcut) {
    if (!fft_native_pow2(n)) return gfft_r2c_inplace_f32(mesh, n, pitch_r, xcut);
    ABACUS_TRY(fft_native_zy(mesh, n, pitch_r, n));
    return fft_native_x(mesh, n, pitch_r, n, (int64_t)n * (pitch_r / 2), pitch_r / 2);
}

const float2 *fft_twiddles(int n) {
    Tables *t;
    if (get_tables(n, &t) != 0) return nullptr;
    return t->twN.as<float2>();
}
int fft_num_cus() { return num_cus(); }

// Full-mesh transform with the first radix-2 stage of y and x fused into the z pass (see fft_z_r2c<.., FUSE>): the y and
// x passes are n/2-point column transforms with C columns.  Output rows are in the permuted order
// f = 2 (r mod n/2) + (r div n/2) along x and y (power.hip's binning undoes it in its index arithmetic).
template <int N, int C>
int fft3d_fused(float *mesh, int pitch_r, Tables *t, Tables *th, bool with_x, float xcut = 0.f) {
    constexpr int H = N / 2;
    const int pitch_c = pitch_r / 2, kzlen = N / 2 + 1;
    const int ntile_c = (kzlen + C - 1) / C;
    if (ntile_c * C > pitch_c) return fail("fft: row pitch too small for the column tiles");
    // ping-pong through a second buffer when there is room for one: z pass mesh -> scratch, y pass scratch -> mesh (the x
    // pass and everything behind it find the spectrum where the in-place form leaves it).  Rows the y pass does not write
    // (beyond k_max) keep the deposit's values in `mesh`: finite, and never read.
    // Measured: 1024^3 z 2.13 -> 1.90 ms, step 10.27 -> 10.11; 2048^3 z 15.15 -> 15.12, step 49.7 -> 49.6 for 35 GB more -
    // the 2048^3 z pass is held by its LDS and vector-ALU work as much as by the memory system - so only 1024^3 takes it.
    float *scratch = nullptr;
    if (N == 1024 && !option("fft_inplace")) {
        const size_t bytes = (size_t)N * N * pitch_r * sizeof(float);
        if (g_scratch.cap >= bytes || g_scratch.reserve(bytes) == 0) scratch = g_scratch.as<float>();
        else (void)hipGetLastError();                                  // no room: in place
    }
    ZFold zf{N / 2, N / 2, 0};
    if (scratch) zf.dst_off = scratch - mesh;
    ABACUS_TRY((launch_z<N / 2, 4, 1>(mesh, (int64_t)N * N, pitch_r, t, zf)));
    float2 *data = reinterpret_cast<float2 *>(mesh);
    // y: 2 N half-planes of H rows each, contiguous in memory
    ColsPack py;
    py.out = nullptr, py.lg_nyl = 0, py.x0 = 0, py.peer_stride = 0, py.x_stride = 0;
    py.wskip_cut = option("dbg_fft") & 16 ? 0.f : xcut, py.skip_n = N;
    if (scratch) py.dst_off = data - reinterpret_cast<float2 *>(scratch);
    ABACUS_TRY((launch_cols<H, C>("fft_cols_y", scratch ? reinterpret_cast<float2 *>(scratch) : data, pitch_c, ntile_c, 2 * (int64_t)N,
                                  (int64_t)H * pitch_c, th->twN.as<float2>(), (int64_t)1 << 40, 0, py)));
    if (!with_x) return 0;   // the caller runs the last pass fused with the binning (xbin.hip)
    // x: for every y and either half of x, H planes apart by N * pitch_c
    const int64_t S = (int64_t)N * pitch_c;
    ColsPack pk;
    pk.out = nullptr, pk.lg_nyl = 0, pk.x0 = 0, pk.peer_stride = 0, pk.x_stride = 0;
    pk.skip_cut = option("dbg_fft") & 16 ? 0.f : xcut, pk.skip_n = N;
    return launch_cols<H, C>("fft_cols_x", data, S, ntile_c, 2 * (int64_t)N, pitch_c, th->twN.as<float2>(), N, (int64_t)H * S, pk);
}

// ---- compact pencil transpose ---------------------------------------------------------------------------------------------
// When the spectrum goes to a binning that ends at |k|^2 = cut (fundamental units) and nowhere else, the columns of a y row beyond
// sqrt(cut - ky^2) are never read: 21 % of the half-spectrum when the bins end at the Nyquist frequency.  They need not cross the
// links.  A row of the transposed layout keeps its first row_len columns - a multiple of 16, decided for aligned groups of 16
// rows from the smallest |ky| of the group (a wave of the y pass stores 8 or 16 rows of one group at a time) - and the rows of
// a peer's plane block are packed back to back.  Sender (fft_cols<PACK>, ColsPack) and receiver (fft_x_bin2, XBinGeom) take the
// same tables, a function of (n, ranks, cut) alone.
struct SlabLayout {
    int n = 0, W = 0;
    float cut = 0.f;
    std::vector<unsigned int> row_off;      // (n) offset of permuted row yr inside its peer's plane block, complex elements
    std::vector<unsigned short> row_len;    // (n)
    std::vector<int64_t> P;                 // (W) elements per plane of the block that goes to peer p
    DevBuf d_off;                           // row_off (receiver: fft_x_bin2)
    std::map<int, DevBuf> d_ent;            // sender table per h (planes of a rank / 2): ColsPack::row_ent
};
static std::vector<SlabLayout *> g_slab_layouts;

const SlabLayout *slab_layout(int n, int W, float cut, int pitch_c) {
    if (!(cut > 0.f) || W < 1 || W > 16 || n % W || (n / W) % 16 || (n != 256 && n != 1024 && n != 2048)) return nullptr;
    for (const SlabLayout *l : g_slab_layouts)
        if (l->n == n && l->W == W && l->cut == cut) return l;
    SlabLayout *l = new SlabLayout();
    l->n = n, l->W = W, l->cut = cut;
    l->row_off.resize((size_t)n), l->row_len.resize((size_t)n), l->P.assign((size_t)W, 0);
    const int H = n / 2, nyl = n / W, ntile = (n / 2 + 1 + 15) / 16;
    if (ntile * 16 > pitch_c) {
        delete l;
        return nullptr;
    }
    for (int g0 = 0; g0 < n; g0 += 16) {
        int m = n;
        for (int yr = g0; yr < g0 + 16; yr++) {          // permuted row yr holds frequency 2 (yr mod n/2) + (yr div n/2)
            const int j = 2 * (yr % H) + yr / H;
            m = std::min(m, j < H ? j : n - j);
        }
        int live = 0;
        for (int ct = 0; ct < ntile; ct++) live += (float)(m * m + (ct * 16) * (ct * 16)) > cut ? 0 : 1;   // the binning's own test
        for (int yr = g0; yr < g0 + 16; yr++) {
            const int p = yr / nyl;
            l->row_len[(size_t)yr] = (unsigned short)(live * 16);
            l->row_off[(size_t)yr] = (unsigned int)l->P[(size_t)p];
            l->P[(size_t)p] += live * 16;
        }
    }
    if (l->d_off.reserve((size_t)n * 4) != 0 ||
        hipMemcpyAsync(l->d_off.p, l->row_off.data(), (size_t)n * 4, hipMemcpyHostToDevice, stream()) != hipSuccess ||
        hipStreamSynchronize(stream()) != hipSuccess) {
        delete l;
        return nullptr;
    }
    g_slab_layouts.push_back(l);
    return l;
}
// the sender's tables for blocks of 2 h planes per peer
static int slab_layout_sender(SlabLayout *l, int h, const uint4 **row_ent) {
    auto it = l->d_ent.find(h);
    if (it == l->d_ent.end()) {
        const int n = l->n, nyl = n / l->W;
        std::vector<int64_t> poff((size_t)l->W);
        std::vector<uint4> ent((size_t)n);
        int64_t off = 0;
        for (int p = 0; p < l->W; p++) poff[(size_t)p] = off, off += 2 * (int64_t)h * l->P[(size_t)p];
        for (int yr = 0; yr < n; yr++) {
            const int p = yr / nyl;
            const uint64_t base = (uint64_t)(poff[(size_t)p] + l->row_off[(size_t)yr]);
            ent[(size_t)yr] = make_uint4((unsigned int)base, (unsigned int)(base >> 32), (unsigned int)l->P[(size_t)p], l->row_len[(size_t)yr]);
        }
        DevBuf b;
        ABACUS_TRY(b.reserve((size_t)n * 16));
        HIP_TRY(hipMemcpyAsync(b.p, ent.data(), (size_t)n * 16, hipMemcpyHostToDevice, stream()));
        HIP_TRY(hipStreamSynchronize(stream()));
        it = l->d_ent.emplace(h, b).first;
    }
    *row_ent = it->second.as<uint4>();
    return 0;
}
int slab_layout_query(int n, int W, float cut, int pitch_c, int64_t *P_out, const unsigned int **row_off_dev) {
    const SlabLayout *l = slab_layout(n, W, cut, pitch_c);
    if (!l) return 1;
    if (P_out)
        for (int p = 0; p < W; p++) P_out[p] = l->P[(size_t)p];
    if (row_off_dev) *row_off_dev = l->d_off.as<unsigned int>();
    return 0;
}

// The fused form on the folded slabs of a multi-GPU mesh (analysis/slab_power.py): a rank owns h plane pairs (x, x + n/2),
// the first half at `mesh`, the second xsep planes behind it, so the z pass fuses the first radix-2 stage of y AND x exactly
// as on the whole mesh (ZFold) and everything behind it is the single-GPU form: the y pass as two n/2-point transforms per
// plane with C columns, the x pass - behind the pencil transpose - two n/2-point transforms over the sum rows and the
// difference rows (fft_native_fused_x_slab, or the fused last pass + binning of xbin.hip).  Works on pairs [p0, p0 + pc).
// pack_out != nullptr: the y pass writes the send buffer of the pencil transpose (ColsPack), send[peer][s h + p][yl][k]
template <int N, int C>
int fft3d_fused_zy_slab(float *mesh, int pitch_r, Tables *t, Tables *th, int h, int64_t xsep, int xg0, int p0, int pc,
                        float *pack_out, int world, const SlabLayout *lay) {
    constexpr int H = N / 2;
    const int pitch_c = pitch_r / 2, ntile_c = (N / 2 + 1 + C - 1) / C;
    if (ntile_c * C > pitch_c) return fail("fft: row pitch too small for the column tiles");
    const int64_t plane = (int64_t)N * pitch_r;
    ABACUS_TRY((launch_z<N / 2, 4, 1>(mesh + p0 * plane, 0, pitch_r, t, ZFold{pc, (int)xsep, xg0 + p0})));
    ColsPack pk;
    if (pack_out) {
        const int nyl = N / world;
        pk.out = reinterpret_cast<float2 *>(pack_out);
        pk.lg_nyl = 0;
        while ((1 << pk.lg_nyl) < nyl) pk.lg_nyl++;
        if ((1 << pk.lg_nyl) != nyl || nyl > H) return fail("fft: packed y pass needs a power-of-two number of ranks >= 2 (nyl %d)", nyl);
        pk.x_stride = (int64_t)nyl * pitch_c;
        pk.peer_stride = 2 * (int64_t)h * nyl * pitch_c;
        if (lay) ABACUS_TRY(slab_layout_sender(const_cast<SlabLayout *>(lay), h, &pk.row_ent));   // compact transpose
    }
    for (int s = 0; s < 2; s++) {
        float2 *data = reinterpret_cast<float2 *>(mesh + (s * xsep + p0) * plane);
        if (!pack_out) {
            if (xsep == pc && p0 == 0) {     // the halves are contiguous (one rank, whole mesh): one launch
                return launch_cols<H, C>("fft_cols_y", data, pitch_c, ntile_c, 4 * (int64_t)pc, (int64_t)H * pitch_c, th->twN.as<float2>());
            }
            ABACUS_TRY((launch_cols<H, C>("fft_cols_y", data, pitch_c, ntile_c, 2 * (int64_t)pc, (int64_t)H * pitch_c, th->twN.as<float2>())));
        } else {
            pk.x0 = s * h + p0;
            ABACUS_TRY((launch_cols1<H, C, false, true>("fft_cols_y", data, pitch_c, ntile_c, 2 * (int64_t)pc, (int64_t)H * pitch_c,
                                                        th->twN.as<float2>(), (int64_t)1 << 40, 0, pk)));
        }
    }
    return 0;
}
template <int N, int C>
int fft3d_fused_x_slab(float *mesh, int pitch_r, Tables *th, int64_t ny_local) {
    constexpr int H = N / 2;
    const int pitch_c = pitch_r / 2, ntile_c = (N / 2 + 1 + C - 1) / C;
    // layout (y_local, x, k): the two halves of x of one y row are contiguous blocks of H rows
    return launch_cols<H, C>("fft_cols_x", reinterpret_cast<float2 *>(mesh), pitch_c, ntile_c, 2 * ny_local, (int64_t)H * pitch_c,
                             th->twN.as<float2>());
}
int fft_native_fused_zy_slab(float *mesh, int n, int pitch_r, int h, int64_t xsep, int xg0, int p0, int pc, float *pack_out,
                             int world, float cut) {
    Tables *t, *th;
    ABACUS_TRY(get_tables(n, &t));
    ABACUS_TRY(get_tables(n / 2, &th));
    if (h < 1 || pc < 1 || p0 < 0 || p0 + pc > h || xsep < h) return fail("fft: plane pairs [%d, +%d) of %d", p0, pc, h);
    const SlabLayout *lay = nullptr;
    if (cut > 0.f) {
        if (!pack_out) return fail("fft: the compact transpose needs the packed y pass");
        lay = slab_layout(n, world, cut, pitch_r / 2);
        if (!lay) return fail("fft: no compact transpose layout for a mesh of %d over %d ranks", n, world);
    }
    switch (n) {
        case 256: return fft3d_fused_zy_slab<256, 16>(mesh, pitch_r, t, th, h, xsep, xg0, p0, pc, pack_out, world, lay);
        case 1024: return fft3d_fused_zy_slab<1024, 16>(mesh, pitch_r, t, th, h, xsep, xg0, p0, pc, pack_out, world, lay);
        case 2048: return fft3d_fused_zy_slab<2048, 16>(mesh, pitch_r, t, th, h, xsep, xg0, p0, pc, pack_out, world, lay);
    }
    return fail("fft: the fused transform supports n = 1024 and 2048");
}
int fft_native_fused_x_slab(float *mesh, int n, int pitch_r, int64_t ny_local) {
    Tables *th;
    ABACUS_TRY(get_tables(n / 2, &th));
    switch (n) {
        case 256: return fft3d_fused_x_slab<256, 16>(mesh, pitch_r, th, ny_local);
        case 1024: return fft3d_fused_x_slab<1024, 16>(mesh, pitch_r, th, ny_local);
        case 2048: return fft3d_fused_x_slab<2048, 16>(mesh, pitch_r, th, ny_local);
    }
    return fail("fft: the fused transform supports n = 1024 and 2048");
}

// n = 256 only on request (tests against the CPU oracle): small meshes gain nothing from the fused form
int fft_native_fused_supported(int n) { return n == 2048 || n == 1024 || (n == 256 && option("fft_fuse_small")); }

static int fused_impl(float *mesh, int n, int pitch_r, bool with_x, float xcut = 0.f) {
    Tables *t, *th;
    ABACUS_TRY(get_tables(n, &t));
    ABACUS_TRY(get_tables(n / 2, &th));
    switch (n) {
        case 256: return fft3d_fused<256, 16>(mesh, pitch_r, t, th, with_x, xcut);
        case 1024: return fft3d_fused<1024, 16>(mesh, pitch_r, t, th, with_x, xcut);
        case 2048: return fft3d_fused<2048, 16>(mesh, pitch_r, t, th, with_x, xcut);
    }
    return fail("fft: the fused transform supports n = 1024 and 2048");
}
// xcut > 0: the caller bins the spectrum up to |k|^2 = xcut (fundamental units) and reads nothing beyond - the x pass leaves the
// column tiles that lie entirely beyond untransformed (their content is then NOT the spectrum)
int fft_native_r2c_fused(float *mesh, int n, int pitch_r, float xcut) { return fused_impl(mesh, n, pitch_r, true, xcut); }
// z and y passes only: the x pass is left to fft_x_bin_run (last pass fused with the binning)
int fft_native_r2c_fused_zy(float *mesh, int n, int pitch_r, float xcut) { return fused_impl(mesh, n, pitch_r, false, xcut); }

int fft_trim_scratch() { return g_scratch.release(); }

int fft_native_release() {
    ABACUS_TRY(g_scratch.release());
    for (SlabLayout *l : g_slab_layouts) {
        ABACUS_TRY(l->d_off.release());
        for (auto &kv : l->d_ent) ABACUS_TRY(kv.second.release());
        delete l;
    }
    g_slab_layouts.clear();
    for (auto &kv : g_tables) {
        ABACUS_TRY(kv.second.twN.release());
        ABACUS_TRY(kv.second.twHalf.release());
        ABACUS_TRY(kv.second.tw2.release());
    }
    g_tables.clear();
    return gfft_release();
}

}  // namespace abacus
