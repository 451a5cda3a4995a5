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

constexpr int FFT_THREADS = 512;   // column passes
constexpr int Z_THREADS = 256;     // z pass: small tiles (4 rows), several workgroups per CU
constexpr int PADSHIFT = 4;   // one pad element per 16: de-conflicts the stride-R accesses of the late passes

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ int padq(int q) { return q + (q >> PADSHIFT); }

// Asynchronous 16-B global load the compiler does not track: the prefetch of the next tile stays in flight across the
// transform AND the write-back of the current one.  hipcc's own vmcnt bookkeeping cannot express "wait for the loads
// but not for the stores issued after them" once loop paths merge (it degenerates to waiting for every store, which
// exposes the store latency once per tile), so the wait is written by hand: vmcnt retires in issue order on gfx9, so
// after `prefetch; ...; K stores` a vmcnt(K) guarantees that every prefetch load has landed.
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gload16_async(v4f &dst, const void *p) {
    v4f t;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(p) : "memory");
    dst = t;
}
template <int K>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory");
}
__device__ __forceinline__ void wait_vmcnt_upto8(int k) {   // runtime count (uniform), immediate operand
    switch (k) {
        case 1: wait_vmcnt<1>(); break;
        case 2: wait_vmcnt<2>(); break;
        case 3: wait_vmcnt<3>(); break;
        case 4: wait_vmcnt<4>(); break;
        case 5: wait_vmcnt<5>(); break;
        case 6: wait_vmcnt<6>(); break;
        case 7: wait_vmcnt<7>(); break;
        case 8: wait_vmcnt<8>(); break;
        default: wait_vmcnt<0>(); break;
    }
}
// ties registers to the wait above: their uses cannot be scheduled before it
__device__ __forceinline__ void touch(v4f &a) {
    v4f t = a;
    asm volatile("" : "+v"(t));
    a = t;
}

// forward DFTs of size R in registers, natural order in and out (W = exp(-2 pi i / R))
template <int R>
__device__ __forceinline__ void dft(float2 (&a)[R]);
template <>
__device__ __forceinline__ void dft<2>(float2 (&a)[2]) {
    const float2 t = a[0];
    a[0] = cadd(t, a[1]);
    a[1] = csub(t, a[1]);
}
__device__ __forceinline__ void dft4(float2 &a0, float2 &a1, float2 &a2, float2 &a3) {
    const float2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = csub(a1, a3);
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = make_float2(t1.x + t3.y, t1.y - t3.x);   // t1 - i t3
    a3 = make_float2(t1.x - t3.y, t1.y + t3.x);   // t1 + i t3
}
template <>
__device__ __forceinline__ void dft<4>(float2 (&a)[4]) { dft4(a[0], a[1], a[2], a[3]); }
template <>
__device__ __forceinline__ void dft<8>(float2 (&a)[8]) {
    dft4(a[0], a[2], a[4], a[6]);   // even
    dft4(a[1], a[3], a[5], a[7]);   // odd
    const float h = 0.70710678118654752440f;
    const float2 o0 = a[1];
    const float2 o1 = make_float2(h * (a[3].x + a[3].y), h * (a[3].y - a[3].x));   // * (h, -h)
    const float2 o2 = make_float2(a[5].y, -a[5].x);                                // * (-i)
    const float2 o3 = make_float2(h * (a[7].y - a[7].x), -h * (a[7].x + a[7].y));  // * (-h, -h)
    const float2 e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6];
    a[0] = cadd(e0, o0);
    a[4] = csub(e0, o0);
    a[1] = cadd(e1, o1);
    a[5] = csub(e1, o1);
    a[2] = cadd(e2, o2);
    a[6] = csub(e2, o2);
    a[3] = cadd(e3, o3);
    a[7] = csub(e3, o3);
}

constexpr int radix_of(int L) { return L >= 8 ? 8 : L; }   // greedy radix-8, then one radix-4 or radix-2 pass

// position of frequency f after the in-place DIF passes (mixed-radix digit reversal)
template <int N>
__device__ __forceinline__ int revpos(int f) {
    int pos = 0, L = N;
#pragma unroll
    for (int it = 0; it < 12; it++) {
        if (L == 1) break;
        const int R = radix_of(L);
        pos += (f % R) * (L / R);
        f /= R;
        L /= R;
    }
    return pos;
}

// padded position of element base + r * LR: when LR is a multiple of the padding period the pad term is additive, so the
// R addresses of a butterfly are one computed address plus compile-time offsets (immediate offsets of the LDS instructions)
template <int LR>
__device__ __forceinline__ int padq_strided(int base, int r) {
    if constexpr (LR % (1 << PADSHIFT) == 0) return padq(base) + r * (LR + (LR >> PADSHIFT));
    else return padq(base + r * LR);
}

// one DIF pass of sub-length L over `ncol` columns of N elements each (column c at lds + c*colpitch, padded index)
template <int N, int L, int R, int NT = FFT_THREADS>
__device__ __forceinline__ void dif_pass(float2 *lds, int colpitch, int ncol, const float2 *tw) {
    constexpr int BPC = N / R;      // butterflies per column
    constexpr int LR = L / R;
    const int total = ncol * BPC;
    for (int b = threadIdx.x; b < total; b += NT) {
        const int col = b / BPC, t = b % BPC;
        const int blk = t / LR, j = t % LR;
        float2 *c = lds + col * colpitch;
        const int base = blk * L + j;
        float2 a[R];
#pragma unroll
        for (int r = 0; r < R; r++) a[r] = c[padq_strided<LR>(base, r)];
        dft<R>(a);
        if (L > R) {   // the last pass has j = 0: all twiddles are one
#pragma unroll
            for (int r = 1; r < R; r++) a[r] = cmul(a[r], tw[j * r * (N / L)]);
        }
#pragma unroll
        for (int r = 0; r < R; r++) c[padq_strided<LR>(base, r)] = a[r];
    }
    __syncthreads();
}

// frequency held at position `pos` after the in-place DIF passes (inverse of revpos)
template <int N>
__device__ __forceinline__ int freq_of_pos(int pos) {
    int f = 0, L = N, w = 1;
#pragma unroll
    for (int it = 0; it < 12; it++) {
        if (L == 1) break;
        const int R = radix_of(L);
        const int d = pos / (L / R);
        pos -= d * (L / R);
        f += d * w;
        w *= R;
        L /= R;
    }
    return f;
}

// The last pass (L == R, no twiddles), writing NATURAL order: every butterfly of the workgroup is loaded and transformed
// into registers first, then - behind a barrier, the writes land on other threads' inputs - element f goes to position
// f.  The write-backs then read consecutive positions: conflict-free, where digit-reversed reads of consecutive
// frequencies land 4 (z pass) or 2 (column passes) lanes on every bank pair.
template <int N, int R, int NT, int MAXCOL>
__device__ __forceinline__ void dif_last_natural(float2 *lds, int colpitch, int ncol) {
    constexpr int BPC = N / R;
    constexpr int MAXIT = (MAXCOL * BPC + NT - 1) / NT;
    const int total = ncol * BPC;
    float2 a[MAXIT][R];
#pragma unroll
    for (int it = 0; it < MAXIT; it++) {
        const int b = it * NT + threadIdx.x;
        if (b < total) {
            const int col = b / BPC, t = b % BPC;
            const float2 *c = lds + col * colpitch;
#pragma unroll
            for (int r = 0; r < R; r++) a[it][r] = c[padq(t * R + r)];
            dft<R>(a[it]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < MAXIT; it++) {
        const int b = it * NT + threadIdx.x;
        if (b < total) {
            const int col = b / BPC, t = b % BPC;
            float2 *c = lds + col * colpitch;
            const int f0 = freq_of_pos<N>(t * R);          // position t*R + r holds frequency f0 + r * (N / R)
#pragma unroll
            for (int r = 0; r < R; r++) c[padq(f0 + r * (N / R))] = a[it][r];
        }
    }
    __syncthreads();
}

// MAXCOL > 0: the last pass leaves natural order (dif_last_natural, MAXCOL = most columns a tile can hold);
// MAXCOL = 0: every pass in place, frequency f at revpos(f) - the column passes at N = 1024 x 16 columns would need
// 32 more complex registers per thread next to the 64 prefetch registers and spill.
template <int N, int L, int NT = FFT_THREADS, int MAXCOL = 0>
struct Passes {
    static __device__ __forceinline__ void run(float2 *lds, int colpitch, int ncol, const float2 *tw) {
        constexpr int R = radix_of(L);
        if constexpr (L == R && MAXCOL > 0) {
            dif_last_natural<N, R, NT, MAXCOL>(lds, colpitch, ncol);
        } else {
            dif_pass<N, L, R, NT>(lds, colpitch, ncol, tw);
            if constexpr (L > R) Passes<N, L / R, NT, MAXCOL>::run(lds, colpitch, ncol, tw);
        }
    }
};

// ---- wave-local transforms: one 64-lane wave owns a whole column (N >= 512: at least 64 butterflies per pass) --------
// A column's passes then need no workgroup barrier at all - LDS executes one wave's instructions in order, so a wave-level
// fence between the passes is enough - and the waves of a workgroup drift apart, hiding each other's LDS latency.  The
// last pass holds only N/64 complex values per lane, so it can always leave natural order.
template <int N, int L, int R>
__device__ __forceinline__ void butterfly_w(float2 *c, const float2 *tw, int t) {
    constexpr int LR = L / R;
    const int blk = t / LR, j = t % LR;
    const int base = blk * L + j;
    float2 a[R];
#pragma unroll
    for (int r = 0; r < R; r++) a[r] = c[padq_strided<LR>(base, r)];
    dft<R>(a);
#pragma unroll
    for (int r = 1; r < R; r++) a[r] = cmul(a[r], tw[j * r * (N / L)]);
#pragma unroll
    for (int r = 0; r < R; r++) c[padq_strided<LR>(base, r)] = a[r];
}

// UNR: unroll the butterflies of one lane (more LDS reads in flight; the column pass at N = 1024 has no registers to spare)
template <int N, int L, int R, bool UNR>
__device__ __forceinline__ void dif_pass_w(float2 *c, const float2 *tw, int lane) {
    constexpr int BPC = N / R;
    static_assert(BPC % 64 == 0, "wave-local passes need a multiple of 64 butterflies per column");
    if constexpr (UNR) {
#pragma unroll
        for (int t0 = 0; t0 < BPC; t0 += 64) butterfly_w<N, L, R>(c, tw, t0 + lane);
    } else {
#pragma unroll 1
        for (int t0 = 0; t0 < BPC; t0 += 64) butterfly_w<N, L, R>(c, tw, t0 + lane);
    }
    wave_sync();
}

template <int N, int R>
__device__ __forceinline__ void dif_last_w(float2 *c, int lane) {
    constexpr int BPC = N / R, IT = BPC / 64;
    static_assert(BPC % 64 == 0, "wave-local passes need a multiple of 64 butterflies per column");
    float2 a[IT][R];
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int t = it * 64 + lane;
#pragma unroll
        for (int r = 0; r < R; r++) a[it][r] = c[padq(t * R + r)];
        dft<R>(a[it]);
    }
    wave_sync();
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int f0 = freq_of_pos<N>((it * 64 + lane) * R);
#pragma unroll
        for (int r = 0; r < R; r++) c[padq(f0 + r * (N / R))] = a[it][r];
    }
}

template <int N, int L, bool UNR = false>
struct PassesW {
    static __device__ __forceinline__ void run(float2 *c, const float2 *tw, int lane) {
        constexpr int R = radix_of(L);
        if constexpr (L == R) {
            dif_last_w<N, R>(c, lane);
        } else {
            dif_pass_w<N, L, R, UNR>(c, tw, lane);
            PassesW<N, L / R, UNR>::run(c, tw, lane);
        }
    }
};
constexpr bool wave_local(int N) { return N == 512 || N == 1024; }   // the production sizes (n = 1024, 2048)

template <int N>
constexpr int colpitch_of() { return N + (N >> PADSHIFT) + 1; }   // odd: adjacent columns fall into different banks

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
template <int N, int B, bool FUSE>
__global__ __launch_bounds__(Z_THREADS) void fft_z_r2c(float *__restrict__ mesh, int64_t nrows, int pitch_r,
                                                         const float2 *__restrict__ twN, const float2 *__restrict__ tw2N,
                                                         int dbg) {
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
    const int64_t ntiles = FUSE ? (int64_t)N * N : (nrows + B - 1) / B;
    // first row of a tile, and the row of its r-th member
    auto row_of = [&](int64_t tile, int r) -> int64_t {
        if (!FUSE) return tile * B + r;
        const int64_t x = tile / N, y = tile % N;
        return (x + (r >> 1) * N) * NF + y + (r & 1) * N;
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
    auto stage = [&]() {   // registers -> LDS: two complex (= four consecutive reals) per 16-B load
#pragma unroll
        for (int q = 0; q < NLD; q++) touch(regs[q]);
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
                for (int r = tid >> 6; r < nb; r += Z_THREADS / 64) PassesW<N, N>::run(lds + r * CP, tw, tid & 63);
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
                *reinterpret_cast<float4 *>(mesh + (row0 + r) * pitch_r + 2 * k0) =
                    make_float4(X[0].x, X[0].y, X[1].x, X[1].y);
            }
        if (!(dbg & 2) && FUSE) {
            const int x = (int)(tile / N), y = (int)(tile % N);
            float sy, cy, sx, cx;
            sincospif((float)y / (float)N, &sy, &cy);        // W_n^y = exp(-2 pi i y / n), n = 2N
            sincospif((float)x / (float)N, &sx, &cx);
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
                    *reinterpret_cast<float4 *>(mesh + row_of(tile, r) * pitch_r + 2 * k0) =
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
template <int N, int C>
__global__ __launch_bounds__(FFT_THREADS) void fft_cols(float2 *__restrict__ data, int64_t S, int ntile_c,
                                                        int64_t ntiles, int64_t outer_stride, int64_t outer_mod,
                                                        int64_t outer_stride2, const float2 *__restrict__ twN, int dbg) {
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
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < NLD; q++) touch(regs[q]);
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
    if (og >= n_og) return;
    float2 *gcur = tile_ptr(og * ostep + grp, ct);
    prefetch(gcur);
    wait_vmcnt<0>();
    stage();
    for (;;) {
        __syncthreads();
        og += dg, ct += dc;
        if (ct >= ntile_c) ct -= ntile_c, og++;
        const bool has_next = og < n_og;
        float2 *gnext = has_next ? tile_ptr(og * ostep + grp, ct) : gcur;
        if (has_next) prefetch(gnext);
        if (!(dbg & 1)) {
            if constexpr (wave_local(N)) {
#pragma unroll 1
                for (int c = tid >> 6; c < C; c += FFT_THREADS / 64) PassesW<N, N>::run(lds + c * CP, tw, tid & 63);
                __syncthreads();
            } else {
                Passes<N, N>::run(lds, CP, C, tw);
            }
        }
        if (!(dbg & 2)) {
            float2 *g = gcur;
            // constant trip count: the compiler can then count these stores in its vmcnt bookkeeping
#pragma unroll
            for (int q = 0; q < NLD; q++) {
                const int e = q * FFT_THREADS + tid;
                const int c2 = (e % (C / 2)) * 2, f = e / (C / 2);
                if (WHOLE || f < N) {
                    const int p = padq(wave_local(N) ? f : revpos<N>(f));
                    const float2 a = lds[c2 * CP + p], b = lds[(c2 + 1) * CP + p];
                    *reinterpret_cast<float4 *>(g + (int64_t)f * S + c2) = make_float4(a.x, a.y, b.x, b.y);
                }
            }
        }
        if (!has_next) break;
        __syncthreads();
        // the NLD stores above were issued after the prefetch loads: vmcnt(NLD) = all loads landed, stores in flight
        if (dbg & 2) wait_vmcnt<0>();
        else wait_vmcnt<(NLD < 60 ? NLD : 0)>();
        stage();
        gcur = gnext;
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

template <int N, int B, bool FUSE = false>
int launch_z(float *mesh, int64_t nrows, int pitch_r, Tables *t) {
    const size_t lds = (size_t)(2 * N + 2 + B * colpitch_of<N>()) * sizeof(float2);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fft_z_r2c<N, B, FUSE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    const int64_t ntiles = FUSE ? (int64_t)N * N : ceil_div(nrows, B);
    int per_cu = 1;   // persistent grid = exactly the resident workgroups (a larger grid would run in two uneven waves)
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fft_z_r2c<N, B, FUSE>, Z_THREADS, lds));
    const unsigned int grid = (unsigned int)std::min<int64_t>(ntiles, (int64_t)num_cus() * std::max(per_cu, 1));
    ABACUS_LAUNCH("fft_z_r2c", (fft_z_r2c<N, B, FUSE>), dim3(grid), dim3(Z_THREADS), lds, mesh, nrows,
                  pitch_r, t->twHalf.as<float2>(), t->tw2.as<float2>(), getenv("ABACUS_DBG_FFT") ? atoi(getenv("ABACUS_DBG_FFT")) : 0);
    return 0;
}

template <int N, int C>
int launch_cols(const char *name, float2 *data, int64_t S, int ntile_c, int64_t outer, int64_t outer_stride, const float2 *tw,
                int64_t outer_mod = (int64_t)1 << 40, int64_t outer_stride2 = 0) {
    const size_t lds = (size_t)(N + C * colpitch_of<N>()) * sizeof(float2);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fft_cols<N, C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    const int64_t ntiles = outer * ntile_c;
    int per_cu = 1;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fft_cols<N, C>, FFT_THREADS, lds));
    const unsigned int grid = (unsigned int)std::min<int64_t>(ntiles, (int64_t)num_cus() * std::max(per_cu, 1));
    ABACUS_LAUNCH(name, (fft_cols<N, C>), dim3(grid), dim3(FFT_THREADS), lds, data, S, ntile_c, ntiles,
                  outer_stride, outer_mod, outer_stride2, tw, getenv("ABACUS_DBG_FFT") ? atoi(getenv("ABACUS_DBG_FFT")) : 0);
    return 0;
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

bool fft_native_supported(int n) { return n >= 64 && n <= 2048 && (n & (n - 1)) == 0; }

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

int fft_native_r2c_inplace(float *mesh, int n, int pitch_r) {
    ABACUS_TRY(fft_native_zy(mesh, n, pitch_r, n));
    return fft_native_x(mesh, n, pitch_r, n, (int64_t)n * (pitch_r / 2), pitch_r / 2);
}

// Full-mesh transform with the first radix-2 stage of y and x fused into the z pass (see fft_z_r2c<.., FUSE>): the y and
// x passes are n/2-point column transforms with C columns.  Output rows are in the permuted order
// f = 2 (r mod n/2) + (r div n/2) along x and y (power.hip's binning undoes it in its index arithmetic).
template <int N, int C>
int fft3d_fused(float *mesh, int pitch_r, Tables *t, Tables *th) {
    constexpr int H = N / 2;
    const int pitch_c = pitch_r / 2, kzlen = N / 2 + 1;
    const int ntile_c = (kzlen + C - 1) / C;
    if (ntile_c * C > pitch_c) return fail("fft: row pitch too small for the column tiles");
    ABACUS_TRY((launch_z<N / 2, 4, true>(mesh, (int64_t)N * N, pitch_r, t)));
    float2 *data = reinterpret_cast<float2 *>(mesh);
    // y: 2 N half-planes of H rows each, contiguous in memory
    ABACUS_TRY((launch_cols<H, C>("fft_cols_y", data, pitch_c, ntile_c, 2 * (int64_t)N, (int64_t)H * pitch_c,
                                  th->twN.as<float2>())));
    // x: for every y and either half of x, H planes apart by N * pitch_c
    const int64_t S = (int64_t)N * pitch_c;
    return launch_cols<H, C>("fft_cols_x", data, S, ntile_c, 2 * (int64_t)N, pitch_c, th->twN.as<float2>(), N, (int64_t)H * S);
}

// n = 256 only on request (tests against the CPU oracle): small meshes gain nothing from the fused form
int fft_native_fused_supported(int n) { return n == 2048 || n == 1024 || (n == 256 && getenv("ABACUS_FFT_FUSE_SMALL")); }

int fft_native_r2c_fused(float *mesh, int n, int pitch_r) {
    Tables *t, *th;
    ABACUS_TRY(get_tables(n, &t));
    ABACUS_TRY(get_tables(n / 2, &th));
    switch (n) {
        case 256: return fft3d_fused<256, 16>(mesh, pitch_r, t, th);
        case 1024: return fft3d_fused<1024, 16>(mesh, pitch_r, t, th);
        case 2048: return fft3d_fused<2048, 16>(mesh, pitch_r, t, th);
    }
    return fail("fft: the fused transform supports n = 1024 and 2048");
}

int fft_native_release() {
    for (auto &kv : g_tables) {
        ABACUS_TRY(kv.second.twN.release());
        ABACUS_TRY(kv.second.twHalf.release());
        ABACUS_TRY(kv.second.tw2.release());
    }
    g_tables.clear();
    return 0;
}

}  // namespace abacus
