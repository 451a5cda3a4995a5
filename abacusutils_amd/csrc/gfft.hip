// Mixed-radix 3-D R2C transform for the meshes the power-of-two kernels of fft.hip do not cover, and for float64 meshes:
// nmesh = 2^a 3^b 5^c 7^d 11^e 13^f (even) - AbacusHOD.compute_power's default num_cells = 550 (hod/abacus_hod.py:1347),
// the reference's own test mesh 72 (tests/test_power.py:33), 96, 384, 768, 1536 ... - replaces scipy.fft.rfftn at
// analysis/power_spectrum.py:980,986,1059 there.  In place on the padded R2C layout (n, n, pitch), no work area, three
// passes, each one read + one write of the mesh:
//   gfft_rows   rows of n reals -> n/2 + 1 complex: a length-n/2 complex transform of (x[2m], x[2m+1]) and the even / odd split;
//   gfft_cols   y, then x: tiles of C adjacent kz columns x all n rows in LDS (C x 8 B row segments).
// A sequence is transformed by Stockham autosort stages in LDS (natural order out, no digit reversal), one radix
// 2 / 3 / 4 / 5 / 7 / 8 / 11 / 13 per stage from a runtime plan; a stage reads all its butterflies into registers, meets at
// a barrier and writes them back, so the tile needs one buffer.  Templated on the scalar type: float for calc_power's
// default, double for dtype=np.float64 meshes (analysis/power_spectrum.py:808,1148).
// The power-of-two meshes keep the tuned kernels of fft.hip (wave-local passes, fused stages, 0.5 of the HBM peak); this
// path runs at 0.35 - 0.37 per pass (1536^3: 30.4 ms against hipFFT's 38.8, profiles/r04/gfft_vs_hipfft.txt): a pass is
// bound by the vector ALU of its stages (~8.7 ms of compute at 1536^3), which the prefetch hides the loads behind.
#include <cmath>
#include <map>
#include <vector>

#include "common.hpp"
#include "bin_device.hpp"

using namespace abacus;

namespace abacus {
int xdesc_lookup(int n, const BinArgs &b, bool comp, size_t lds_other, const unsigned int **lut, const int **U, int *ncell, int *sh, int *off,
                 int *vtop, const unsigned long long **cnt, const double **ksum, int *ok);   // xbin.hip
}

namespace {

constexpr int G_NT = 512;
constexpr int G_MAXF = 14;

template <typename T>
struct C2 {
    T x, y;
};
template <typename T>
__device__ __forceinline__ C2<T> cmul(C2<T> a, C2<T> b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
template <typename T>
__device__ __forceinline__ C2<T> cadd(C2<T> a, C2<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T>
__device__ __forceinline__ C2<T> csub(C2<T> a, C2<T> b) { return {a.x - b.x, a.y - b.y}; }

template <typename T>
struct alignas(16) CPair {
    C2<T> a, b;
};

// loads through vector types: a conditional assignment of a 16-byte struct from global memory is lowered to a copy into a
// stack slot (scratch, and a wait per load) - a vector value stays in registers
template <typename T>
__device__ __forceinline__ CPair<T> ld_pair(const C2<T> *p) {
    if constexpr (sizeof(T) == 4) {
        typedef float V4 __attribute__((ext_vector_type(4)));
        const V4 v = *reinterpret_cast<const V4 *>(p);
        return {{v.x, v.y}, {v.z, v.w}};
    } else {
        typedef double V2 __attribute__((ext_vector_type(2)));
        const V2 a = reinterpret_cast<const V2 *>(p)[0], b = reinterpret_cast<const V2 *>(p)[1];
        return {{a.x, a.y}, {b.x, b.y}};
    }
}
template <typename T>
__device__ __forceinline__ C2<T> ld_one(const C2<T> *p) {
    typedef T V2 __attribute__((ext_vector_type(2)));
    const V2 a = *reinterpret_cast<const V2 *>(p);
    return {a.x, a.y};
}

struct GPlan {
    int n;                 // sequence length
    int nf;                // stages
    int radix[G_MAXF];
};

// ---- small DFTs in registers (forward, exp(-2 pi i jk / R)), natural order in and out ----------------------------------
template <typename T>
__device__ __forceinline__ void dft4g(C2<T> &a0, C2<T> &a1, C2<T> &a2, C2<T> &a3) {
    const C2<T> t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = csub(a1, a3);
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = {t1.x + t3.y, t1.y - t3.x};   // t1 - i t3
    a3 = {t1.x - t3.y, t1.y + t3.x};   // t1 + i t3
}
// odd prime R: out[k], out[R - k] from the sums / differences of the pairs (j, R - j): (R - 1)^2 / 2 real multiplies by cosines
// and as many by sines
// cos(2 pi i / R), sin(2 pi i / R) for i = 1 .. (R - 1) / 2
template <int R>
__host__ __device__ constexpr double trig_c(int i) {
    if (R == 3) return -0.5;
    if (R == 5) return i == 1 ? 0.30901699437494742410 : -0.80901699437494742410;
    if (R == 7) return i == 1 ? 0.62348980185873353053 : i == 2 ? -0.22252093395631440429 : -0.90096886790241912624;
    if (R == 11)
        return i == 1 ? 0.84125353283118116886 : i == 2 ? 0.41541501300188642553 : i == 3 ? -0.14231483827328514044
             : i == 4 ? -0.65486073394528506406 : -0.95949297361449738989;
    return i == 1 ? 0.88545602565320989590 : i == 2 ? 0.56806474673115580251 : i == 3 ? 0.12053668025532305335
         : i == 4 ? -0.35460488704253562597 : i == 5 ? -0.74851074817110109863 : -0.97094181742605202716;
}
template <int R>
__host__ __device__ constexpr double trig_s(int i) {
    if (R == 3) return 0.86602540378443864676;
    if (R == 5) return i == 1 ? 0.95105651629515357212 : 0.58778525229247312917;
    if (R == 7) return i == 1 ? 0.78183148246802980871 : i == 2 ? 0.97492791218182360702 : 0.43388373911755812048;
    if (R == 11)
        return i == 1 ? 0.54064081745559758211 : i == 2 ? 0.90963199535451837141 : i == 3 ? 0.98982144188093273238
             : i == 4 ? 0.75574957435425828377 : 0.28173255684142969771;
    return i == 1 ? 0.46472317204376854566 : i == 2 ? 0.82298386589365639458 : i == 3 ? 0.99270887409805399280
         : i == 4 ? 0.93501624268541482344 : i == 5 ? 0.66312265824079520238 : 0.23931566428755776715;
}
template <typename T, int R>
__device__ __forceinline__ void dft_prime(C2<T> (&a)[R]) {
    constexpr int H = (R - 1) / 2;
    C2<T> p[H], m[H];
#pragma unroll
    for (int j = 0; j < H; j++) p[j] = cadd(a[j + 1], a[R - 1 - j]), m[j] = csub(a[j + 1], a[R - 1 - j]);
    C2<T> sum = a[0];
#pragma unroll
    for (int j = 0; j < H; j++) sum = cadd(sum, p[j]);
    C2<T> out[R];
    out[0] = sum;
#pragma unroll
    for (int k = 1; k <= H; k++) {
        T re = a[0].x, im = a[0].y, sr = 0, si = 0;
#pragma unroll
        for (int j = 1; j <= H; j++) {
            const int q = (j * k) % R;                    // cos(2 pi q / R), sin(2 pi q / R) from the half table
            const int qi = q <= H ? q : R - q;
            const T cq = (T)trig_c<R>(qi), sq = (T)(q <= H ? trig_s<R>(qi) : -trig_s<R>(qi));
            re += cq * p[j - 1].x, im += cq * p[j - 1].y;
            sr += sq * m[j - 1].x, si += sq * m[j - 1].y;
        }
        // out[k] = A - i B with A = (re, im), B = (sr, si): exp(-i t) = cos t - i sin t
        out[k] = {re + si, im - sr};
        out[R - k] = {re - si, im + sr};
    }
#pragma unroll
    for (int k = 0; k < R; k++) a[k] = out[k];
}
template <typename T, int R>
__device__ __forceinline__ void dftg(C2<T> (&a)[R]) {
    if constexpr (R == 2) {
        const C2<T> t = a[0];
        a[0] = cadd(t, a[1]);
        a[1] = csub(t, a[1]);
    } else if constexpr (R == 4) {
        dft4g(a[0], a[1], a[2], a[3]);
    } else if constexpr (R == 8) {
        dft4g(a[0], a[2], a[4], a[6]);
        dft4g(a[1], a[3], a[5], a[7]);
        const T h = (T)0.70710678118654752440;
        const C2<T> o0 = a[1], o1 = {h * (a[3].x + a[3].y), h * (a[3].y - a[3].x)}, o2 = {a[5].y, -a[5].x},
                    o3 = {h * (a[7].y - a[7].x), -h * (a[7].x + a[7].y)};
        const C2<T> e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6];
        a[0] = cadd(e0, o0), a[4] = csub(e0, o0);
        a[1] = cadd(e1, o1), a[5] = csub(e1, o1);
        a[2] = cadd(e2, o2), a[6] = csub(e2, o2);
        a[3] = cadd(e3, o3), a[7] = csub(e3, o3);
    } else {
        dft_prime<T, R>(a);
    }
}

// ---- Stockham stages over NSEQ interleaved sequences of N elements in LDS ----------------------------------------------------
// Element e of sequence s sits at lds[e * P + s], P = NSEQ + 1: the tile of a column pass is the mesh's own (row, column)
// layout, and the odd pitch spreads the stride-R rows a stage writes over the banks.  A work item is butterfly j of CG adjacent
// sequences (4 in float, 2 in double): the stage's index arithmetic - j -> (j / Ns, j mod Ns), twiddle indices - and the
// twiddle reads are shared by the CG sequences (the first form of this file did them per element and was bound by them:
// 60.9 ms at 1536^3 against hipFFT's 38.5).
//   v[r] = in[j + r N/R] * W_{Ns R}^{r (j mod Ns)};  DFT_R;  out[(j / Ns) Ns R + (j mod Ns) + r Ns] = v[r]
// tw: exp(-2 pi i q / (tws N)) in LDS (the row pass shares the n-entry table of its even / odd split: tws = 2)
__device__ __forceinline__ unsigned int magic_of(unsigned int d) { return 0xffffffffu / d + 1u; }   // floor(a / d), a, d < 65536
__device__ __forceinline__ int fdiv(int a, unsigned int m) { return (int)__umulhi((unsigned int)a, m); }
template <typename T>
constexpr int cg_of() { return sizeof(T) == 4 ? 4 : 2; }

template <typename T, int R, int MAXV>
__device__ __forceinline__ void g_stage(C2<T> *lds, int lgG, int P, int N, int Ns, const C2<T> *tw, int tws) {
    // the large radices take half the sequences per item (their butterflies hold R x CG values: the registers of a prefetched
    // tile must survive them)
    constexpr int CG = R >= 7 ? cg_of<T>() / 2 : cg_of<T>();
    constexpr int MAXIT = (MAXV + R * CG - 1) / (R * CG);
    lgG += cg_of<T>() == CG ? 0 : 1;
    const int BPS = N / R, total = BPS << lgG, step = tws * (N / (Ns * R));
    const unsigned int mN = Ns > 1 ? magic_of(Ns) : 0u;
    C2<T> v[MAXIT][CG][R];
    int off[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; it++) {
        const int item = it * G_NT + threadIdx.x;
        off[it] = -1;
        if (item < total) {
            const int j = item >> lgG, c0 = (item & ((1 << lgG) - 1)) * CG;
            const int q = Ns > 1 ? fdiv(j, mN) : j, k = j - q * Ns;
            const C2<T> *src = lds + j * P + c0;
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int c = 0; c < CG; c++) v[it][c][r] = src[r * BPS * P + c];
            if (Ns > 1) {
#pragma unroll
                for (int r = 1; r < R; r++) {
                    const C2<T> w = tw[r * k * step];
#pragma unroll
                    for (int c = 0; c < CG; c++) v[it][c][r] = cmul(v[it][c][r], w);
                }
            }
#pragma unroll
            for (int c = 0; c < CG; c++) dftg<T, R>(v[it][c]);
            off[it] = (q * Ns * R + k) * P + c0;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < MAXIT; it++)
        if (off[it] >= 0) {
            C2<T> *dst = lds + off[it];
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int c = 0; c < CG; c++) dst[r * Ns * P + c] = v[it][c][r];
        }
    __syncthreads();
}

template <typename T, int MAXV>
__device__ __forceinline__ void g_transform(C2<T> *lds, int lgG, int P, const GPlan &p, const C2<T> *tw, int tws) {
    int Ns = 1;
    for (int f = 0; f < p.nf; f++) {
        const int R = p.radix[f];
        switch (R) {
            case 2: g_stage<T, 2, MAXV>(lds, lgG, P, p.n, Ns, tw, tws); break;
            case 3: g_stage<T, 3, MAXV>(lds, lgG, P, p.n, Ns, tw, tws); break;
            case 4: g_stage<T, 4, MAXV>(lds, lgG, P, p.n, Ns, tw, tws); break;
            case 5: g_stage<T, 5, MAXV>(lds, lgG, P, p.n, Ns, tw, tws); break;
            case 7: g_stage<T, 7, MAXV>(lds, lgG, P, p.n, Ns, tw, tws); break;
            case 8: g_stage<T, 8, MAXV>(lds, lgG, P, p.n, Ns, tw, tws); break;
            case 11: g_stage<T, 11, MAXV>(lds, lgG, P, p.n, Ns, tw, tws); break;
            default: g_stage<T, 13, MAXV>(lds, lgG, P, p.n, Ns, tw, tws); break;
        }
        Ns *= R;
    }
}

// Compile-time plans (make_plan's order; checked against it at launch) for the meshes the bench and compute_power's default
// run: the stage loop unrolls with N, Ns and the tile pitch as constants - the index arithmetic of a stage folds and the LDS
// offsets r * (N / R) * P, r * Ns * P become instruction immediates instead of one vector add per access.
template <int N>
struct GFixed {
    static constexpr int nf = 0;
    static constexpr int radix[1] = {0};
};
#define GFIXED(N, ...)                                                \
    template <>                                                       \
    struct GFixed<N> {                                                \
        static constexpr int radix[] = {__VA_ARGS__};                 \
        static constexpr int nf = sizeof(radix) / sizeof(int);        \
    };
GFIXED(1536, 8, 8, 8, 3)
GFIXED(768, 8, 8, 4, 3)
GFIXED(384, 8, 8, 2, 3)
GFIXED(550, 2, 5, 5, 11)
GFIXED(275, 5, 5, 11)
#undef GFIXED
template <int N>
bool fixed_plan_matches(const GPlan &p) {
    if (p.n != N || p.nf != GFixed<N>::nf) return false;
    for (int f = 0; f < p.nf; f++)
        if (p.radix[f] != GFixed<N>::radix[f]) return false;
    return true;
}
template <typename T, int MAXV, int N, int LGG, int F = 0, int NS = 1>
__device__ __forceinline__ void g_transform_fixed(C2<T> *lds, const C2<T> *tw, int tws) {
    if constexpr (F < GFixed<N>::nf) {
        constexpr int R = GFixed<N>::radix[F];
        g_stage<T, R, MAXV>(lds, LGG, (cg_of<T>() << LGG) + 1, N, NS, tw, tws);
        g_transform_fixed<T, MAXV, N, LGG, F + 1, NS * R>(lds, tw, tws);
    }
}

// ---- rows: n reals -> n/2 + 1 complex, in place (row pitch `pitch_r` scalars) ------------------------------------------
// a tile is NSEQ = CG << lgG rows (a power of two); LDS: [n twiddles][(n / 2) x (NSEQ + 1)]
// FN > 0: the mesh size and lgG are compile-time (GFixed<FN / 2>)
template <typename T, int MAXV, int FN = 0, int FLG = 0>
__global__ __launch_bounds__(G_NT) void gfft_rows(T *__restrict__ mesh, int64_t nrows, int n_, int pitch_r, int lgG_, GPlan p,
                                                  const C2<T> *__restrict__ twn, int dbg) {
    const int n = FN > 0 ? FN : n_, lgG = FN > 0 ? FLG : lgG_;
    extern __shared__ __align__(16) unsigned char smem[];
    C2<T> *tw = reinterpret_cast<C2<T> *>(smem);
    C2<T> *lds = tw + n;
    for (int q = threadIdx.x; q < n; q += G_NT) tw[q] = twn[q];
    const int NSEQ = cg_of<T>() << lgG, P = NSEQ + 1, Nh = n / 2;
    const unsigned int mH1 = magic_of(Nh + 1);
    for (int64_t r0 = (int64_t)blockIdx.x * NSEQ; r0 < nrows; r0 += (int64_t)gridDim.x * NSEQ) {
        const int ns = (int)min((int64_t)NSEQ, nrows - r0);
        {   // a batch of 16-byte loads (two complex of a row) in flight before its LDS stores: with one workgroup per CU nothing
            // else hides the latency.  An odd n / 2 reads one complex into the row's padding (pitch_r >= n + 2) and drops it.
            constexpr int B = MAXV / 4;
            using Pair = CPair<T>;
            const int np2 = (Nh + 1) >> 1, total2 = NSEQ * np2;
            const unsigned int mP = magic_of(np2);
            Pair v[B];
#pragma unroll 1
            for (int q0 = threadIdx.x; q0 < total2; q0 += B * G_NT) {
#pragma unroll
                for (int it = 0; it < B; it++) {
                    const int q = q0 + it * G_NT, s = fdiv(q, mP), m2 = q - s * np2;
                    v[it] = (q < total2 && s < ns && !(dbg & 2)) ? reinterpret_cast<const Pair *>(mesh + (r0 + s) * pitch_r)[m2]
                                                                 : Pair{{(T)0, (T)0}, {(T)0, (T)0}};
                }
#pragma unroll
                for (int it = 0; it < B; it++) {
                    const int q = q0 + it * G_NT, s = fdiv(q, mP), m2 = q - s * np2;
                    if (q < total2) {
                        lds[(2 * m2) * P + s] = v[it].a;
                        if (2 * m2 + 1 < Nh) lds[(2 * m2 + 1) * P + s] = v[it].b;
                    }
                }
            }
        }
        __syncthreads();
        if (!(dbg & 1)) {
            if constexpr (FN > 0) g_transform_fixed<T, MAXV, FN / 2, FLG>(lds, tw, 2);
            else g_transform<T, MAXV>(lds, lgG, P, p, tw, 2);
        }
        // X[k] = (Z[k] + conj Z[Nh - k]) / 2 - (i / 2) w^k (Z[k] - conj Z[Nh - k]),  w = exp(-2 pi i / n),  Z[Nh] = Z[0]
        for (int q = threadIdx.x; q < ns * (Nh + 1); q += G_NT) {
            const int s = fdiv(q, mH1), k = q - s * (Nh + 1);
            const C2<T> zk = lds[(k == Nh ? 0 : k) * P + s], zm = lds[(k == 0 ? 0 : Nh - k) * P + s];
            const C2<T> e = {(T)0.5 * (zk.x + zm.x), (T)0.5 * (zk.y - zm.y)};      // even part
            const C2<T> o = {(T)0.5 * (zk.x - zm.x), (T)0.5 * (zk.y + zm.y)};      // (Z[k] - conj Z[Nh - k]) / 2
            const C2<T> w = k == Nh ? C2<T>{(T)-1, (T)0} : tw[k];
            const C2<T> wo = cmul(w, o);
            if (!(dbg & 2) || e.x == (T)1.2345) reinterpret_cast<C2<T> *>(mesh + (r0 + s) * pitch_r)[k] = {e.x + wo.y, e.y - wo.x};   // e - i w o
        }
        __syncthreads();
    }
}

// ---- columns: element (row, col) of tile t at data[tile_base(t) + row * S + col] ------------------------------------------
// tiles: outer index o < nouter (stride outer_stride) x column tile ct < ntile_c (C = CG << lgG columns).
// One workgroup per CU (the tile fills the LDS), so the loads of the NEXT tile are issued into registers before the stages
// of the current one and land in LDS after its write-back: the memory phase (7 ms of a 1536^3 pass) runs under the
// stages (8.7 ms) instead of in front of them (1536^3: 12.2 -> 9.7 ms per pass).  The row pass keeps batched loads in
// front of its stages: walking a row per thread group, which the one-pointer prefetch needs, measured 12.1 ms against 10.3.
template <typename T, int MAXV, int FN = 0, int FLG = 0>
__global__ __launch_bounds__(G_NT) void gfft_cols(C2<T> *__restrict__ data, int n_, int64_t S, int lgG_, int ntile_c, int ncols,
                                                  int64_t nouter, int64_t outer_stride, GPlan p, const C2<T> *__restrict__ twn, int dbg, float xcut) {
    const int n = FN > 0 ? FN : n_, lgG = FN > 0 ? FLG : lgG_;
    extern __shared__ __align__(16) unsigned char smem[];
    C2<T> *tw = reinterpret_cast<C2<T> *>(smem);
    C2<T> *lds = tw + n;
    for (int q = threadIdx.x; q < n; q += G_NT) tw[q] = twn[q];
    const int lgC = lgG + (cg_of<T>() == 4 ? 2 : 1), C = 1 << lgC, P = C + 1;
    const int64_t ntiles = nouter * ntile_c;
    const int total2 = n << (lgC - 1);            // 16-byte (32-byte in double) pairs of adjacent columns: <= MAXV / 2 per thread
    constexpr int NPRE = MAXV / 2;
    CPair<T> pre[NPRE];
    auto tile_base = [&](int64_t t, int &nc) {
        const int64_t o = t / ntile_c;
        const int c0 = (int)(t - o * ntile_c) * C;
        nc = min(C, ncols - c0);
        return data + o * outer_stride + c0;
    };
    // pair q = it * G_NT + tid of a tile: column pair c = 2 (q mod C / 2) is the thread's own for every it, its row advances
    // by dr = G_NT / (C / 2) - one pointer and a uniform stride, no per-load offsets to keep
    const int c = (threadIdx.x & (C / 2 - 1)) * 2, row0 = threadIdx.x >> (lgC - 1), dr = G_NT >> (lgC - 1);
    auto issue = [&](int64_t t) {
        int nc;
        const C2<T> *src = tile_base(t, nc) + (int64_t)row0 * S + c;
        const int64_t stride = (int64_t)dr * S;
        const int mode = (dbg & 2) ? 0 : c + 1 < nc ? 2 : c < nc ? 1 : 0;
#pragma unroll
        for (int it = 0; it < NPRE; it++) {
            if constexpr (sizeof(T) == 4) {
                pre[it] = CPair<T>{{(T)0, (T)0}, {(T)0, (T)0}};
                if (row0 + it * dr < n) {
                    if (mode == 2) pre[it] = *reinterpret_cast<const CPair<T> *>(src);
                    else if (mode == 1) pre[it].a = *src;
                }
            } else {     // (a conditional 16-byte struct assignment from global memory would go through a stack slot)
                T ax = 0, ay = 0, bx = 0, by = 0;
                if (row0 + it * dr < n) {
                    if (mode == 2) {
                        const CPair<T> v = ld_pair(src);
                        ax = v.a.x, ay = v.a.y, bx = v.b.x, by = v.b.y;
                    } else if (mode == 1) {
                        const C2<T> v = ld_one(src);
                        ax = v.x, ay = v.y;
                    }
                }
                pre[it].a = {ax, ay}, pre[it].b = {bx, by};
            }
            src += stride;
        }
    };
    // xcut > 0 (x pass, the caller bins up to |k|^2 = xcut in fundamental units and reads nothing beyond): tiles whose first
    // column already has ky^2 + kz^2 > xcut hold no mode anybody reads - neither loaded nor transformed
    auto dead = [&](int64_t t) {
        if (!(xcut > 0.f)) return false;
        const int64_t o = t / ntile_c;
        const int j = (int)o, jj = j < n / 2 ? j : j - n, k0 = (int)(t - o * ntile_c) * C;
        return (float)(jj * jj + k0 * k0) > xcut;
    };
    auto next = [&](int64_t t) {
        do t += gridDim.x;
        while (t < ntiles && dead(t));
        return t;
    };
    int64_t t = blockIdx.x;
    if (t < ntiles && dead(t)) t = next(t);
    if (t < ntiles) issue(t);
    for (int64_t tn; t < ntiles; t = tn) {
        tn = next(t);
        {
            C2<T> *l = lds + row0 * P + c;
#pragma unroll
            for (int it = 0; it < NPRE; it++) {
                if (row0 + it * dr < n) l[0] = pre[it].a, l[1] = pre[it].b;
                l += dr * P;
            }
        }
        __syncthreads();
        if (tn < ntiles && !(dbg & 4)) issue(tn);
        if (!(dbg & 1)) {
            if constexpr (FN > 0) g_transform_fixed<T, MAXV, FN, FLG>(lds, tw, 1);
            else g_transform<T, MAXV>(lds, lgG, P, p, tw, 1);
        }
        int nc;
        C2<T> *base = tile_base(t, nc);
        for (int q = threadIdx.x; q < total2; q += G_NT) {
            const int row = q >> (lgC - 1), c = (q & (C / 2 - 1)) * 2;
            const CPair<T> w = {lds[row * P + c], lds[row * P + c + 1]};
            if ((dbg & 2) && w.a.x != (T)1.2345) continue;
            C2<T> *dst = base + (int64_t)row * S + c;
            if (c + 1 < nc) *reinterpret_cast<CPair<T> *>(dst) = w;
            else if (c < nc) *dst = w.a;
        }
        __syncthreads();
        if (tn < ntiles && (dbg & 4)) issue(tn);
    }
}

// ---- last pass fused with the (k, mu) / multipole binning, mixed-radix meshes ------------------------------------------------
// The x pass of gfft_cols without its write-back: the transformed tile (all n x-rows of C kz columns of one y row) is binned
// from LDS - |delta_k|^2 of the modes +i and -i of a column share their bin - into the workgroup's float64 LDS histogram, with
// the cached geometry descriptor of xbin.hip (xdesc_device.hpp: bin of a mode from a cell table indexed by the float bits of
// kmag2 and per-kz mu thresholds, N_mode and sum |k| precomputed; validated over every mode of the mesh when it is built).
// The counterpart of fft_x_bin2 (xbin.hip) for the sizes gfft serves; natural frequency order, one wave per column, lane l
// walks the run i = l RUN .. l RUN + RUN - 1 of |kx| and keeps the sums of its current bin in registers.
// Replaces gfft_cols (x) + spectrum_bin for the auto power of one non-interlaced field: 4M bytes read instead of 4M read +
// 4M written + 4M read.  analysis/power_spectrum.py:150-300 (bin_kmu), :707-727, :1058-1069.
#include "xdesc_device.hpp"

struct GXArgs {
    int n, kzlen, lgG, ntile_c, ustride;
    int64_t S, ys;            // element stride along x, along y
    float inv2;               // f32(1/M)^2; interlaced pair: f32(0.5/M)^2
    const float *W;           // (n,) compensation window or nullptr
    const C2<float> *data2;   // interlaced pair: the half-cell-shifted field after its rows and y passes
    const float2 *phase;      // interlaced pair: exp(i pi q / n), q in [0, 2n)
};

// INTER: the interlaced pair of fields in one pass (the reference's default estimator, power_spectrum.py:951-998; what
// AbacusHOD.compute_power runs on its default 550^3 mesh): both fields' tiles are transformed in LDS side by side and a mode is
// binned as |(a + a' exp(i pi (i + j + k) / n)) f32(0.5 / M)|^2 with signed-folded i, j (shift_field_fft, :904-948; i = n/2
// folds to -n/2) - what spectrum_bin<INTER> computes from two spectra in HBM, without the two x passes writing them and the
// binning reading them back.
// LEAN: the twiddle table and the per-kz mu thresholds stay in global memory (read through the vector cache) - what lets a
// 1536-row tile, the float64 histogram of 512 k bins and the cell table share the 160 KiB of LDS
template <bool INTER, bool LEAN, int FN = 0, int FLG = 0>
__global__ __launch_bounds__(G_NT) void gfft_x_bin(const C2<float> *__restrict__ data, GXArgs g, GPlan p, const C2<float> *__restrict__ twn,
                                                   BinArgs b, XDesc d) {
    typedef float T;
    constexpr int MAXV = 24;
    extern __shared__ __align__(16) unsigned char smem[];
    const int n = FN > 0 ? FN : g.n, lgG = FN > 0 ? FLG : g.lgG, lgC = lgG + 2, C = 1 << lgC, P = C + 1;   // FN > 0: compile-time plan (GFixed)
    const int Nk = b.Nk, Nmu = b.Nmu, nrow = Nk + 2, nbx = nrow * Nmu;
    C2<T> *twl = reinterpret_cast<C2<T> *>(smem);
    C2<T> *lds = twl + (LEAN ? 0 : n);
    C2<T> *lds2 = lds + (size_t)n * P;                                 // the shifted field's tile (INTER)
    double *h_sum = reinterpret_cast<double *>(lds + (size_t)n * P * (INTER ? 2 : 1));
    double *h_m2 = h_sum + nbx, *h_m4 = h_m2 + nrow;
    unsigned int *lut = reinterpret_cast<unsigned int *>(h_m4 + nrow);
    int *Ul = reinterpret_cast<int *>(lut + d.ncell);
    float *Wl = reinterpret_cast<float *>(Ul + (LEAN ? 0 : g.kzlen * g.ustride));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const C2<T> *tw = LEAN ? twn : twl;
    if (!LEAN)
        for (int q = tid; q < n; q += G_NT) twl[q] = twn[q];
    for (int q = tid; q < nbx; q += G_NT) h_sum[q] = 0.0;
    for (int q = tid; q < 2 * nrow; q += G_NT) h_m2[q] = 0.0;
    for (int q = tid; q < d.ncell; q += G_NT) lut[q] = d.lut[q];
    if (!LEAN)
        for (int q = tid; q < g.kzlen * g.ustride; q += G_NT) Ul[q] = d.U[(q / g.ustride) * XD_USTRIDE + q % g.ustride];
    const bool comp = g.W != nullptr;
    if (comp)
        for (int q = tid; q < n; q += G_NT) Wl[q] = g.W[q];
    const unsigned int *lut0 = lut - d.off;
    const int sh = d.sh, nmu1 = Nmu - 1;
    const int64_t ntiles = (int64_t)n * g.ntile_c;
    constexpr int NPRE = MAXV / 2;
    CPair<T> pre[NPRE], pre2[INTER ? NPRE : 1];
    const int c = (tid & (C / 2 - 1)) * 2, row0 = tid >> (lgC - 1), dr = G_NT >> (lgC - 1);
    auto issue = [&](int64_t t) {
        const int64_t o = t / g.ntile_c;
        const int c0 = (int)(t - o * g.ntile_c) * C, nc = min(C, g.kzlen - c0);
        const int64_t off0 = o * g.ys + c0 + (int64_t)row0 * g.S + c;
        const C2<T> *src = data + off0;
        const int64_t stride = (int64_t)dr * g.S;
        const int mode = c + 1 < nc ? 2 : c < nc ? 1 : 0;
#pragma unroll
        for (int it = 0; it < NPRE; it++) {
            pre[it] = CPair<T>{{(T)0, (T)0}, {(T)0, (T)0}};
            if (row0 + it * dr < n) {
                if (mode == 2) pre[it] = *reinterpret_cast<const CPair<T> *>(src);
                else if (mode == 1) pre[it].a = *src;
            }
            src += stride;
        }
        if (INTER) {
            const C2<T> *src2 = g.data2 + off0;
#pragma unroll
            for (int it = 0; it < NPRE; it++) {
                pre2[INTER ? it : 0] = CPair<T>{{(T)0, (T)0}, {(T)0, (T)0}};
                if (row0 + it * dr < n) {
                    if (mode == 2) pre2[INTER ? it : 0] = *reinterpret_cast<const CPair<T> *>(src2);
                    else if (mode == 1) pre2[INTER ? it : 0].a = *src2;
                }
                src2 += stride;
            }
        }
    };
    auto dead = [&](int64_t t) {     // the tile's first column already beyond the last edge for every kx
        const int64_t o = t / g.ntile_c;
        const int j = (int)o, jj = j < n / 2 ? j : j - n, k0 = (int)(t - o * g.ntile_c) * C;
        return jj * jj + k0 * k0 > d.vtop;
    };
    auto next = [&](int64_t t) {
        do t += gridDim.x;
        while (t < ntiles && dead(t));
        return t;
    };
    const int RUN = (n / 2 + 1 + 63) / 64;
    __syncthreads();
    int64_t t = blockIdx.x;
    if (t < ntiles && dead(t)) t = next(t);
    if (t < ntiles) issue(t);
    for (int64_t tn; t < ntiles; t = tn) {
        tn = next(t);
        {
            C2<T> *l = lds + row0 * P + c;
#pragma unroll
            for (int it = 0; it < NPRE; it++) {
                if (row0 + it * dr < n) {
                    l[0] = pre[it].a, l[1] = pre[it].b;
                    if (INTER) l[(size_t)n * P] = pre2[INTER ? it : 0].a, l[(size_t)n * P + 1] = pre2[INTER ? it : 0].b;
                }
                l += dr * P;
            }
        }
        __syncthreads();
        if (tn < ntiles) issue(tn);
        if constexpr (FN > 0) {
            g_transform_fixed<T, MAXV, FN, FLG>(lds, tw, 1);
            if (INTER) g_transform_fixed<T, MAXV, FN, FLG>(lds2, tw, 1);
        } else {
            g_transform<T, MAXV>(lds, lgG, P, p, tw, 1);       // ends on a workgroup barrier: the whole tile is transformed
            if (INTER) g_transform<T, MAXV>(lds2, lgG, P, p, tw, 1);
        }
        const int64_t o = t / g.ntile_c;
        const int j = (int)o, jj = j < n / 2 ? j : j - n, c0 = (int)(t - o * g.ntile_c) * C;
#pragma unroll 1
        for (int cc = wave; cc < C; cc += G_NT / 64) {
            const int k = c0 + cc, r2 = jj * jj + k * k;
            if (k >= g.kzlen || r2 > d.vtop) continue;             // padding column / the whole column beyond the last edge
            int Uk[7];
#pragma unroll
            for (int m = 0; m < 7; m++) Uk[m] = m < nmu1 ? (LEAN ? d.U[k * XD_USTRIDE + m] : Ul[k * g.ustride + m]) : -2;
            const float k2f = (float)(k * k), scale = (k == 0 ? 1.f : 2.f) * g.inv2;      // weight (:258-262) x f32(1/M)^2 (:1058-1060)
            const float wjk = comp ? Wl[j] * Wl[k] : 1.f;
            int cur = 0, curk = 0;
            float sp = 0.f, s2 = 0.f, s4 = 0.f;
            bool first = true;
            const int i0 = lane * RUN;
            int v = r2 + i0 * i0;
            for (int s = 0; s < RUN; s++) {
                const int i = i0 + s;
                if (2 * i > n) break;
                C2<T> a = lds[i * P + cc];
                if (INTER) {                                                               // a + a' exp(i pi m / n), m = ii + jj + k
                    int m = (2 * i < n ? i : i - n) + jj + k;                              // (i = n/2 folds to -n/2, :940-942)
                    m += m < 0 ? 2 * n : 0;
                    const float2 ph = g.phase[m];
                    const C2<T> a2 = lds2[i * P + cc];
                    a = {a.x + (a2.x * ph.x - a2.y * ph.y), a.y + (a2.x * ph.y + a2.y * ph.x)};
                }
                float pw = a.x * a.x + a.y * a.y;                                          // get_raw_power (:726)
                if (comp) {                                                                // (:1065-1069)
                    const float sA = __builtin_amdgcn_rcpf(Wl[i] * wjk);
                    pw *= sA * sA;
                }
                if (i > 0 && 2 * i < n) {                                                  // the mode -i: same |k|, same mu
                    C2<T> bq = lds[(n - i) * P + cc];
                    if (INTER) {
                        int m = -i + jj + k;
                        m += m < 0 ? 2 * n : 0;
                        const float2 ph = g.phase[m];
                        const C2<T> b2 = lds2[(n - i) * P + cc];
                        bq = {bq.x + (b2.x * ph.x - b2.y * ph.y), bq.y + (b2.x * ph.y + b2.y * ph.x)};
                    }
                    float pb = bq.x * bq.x + bq.y * bq.y;
                    if (comp) {
                        const float sB = __builtin_amdgcn_rcpf(Wl[n - i] * wjk);
                        pb *= sB * sB;
                    }
                    pw += pb;
                }
                const float vf1 = fmaxf((float)v, 1.f);                                    // kmag2 = 0: the cell of 1, mu2 = 0 (:243)
                const int eb = xd_eb(lut0, sh, v, vf1);
                int bmu = 0;
#pragma unroll
                for (int m = 0; m < 7; m++) bmu += v <= Uk[m] ? 1 : 0;
                const int tb = eb * Nmu + bmu;
                pw *= scale;
                const float mu2 = k2f * __builtin_amdgcn_rcpf(vf1);
                const float t2 = pw * mu2, t4 = t2 * mu2;
                if (!first && tb != cur) {
                    if ((unsigned int)(curk - 1) < (unsigned int)Nk) {
                        atomicAdd(&h_sum[cur], (double)sp);
                        if (b.Np > 0) atomicAdd(&h_m2[curk], (double)s2), atomicAdd(&h_m4[curk], (double)s4);
                    }
                    sp = s2 = s4 = 0.f;
                }
                cur = tb, curk = eb, first = false;
                sp += pw, s2 += t2, s4 += t4;
                v += 2 * i + 1;
            }
            if (!first && (unsigned int)(curk - 1) < (unsigned int)Nk) {
                atomicAdd(&h_sum[cur], (double)sp);
                if (b.Np > 0) atomicAdd(&h_m2[curk], (double)s2), atomicAdd(&h_m4[curk], (double)s4);
            }
        }
        __syncthreads();     // every wave is done with the tile
    }
    __syncthreads();
    for (int q = tid; q < Nk * Nmu; q += G_NT) {
        const double sm = h_sum[q + Nmu];           // row eb = bk + 1
        if (sm != 0.0) atomicAdd(&b.g_sum[q], sm);
        if (blockIdx.x == 0) b.g_cnt[q] = d.cnt[q], b.g_ksum[q] = d.ksum[q];
    }
    if (b.Np > 0) {
        for (int bk = tid; bk < Nk; bk += G_NT) {
            double s0 = 0.0;
            for (int m = 0; m < Nmu; m++) s0 += h_sum[(bk + 1) * Nmu + m];
            const double m2 = h_m2[bk + 1], m4 = h_m4[bk + 1];
            for (int q = 0; q < b.Np; q++) {
                const double c0q = b.poledeg[q] >= 0 ? b.polecoef[q][0] : 0.0, c1q = b.poledeg[q] >= 1 ? b.polecoef[q][1] : 0.0,
                             c2q = b.poledeg[q] >= 2 ? b.polecoef[q][2] : 0.0;
                const double vq = c0q * s0 + c1q * m2 + c2q * m4;
                if (vq != 0.0) atomicAdd(&b.g_pole[q * Nk + bk], vq);
            }
        }
    }
}

// ---- host -------------------------------------------------------------------------------------------------------------
bool make_plan(int n, GPlan &p) {
    p.n = n, p.nf = 0;
    int m = n;
    for (int R : {8, 4, 2, 3, 5, 7, 11, 13})
        while (m % R == 0 && m > 1) {
            if (p.nf >= G_MAXF) return false;
            p.radix[p.nf++] = R;
            m /= R;
        }
    return m == 1;
}

template <typename T>
struct GTables {
    std::map<int, DevBuf> tw;   // n -> exp(-2 pi i q / n), q < n
    int get(int n, const C2<T> **out) {
        auto it = tw.find(n);
        if (it == tw.end()) {
            std::vector<C2<T>> h((size_t)n);
            for (int q = 0; q < n; q++) {
                const double a = -2.0 * M_PI * (double)q / (double)n;
                h[q] = {(T)std::cos(a), (T)std::sin(a)};
            }
            DevBuf b;
            ABACUS_TRY(b.reserve(h.size() * sizeof(C2<T>)));
            HIP_TRY(hipMemcpyAsync(b.p, h.data(), h.size() * sizeof(C2<T>), hipMemcpyHostToDevice, stream()));
            HIP_TRY(hipStreamSynchronize(stream()));
            it = tw.emplace(n, b).first;
        }
        *out = it->second.as<C2<T>>();
        return 0;
    }
    int release() {
        for (auto &kv : tw) ABACUS_TRY(kv.second.release());
        tw.clear();
        return 0;
    }
};
GTables<float> g_tw32;
GTables<double> g_tw64;
template <typename T>
GTables<T> &tables();
template <>
GTables<float> &tables<float>() { return g_tw32; }
template <>
GTables<double> &tables<double>() { return g_tw64; }

template <typename T>
constexpr int maxv() { return sizeof(T) == 4 ? 24 : 12; }   // complex values a thread holds in a stage: 512 threads x 24 x 8 B = 96 KiB of LDS

int num_cus_g() {
    int dev = 0, ncu = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    return ncu;
}

// hipFuncSetAttribute is not free: once per kernel and size, not per launch
int kernel_lds(const void *kern, size_t lds) {
    static std::map<const void *, size_t> set;
    size_t &have = set[kern];
    if (lds > have) {
        HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        have = lds;
    }
    return 0;
}

template <typename T, int MAXV>
int launch_rows(T *mesh, int n, int pitch_r, const GPlan &ph, const C2<T> *twn, int ncu, int dbg) {
    constexpr int CG = sizeof(T) == 4 ? 4 : 2;
    const int Nh = n / 2, cap = MAXV * G_NT;
    int lgG = 0;                                 // NSEQ = CG << lgG rows per tile, NSEQ x n / 2 values <= cap
    while ((CG << (lgG + 1)) <= 32 && (CG << (lgG + 1)) * Nh <= cap) lgG++;
    const int NSEQ = CG << lgG;
    const size_t lds = ((size_t)n + (size_t)Nh * (NSEQ + 1)) * sizeof(C2<T>);
    void (*kern)(T *, int64_t, int, int, int, GPlan, const C2<T> *, int) = gfft_rows<T, MAXV>;
    if constexpr (sizeof(T) == 4) {
        if (!option("gfft_nofixed")) {
            if (n == 1536 && lgG == 2 && fixed_plan_matches<768>(ph)) kern = gfft_rows<T, MAXV, 1536, 2>;
            else if (n == 768 && lgG == 3 && fixed_plan_matches<384>(ph)) kern = gfft_rows<T, MAXV, 768, 3>;
            else if (n == 550 && lgG == 3 && fixed_plan_matches<275>(ph)) kern = gfft_rows<T, MAXV, 550, 3>;
        }
    }
    ABACUS_TRY(kernel_lds(reinterpret_cast<const void *>(kern), lds));
    const int64_t nrows = (int64_t)n * n;
    const unsigned int grid = (unsigned int)std::min<int64_t>(ceil_div(nrows, NSEQ), (int64_t)ncu * (lds > 80 * 1024 ? 1 : 2));
    ABACUS_LAUNCH("gfft_rows", kern, dim3(grid), dim3(G_NT), lds, mesh, nrows, n, pitch_r, lgG, ph, twn, dbg);
    return 0;
}

template <typename T>
int r2c_inplace(T *mesh, int n, int pitch_r, float xcut = 0.f, bool skip_x = false) {
    constexpr int MAXV = maxv<T>();
    GPlan ph, pn;
    if (n < 4 || (n & 1) || !make_plan(n / 2, ph) || !make_plan(n, pn)) return fail("gfft: mesh size %d is not an even product of 2, 3, 5, 7, 11, 13", n);
    const int cap = MAXV * G_NT;                                   // complex values of a tile
    if (n > cap) return fail("gfft: mesh size %d beyond the LDS tile", n);
    if (pitch_r < n + 2 || (pitch_r & 1)) return fail("gfft: row pitch %d too small for %d + 2", pitch_r, n);
    const C2<T> *twn;
    ABACUS_TRY(tables<T>().get(n, &twn));
    const int ncu = num_cus_g(), dbg = (int)option("gfft_dbg");   // diagnostic: 1 no stages, 2 no mesh traffic (results are wrong)
    constexpr int CG = sizeof(T) == 4 ? 4 : 2;
    // rows: half-size tiles (two workgroups per CU) measured slower at 1536 (13.3 against 11.9 ms) and equal below
    ABACUS_TRY((launch_rows<T, MAXV>(mesh, n, pitch_r, ph, twn, ncu, dbg)));
    const int pitch_c = pitch_r / 2, kzlen = n / 2 + 1;
    int lgG = 0;
    while ((CG << (lgG + 1)) <= 16 && (CG << (lgG + 1)) * n <= cap) lgG++;
    const int C = CG << lgG;
    const size_t lds = ((size_t)n + (size_t)n * (C + 1)) * sizeof(C2<T>);
    void (*kern)(C2<T> *, int, int64_t, int, int, int, int64_t, int64_t, GPlan, const C2<T> *, int, float) = gfft_cols<T, MAXV>;
    if constexpr (sizeof(T) == 4) {
        if (!option("gfft_nofixed")) {
            if (n == 1536 && lgG == 1 && fixed_plan_matches<1536>(pn)) kern = gfft_cols<T, MAXV, 1536, 1>;
            else if (n == 768 && lgG == 2 && fixed_plan_matches<768>(pn)) kern = gfft_cols<T, MAXV, 768, 2>;
            else if (n == 550 && lgG == 2 && fixed_plan_matches<550>(pn)) kern = gfft_cols<T, MAXV, 550, 2>;
        }
    }
    ABACUS_TRY(kernel_lds(reinterpret_cast<const void *>(kern), lds));
    const int ntile_c = (kzlen + C - 1) / C;
    const unsigned int grid = (unsigned int)std::min<int64_t>((int64_t)n * ntile_c, (int64_t)ncu * (lds > 80 * 1024 ? 1 : 2));
    C2<T> *data = reinterpret_cast<C2<T> *>(mesh);
    // y: for every x plane, columns along y (stride pitch_c); x: for every y row, columns along x (stride n * pitch_c)
    ABACUS_LAUNCH("gfft_cols_y", kern, dim3(grid), dim3(G_NT), lds, data, n, (int64_t)pitch_c, lgG, ntile_c, kzlen, (int64_t)n,
                  (int64_t)n * pitch_c, pn, twn, dbg, 0.f);
    if (skip_x) return 0;       // the caller runs the last pass fused with the binning (gfft_x_bin)
    ABACUS_LAUNCH("gfft_cols_x", kern, dim3(grid), dim3(G_NT), lds, data, n, (int64_t)n * pitch_c, lgG, ntile_c, kzlen, (int64_t)n,
                  (int64_t)pitch_c, pn, twn, dbg, xcut);
    return 0;
}

// LDS of gfft_x_bin beside the cell table, and the column-tile width it leaves room for (lgG: C = 4 << lgG; -1: none)
size_t gxbin_lds_other(int n, int C, int Nk, int Nmu, bool comp, bool inter = false, bool lean = false) {
    return ((lean ? 0 : (size_t)n) + (size_t)n * (C + 1) * (inter ? 2 : 1)) * 8 + (size_t)(Nk + 2) * Nmu * 8 + (size_t)2 * (Nk + 2) * 8 +
           (lean ? 0 : (size_t)(n / 2 + 1) * std::max(Nmu - 1, 1) * 4) + (comp ? (size_t)n * 4 : 0) + 16;
}
// column-tile width (lgG: C = 4 << lgG) and whether the lean form (twiddles and mu thresholds in global memory) is needed; -1: none
int gxbin_lgG(int n, int Nk, int Nmu, bool comp, bool inter = false, bool *lean_out = nullptr) {
    for (int lean = 0; lean < 2; lean++)
        for (int lgG = 2; lgG >= 1; lgG--) {     // 16 or 8 columns: 4 would read 32-byte row segments (0.4 of the 64-byte rate)
            const int C = 4 << lgG;
            // room for the cell table beside it: 6 KiB at least (24 KiB with the fine bins a mesh beyond 1152 brings)
            const size_t room = n > 1152 ? 24 * 1024 : 6 * 1024;
            if ((int64_t)C * n <= (int64_t)maxv<float>() * G_NT && gxbin_lds_other(n, C, Nk, Nmu, comp, inter, lean != 0) + room <= 160 * 1024) {
                if (lean_out) *lean_out = lean != 0;
                return lgG;
            }
        }
    return -1;
}
bool gxbin_shape_ok(int n, const BinArgs &b) {
    if (!b.h_edges2 || b.Np > 2 || b.Nmu > 8 || b.Nmu < 1) return false;
    for (int q = 0; q < b.Np; q++)
        if (b.poledeg[q] > 2) return false;
    GPlan p;
    return n >= 8 && n <= 2048 && !(n & 1) && make_plan(n, p) && make_plan(n / 2, p);
}

}  // namespace

namespace abacus {

// even sizes whose factors are 2, 3, 5, 7, 11, 13 and that fit the LDS tile (float: n <= 6144 by the cap, held to 4096)
bool gfft_supported(int n, int is_double) {
    GPlan p;
    if (n < 8 || n > 4096 || (n & 1)) return false;
    if (!make_plan(n, p) || !make_plan(n / 2, p)) return false;
    const int CG = is_double ? 2 : 4, cap = (is_double ? maxv<double>() : maxv<float>()) * G_NT;
    if (n * CG > cap) return false;                                   // at least one column group per tile
    // the dynamic LDS launch_rows / r2c_inplace will ask for must fit the CU's 160 KiB (double: 64 n bytes of column tile,
    // beyond it above n = 2560; such sizes go to hipFFT instead of failing at the launch)
    const size_t cbytes = is_double ? 16 : 8;
    int lgR = 0, lgC = 0;
    while ((CG << (lgR + 1)) <= 32 && (CG << (lgR + 1)) * (n / 2) <= cap) lgR++;
    while ((CG << (lgC + 1)) <= 16 && (CG << (lgC + 1)) * n <= cap) lgC++;
    const size_t lds_rows = ((size_t)n + (size_t)(n / 2) * ((CG << lgR) + 1)) * cbytes;
    const size_t lds_cols = ((size_t)n + (size_t)n * ((CG << lgC) + 1)) * cbytes;
    return std::max(lds_rows, lds_cols) <= (size_t)160 * 1024;
}
int gfft_r2c_inplace_f32(float *mesh, int n, int pitch_r, float xcut) { return r2c_inplace<float>(mesh, n, pitch_r, xcut); }
int gfft_r2c_inplace_f64(double *mesh, int n, int pitch_r) { return r2c_inplace<double>(mesh, n, pitch_r); }
// rows and y pass only: what gfft_x_bin_run continues from
int gfft_r2c_zy_f32(float *mesh, int n, int pitch_r) { return r2c_inplace<float>(mesh, n, pitch_r, 0.f, true); }

// can the fused last pass serve this mesh / histogram?  (builds the geometry descriptor of (n, edges) on first use)
bool gfft_xbin_supported(int n, const BinArgs &b, bool comp, bool inter) {
    if (!gxbin_shape_ok(n, b)) return false;
    bool lean = false;
    const int lgG = gxbin_lgG(n, b.Nk, b.Nmu, comp, inter, &lean);
    if (lgG < 0) return false;
    XDesc d;
    int ok = 0;
    if (xdesc_lookup(n, b, comp, gxbin_lds_other(n, 4 << lgG, b.Nk, b.Nmu, comp, inter, lean), &d.lut, &d.U, &d.ncell, &d.sh, &d.off, &d.vtop, &d.cnt, &d.ksum,
                     &ok) != 0)
        return false;
    return ok != 0;
}

// `mesh` holds the transform after gfft_r2c_zy_f32; bins |delta_k|^2 of every mode into the accumulators of `b` (zeroed by the
// caller), N_mode and sum |k| from the cached descriptor
int gfft_x_bin_run(const float *mesh, int n, int pitch_r, float inv_size, const float *W_dev, const BinArgs &b, const float *mesh2,
                   const void *phase) {
    const bool comp = W_dev != nullptr, inter = mesh2 != nullptr;
    if (!gxbin_shape_ok(n, b)) return fail("gfft_x_bin: unsupported mesh / histogram");
    bool lean = false;
    const int lgG = gxbin_lgG(n, b.Nk, b.Nmu, comp, inter, &lean);
    if (lgG < 0) return fail("gfft_x_bin: histogram does not fit beside a tile of %d rows", n);
    const int C = 4 << lgG, pitch_c = pitch_r / 2, kzlen = n / 2 + 1;
    const size_t other = gxbin_lds_other(n, C, b.Nk, b.Nmu, comp, inter, lean);
    XDesc d;
    int ok = 0;
    ABACUS_TRY(xdesc_lookup(n, b, comp, other, &d.lut, &d.U, &d.ncell, &d.sh, &d.off, &d.vtop, &d.cnt, &d.ksum, &ok));
    if (!ok) return fail("gfft_x_bin: no geometry descriptor for this histogram");
    d.ustride = XD_USTRIDE;
    GPlan pn;
    if (!make_plan(n, pn)) return fail("gfft_x_bin: mesh size %d", n);
    const C2<float> *twn;
    ABACUS_TRY(tables<float>().get(n, &twn));
    GXArgs g;
    g.n = n, g.kzlen = kzlen, g.lgG = lgG, g.ntile_c = (kzlen + C - 1) / C, g.ustride = std::max(b.Nmu - 1, 1);
    g.S = (int64_t)n * pitch_c, g.ys = pitch_c, g.W = W_dev;
    g.data2 = reinterpret_cast<const C2<float> *>(mesh2), g.phase = static_cast<const float2 *>(phase);
    const float hs = inter ? (float)(0.5 * (double)inv_size) : inv_size;      // f32(0.5 / M) (:993-997) or f32(1 / M) (:1058-1060)
    g.inv2 = hs * hs;
    if (g.ntile_c * C > pitch_c) return fail("gfft_x_bin: row pitch too small");
    const size_t lds = other + (size_t)d.ncell * 4;
    using XKern = void (*)(const C2<float> *, GXArgs, GPlan, const C2<float> *, BinArgs, XDesc);
    XKern kern = inter ? (lean ? gfft_x_bin<true, true> : gfft_x_bin<true, false>) : (lean ? gfft_x_bin<false, true> : gfft_x_bin<false, false>);
    if (!option("gfft_nofixed")) {
#define GXF(I, L, N, G) \
    if (inter == I && lean == L && n == N && lgG == G && fixed_plan_matches<N>(pn)) kern = gfft_x_bin<I, L, N, G>;
        GXF(false, true, 1536, 1)
        GXF(false, false, 768, 2) GXF(false, false, 768, 1) GXF(true, false, 768, 1)
        GXF(false, false, 550, 2) GXF(false, false, 550, 1) GXF(true, false, 550, 2) GXF(true, false, 550, 1)
#undef GXF
    }
    ABACUS_TRY(kernel_lds(reinterpret_cast<const void *>(kern), lds));
    const unsigned int grid = (unsigned int)std::min<int64_t>((int64_t)n * g.ntile_c, (int64_t)num_cus_g() * (lds > 80 * 1024 ? 1 : 2));
    const C2<float> *m0 = reinterpret_cast<const C2<float> *>(mesh);
    ABACUS_LAUNCH("gfft_x_bin", kern, dim3(grid), dim3(G_NT), lds, m0, g, pn, twn, b, d);
    return 0;
}
int gfft_release() {
    ABACUS_TRY(g_tw32.release());
    return g_tw64.release();
}

}  // namespace abacus
