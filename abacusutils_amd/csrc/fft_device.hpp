// Device-side building blocks of the hand-written FFT passes (fft.hip) and of the fused last pass + binning (xbin.hip):
// asynchronous 16-B loads with hand-counted vmcnt, radix-2/4/8 butterflies, the in-place Sande-Tukey (DIF) passes over
// padded LDS columns - workgroup-wide (`Passes`) and wave-local (`PassesW`: one wave owns a column, no workgroup barrier).
// Everything here has internal linkage (included inside an anonymous namespace by its users).
#pragma once

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
__device__ __forceinline__ void wait_vmcnt_upto16(int k) {   // runtime count (uniform) 0 .. 16
    switch (k) {
        case 9: wait_vmcnt<9>(); break;
        case 10: wait_vmcnt<10>(); break;
        case 11: wait_vmcnt<11>(); break;
        case 12: wait_vmcnt<12>(); break;
        case 13: wait_vmcnt<13>(); break;
        case 14: wait_vmcnt<14>(); break;
        case 15: wait_vmcnt<15>(); break;
        case 16: wait_vmcnt<16>(); break;
        default: wait_vmcnt_upto8(k); break;
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

// 16 = 4 x 4: DFT-4 over n1 of x[4 n1 + n2], twiddles W16^(n2 k1), DFT-4 over n2; X[k1 + 4 k2] lands at 4 k1 + k2
template <>
__device__ __forceinline__ void dft<16>(float2 (&a)[16]) {
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) dft4(a[n2], a[4 + n2], a[8 + n2], a[12 + n2]);
    const float h = 0.70710678118654752440f, c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;
    auto mul = [](float2 v, float wr, float wi) { return make_float2(v.x * wr - v.y * wi, v.x * wi + v.y * wr); };
    // element (k1, n2) sits at 4 k1 + n2
    a[5] = mul(a[5], c1, -s1);                               // W^1
    a[6] = make_float2(h * (a[6].x + a[6].y), h * (a[6].y - a[6].x));       // W^2 = (h, -h)
    a[7] = mul(a[7], s1, -c1);                               // W^3
    a[9] = make_float2(h * (a[9].x + a[9].y), h * (a[9].y - a[9].x));       // W^2
    a[10] = make_float2(a[10].y, -a[10].x);                  // W^4 = -i
    a[11] = make_float2(h * (a[11].y - a[11].x), -h * (a[11].x + a[11].y)); // W^6 = (-h, -h)
    a[13] = mul(a[13], s1, -c1);                             // W^3
    a[14] = make_float2(h * (a[14].y - a[14].x), -h * (a[14].x + a[14].y)); // W^6
    a[15] = mul(a[15], -c1, s1);                             // W^9
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) dft4(a[4 * k1], a[4 * k1 + 1], a[4 * k1 + 2], a[4 * k1 + 3]);
    float2 t[16];
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++)
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) t[k1 + 4 * k2] = a[4 * k1 + k2];
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = t[k];
}

constexpr int radix_of(int L) { return L >= 8 ? 8 : L; }   // greedy radix-8, then one radix-4 or radix-2 pass
// pass schedule of an N-point transform: 1024 = 8 x 8 x 16 (three trips through LDS instead of the four of 8 x 8 x 8 x 2)
template <int N>
constexpr int radix_at(int L) { return (N == 1024 && L == 16) ? 16 : radix_of(L); }

// position of frequency f after the in-place DIF passes (mixed-radix digit reversal)
template <int N>
__device__ __forceinline__ int revpos(int f) {
    int pos = 0, L = N;
#pragma unroll
    for (int it = 0; it < 12; it++) {
        if (L == 1) break;
        const int R = radix_at<N>(L);
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
        const int R = radix_at<N>(L);
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
        constexpr int R = radix_at<N>(L);
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

// a middle pass whose twiddles are lane constants: with LR = L/R <= 64 butterfly t = t0 + lane has j = lane % LR in every
// trip, so its R - 1 twiddles exp(-2 pi i j r / L) live in registers for the whole kernel (twr[r], r >= 1) - no table reads
template <int N, int L, int R>
__device__ __forceinline__ void dif_pass_w_regtw(float2 *c, const float2 (&twr)[R], int lane) {
    constexpr int BPC = N / R, LR = L / R;
    static_assert(BPC % 64 == 0 && LR <= 64 && 64 % LR == 0, "twiddles constant per lane");
#pragma unroll 1
    for (int t0 = 0; t0 < BPC; t0 += 64) {
        const int t = t0 + lane;
        const int base = (t / LR) * L + t % LR;
        float2 a[R];
#pragma unroll
        for (int r = 0; r < R; r++) a[r] = c[padq_strided<LR>(base, r)];
        dft<R>(a);
#pragma unroll
        for (int r = 1; r < R; r++) a[r] = cmul(a[r], twr[r]);
#pragma unroll
        for (int r = 0; r < R; r++) c[padq_strided<LR>(base, r)] = a[r];
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
        constexpr int R = radix_at<N>(L);
        if constexpr (L == R) {
            dif_last_w<N, R>(c, lane);
        } else {
            dif_pass_w<N, L, R, UNR>(c, tw, lane);
            PassesW<N, L / R, UNR>::run(c, tw, lane);
        }
    }
};
constexpr bool wave_local(int N) { return N == 512 || N == 1024; }   // the production sizes (n = 1024, 2048)

// Registers -> LDS THROUGH the first radix-8 DIF pass (sub-length N): when load q of a thread is element j0 + q * RS of
// its sequence (RS = N/8: one butterfly, 8 loads; RS = N/16: two butterflies, 16 loads, q = 2 r + b), the loads of a
// thread ARE the inputs r = 0..7 of butterfly j = j0 + b * RS of that pass.  Each 16-B load carries two sequences
// (.xy -> cA at index offset jA, .zw -> cB at jB: two columns of a column tile, or two adjacent elements of a row).
// The tile is never staged raw and read back: one LDS round trip less; the remaining passes are PassesW<N, N/8>.
// the same with the twiddles of the thread's two butterflies in registers (one butterfly per sequence: NLD = 8)
template <int N>
__device__ __forceinline__ void stage_pass1_regtw(v4f (&regs)[8], float2 *cA, float2 *cB, int jA, int jB,
                                                  const float2 (&twA)[8], const float2 (&twB)[8]) {
    constexpr int LR = N / 8;
    float2 u[8], w[8];
#pragma unroll
    for (int r = 0; r < 8; r++) u[r] = make_float2(regs[r].x, regs[r].y), w[r] = make_float2(regs[r].z, regs[r].w);
    dft<8>(u);
    dft<8>(w);
#pragma unroll
    for (int r = 1; r < 8; r++) u[r] = cmul(u[r], twA[r]), w[r] = cmul(w[r], twB[r]);
#pragma unroll
    for (int r = 0; r < 8; r++) cA[padq_strided<LR>(jA, r)] = u[r], cB[padq_strided<LR>(jB, r)] = w[r];
}

template <int N, int NLD, int RS>
__device__ __forceinline__ void stage_pass1(v4f (&regs)[NLD], float2 *cA, float2 *cB, int jA, int jB, const float2 *tw) {
    constexpr int LR = N / 8, NB = LR / RS;
    static_assert(NB * RS == LR && NLD == 8 * NB && (NB == 1 || NB == 2), "loads of a thread = whole radix-8 butterflies");
#pragma unroll
    for (int b = 0; b < NB; b++) {
        float2 u[8], w[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            u[r] = make_float2(regs[r * NB + b].x, regs[r * NB + b].y);
            w[r] = make_float2(regs[r * NB + b].z, regs[r * NB + b].w);
        }
        dft<8>(u);
        dft<8>(w);
        const int ja = jA + b * RS, jb = jB + b * RS;
#pragma unroll
        for (int r = 1; r < 8; r++) {
            u[r] = cmul(u[r], tw[ja * r]);
            w[r] = cmul(w[r], tw[jb * r]);
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            cA[padq_strided<LR>(ja, r)] = u[r];
            cB[padq_strided<LR>(jb, r)] = w[r];
        }
    }
}

template <int N>
constexpr int colpitch_of() { return N + (N >> PADSHIFT) + 1; }   // odd: adjacent columns fall into different banks
