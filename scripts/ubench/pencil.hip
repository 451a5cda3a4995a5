// microbenchmark (round 6, review item 1): the DEPOSIT HALF of a fused "deposit + z pass" kernel at BASELINE config 3.
// A workgroup owns a pencil of PX x PY cells x the full z extent (1024 cells) as 32-bit fixed-point sums in LDS (4 x 8 x 1028 x 4 B
// = 131.6 KB: one workgroup per CU), deposits the pencil's entry list with lines_deposit32's arithmetic (8-byte entries, 16-bit
// in-cell offsets, 18 LDS atomics per entry, two z-adjacent cells per 64-bit atomic), converts, normalises and writes what the z
// pass would write (32 rows x 513 complex = the pencil's 128 KB) - WITHOUT any transform: a lower bound of the fused kernel.
// Entries are synthetic: nearest cells uniform over [-1, PX] x [-1, PY] x [0, 1024) (every cloud that touches the pencil), i.e. the
// fan-out (PX + 2)(PY + 2) / (PX PY) = 1.875 entries per particle of a pencil-keyed list; `per` = entries per pencil
// (1e8 particles on 1024^3: 3052 home particles per pencil x 1.875 = 5722; the tile lists of today hold 1.345 per particle = 4105).
// To be compared with lines_deposit32 + fft_z_r2c of the same workload: 1.39 + 1.93 ms (profiles/r06/bench_default.json, pk_c3).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int PX = 4, PY = 8, PZ = 1024, ZP = PZ + 4, NT = 1024;

template <int MODE>   // 0: full; 1: no atomics (entries read and decoded); 2: no stores; 3: neither
__global__ __launch_bounds__(NT) void pencil_deposit(const unsigned long long *__restrict__ entries, int per, int npencil, float2 *__restrict__ out,
                                                     float norm) {
    extern __shared__ __align__(16) unsigned int tile[];        // [PX * PY][ZP]
    const int tid = threadIdx.x;
    for (int q = tid; q < PX * PY * ZP; q += NT) tile[q] = 0u;
    __syncthreads();
    const float fx = 4194304.f, fxinv = 1.f / 4194304.f;        // 2^22: 5722 entries per pencil leave the guard bit
    typedef float v2f __attribute__((ext_vector_type(2)));
    // the entries of the NEXT pencil are requested before the flush of the current one (as lines_deposit32 requests its lists ahead)
    constexpr int NE = 6;                                        // 6 x 1024 entries cover a pencil's list
    unsigned long long nx[NE];
    auto request = [&](int p) {
#pragma unroll
        for (int q = 0; q < NE; q++) nx[q] = (p < npencil && q * NT + tid < per) ? entries[(int64_t)p * per + q * NT + tid] : ~0ull;
    };
    request(blockIdx.x);
    for (int p = blockIdx.x; p < npencil; p += gridDim.x) {
        unsigned long long cur[NE];
#pragma unroll
        for (int q = 0; q < NE; q++) cur[q] = nx[q];
        request(p + gridDim.x);
#pragma unroll 1
        for (int q = 0; q < NE; q++) {
            const unsigned long long e = cur[q];
            if (e == ~0ull) continue;
            const unsigned int lo = (unsigned int)e, hi = (unsigned int)(e >> 32);
            const int lx = lo & 7, ly = (lo >> 3) & 15, lz = (lo >> 7) & 1023;            // nearest cell + 1 in x, y; z cell
            const float dx = ((float)(hi & 0xffffu) - 32768.f) * (1.f / 65536.f), dy = ((float)(hi >> 16) - 32768.f) * (1.f / 65536.f),
                        dz = ((float)(lo >> 17) - 16384.f) * (1.f / 32768.f);
            float wx[3], wy[3], wz[3];
            const float d3[3] = {dx, dy, dz};
            float *w3[3] = {wx, wy, wz};
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const float d = d3[a], tm = 0.5f + d, tp = 0.5f - d;
                w3[a][1] = 0.75f - d * d;
                w3[a][0] = 0.5f * (tm * tm);
                w3[a][2] = 0.5f * (tp * tp);
            }
            const bool odd = (lz + 1) & 1;                       // row index lz + 1 (one halo cell below): cells lz .. lz + 2 of the row
            const v2f wzA = {odd ? 0.f : wz[0], odd ? wz[0] : wz[1]}, wzB = {odd ? wz[1] : wz[2], odd ? wz[2] : 0.f};
            const v2f fx2 = {fx, fx}, half2 = {0.5f, 0.5f};
            unsigned long long *zpair = reinterpret_cast<unsigned long long *>(tile) + ((lz + 1) >> 1);
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const int cx = lx - 2 + a;
                if ((unsigned)cx >= (unsigned)PX) continue;
#pragma unroll
                for (int b = 0; b < 3; b++) {
                    const int cy = ly - 2 + b;
                    if ((unsigned)cy >= (unsigned)PY) continue;
                    const float wxy = wx[a] * wy[b];
                    const v2f w2 = {wxy, wxy};
                    const v2f sA = __builtin_elementwise_fma(w2 * wzA, fx2, half2), sB = __builtin_elementwise_fma(w2 * wzB, fx2, half2);
                    unsigned long long *cell = zpair + (cx * PY + cy) * (ZP / 2);
                    if (MODE & 1) {
                        if (sA.x + sB.y == -1.f) tile[0] = 1u;   // keeps the arithmetic alive
                    } else {
                        atomicAdd(cell, ((unsigned long long)(unsigned int)sA.y << 32) | (unsigned int)sA.x);
                        atomicAdd(cell + 1, ((unsigned long long)(unsigned int)sB.y << 32) | (unsigned int)sB.x);
                    }
                }
            }
        }
        __syncthreads();
        // flush: fold the z halo (periodic), convert, normalise, write the pencil's 32 rows as the z pass would (513 complex each),
        // re-zero.  One wave per two rows.
        float2 *dst = out + (int64_t)p * (PX * PY) * 520;
        for (int q = tid; q < PX * PY * (PZ / 2); q += NT) {
            const int row = q / (PZ / 2), c = q - row * (PZ / 2);
            unsigned int *r = tile + row * ZP + 2 + 2 * c;
            unsigned int a = r[0], b = r[1];
            if (c == 0) a += tile[row * ZP + 2 + PZ], b += tile[row * ZP + 3 + PZ];
            if (c == PZ / 2 - 1) a += tile[row * ZP], b += tile[row * ZP + 1];
            r[0] = 0u, r[1] = 0u;
            if (!(MODE & 2)) dst[(int64_t)row * 520 + c] = make_float2((float)a * fxinv * norm - 1.f, (float)b * fxinv * norm - 1.f);
        }
        __syncthreads();
        for (int q = tid; q < PX * PY * 4; q += NT) tile[(q >> 2) * ZP + ((q & 3) < 2 ? (q & 3) : PZ + (q & 3))] = 0u;
        __syncthreads();
    }
}

template <class F>
float timeit(F f, int reps = 3) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int r = 0; r < reps; r++) f();
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const int npencil = (1024 / PX) * (1024 / PY);              // 32768 pencils of a 1024^3 mesh
    const size_t lds = (size_t)PX * PY * ZP * 4;
    float2 *out;
    CHECK(hipMalloc(&out, (size_t)npencil * PX * PY * 520 * sizeof(float2)));
    for (int per : {4105, 5722}) {
        std::vector<unsigned long long> h((size_t)npencil * per);
        unsigned long long s = 0x9E3779B97F4A7C15ull;
        for (auto &e : h) {                                      // xorshift: nearest cell + 1 in [0, PX + 1] x [0, PY + 1], z, offsets
            s ^= s << 13, s ^= s >> 7, s ^= s << 17;
            const unsigned int lx = (unsigned int)(s % (PX + 2)), ly = (unsigned int)((s >> 8) % (PY + 2)), lz = (unsigned int)((s >> 16) & 1023);
            const unsigned int lo = lx | (ly << 3) | (lz << 7) | ((unsigned int)((s >> 26) & 0x7fff) << 17);
            e = ((unsigned long long)(unsigned int)(s >> 32) << 32) | lo;
        }
        unsigned long long *d;
        CHECK(hipMalloc(&d, h.size() * 8));
        CHECK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        const float norm = 1.f / 0.0931f;
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pencil_deposit<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pencil_deposit<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pencil_deposit<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pencil_deposit<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        printf("entries per pencil %d (%.3f per particle), %d pencils, %.2f GB of entries, %.2f GB written\n", per, per / 3052.0, npencil,
               h.size() * 8 / 1e9, (double)npencil * PX * PY * 513 * 8 / 1e9);
        printf("  full (atomics + flush)      %6.3f ms\n", timeit([&] { pencil_deposit<0><<<256, NT, lds>>>(d, per, npencil, out, norm); }));
        printf("  no atomics                  %6.3f ms\n", timeit([&] { pencil_deposit<1><<<256, NT, lds>>>(d, per, npencil, out, norm); }));
        printf("  no stores                   %6.3f ms\n", timeit([&] { pencil_deposit<2><<<256, NT, lds>>>(d, per, npencil, out, norm); }));
        printf("  neither                     %6.3f ms\n", timeit([&] { pencil_deposit<3><<<256, NT, lds>>>(d, per, npencil, out, norm); }));
        fflush(stdout);
        CHECK(hipFree(d));
    }
    return 0;
}
