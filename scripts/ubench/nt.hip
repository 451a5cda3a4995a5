// microbenchmark (round 6): do non-temporal hints move the floors the P(k) passes sit on?
//   - the in-place read-modify-write of a 2048^3 mesh (fft_z_r2c / fft_cols: 4.5 - 4.9 TB/s with plain accesses)
//   - the write of the mesh in 128-byte row pieces at the tile flush's stride (lines_deposit32: 4.8 TB/s)
//   - the out-of-place copy (the 1024^3 ping-pong)
// every variant with plain / non-temporal loads x plain / non-temporal stores (`__builtin_nontemporal_*` on 16-byte vectors)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int NTL, int NTS>
__device__ __forceinline__ v4f ld(const v4f *p) { return NTL ? __builtin_nontemporal_load(p) : *p; }
template <int NTS>
__device__ __forceinline__ void st(v4f *p, v4f v) { if (NTS) __builtin_nontemporal_store(v, p); else *p = v; }

template <int NTL, int NTS>
__global__ __launch_bounds__(256) void rmw_rows(v4f *d, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        v4f v = ld<NTL, 0>(d + i);
        v.x += 1.f;
        st<NTS>(d + i, v);
    }
}
template <int NTL, int NTS>
__global__ __launch_bounds__(256) void copy_rows(const v4f *s, v4f *d, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) st<NTS>(d + i, ld<NTL, 0>(s + i));
}
// the tile flush: a workgroup of 256 threads writes tiles of 16 x 16 rows of 32 floats (128 B), rows zstride floats apart
template <int NTS>
__global__ __launch_bounds__(256) void tile_write(float *grid, int n, int64_t zstride, int64_t ntiles) {
    const int tid = threadIdx.x, zq = tid & 7, yy = (tid >> 3) & 15, x0 = tid >> 7;      // 8 x 16 x 2
    const int ntz = n / 32, nty = n / 16;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tz = (int)(t % ntz), ty = (int)((t / ntz) % nty), tx = (int)(t / ((int64_t)ntz * nty));
        float *dst = grid + ((int64_t)(tx * 16 + x0) * n + (ty * 16 + yy)) * zstride + tz * 32 + 4 * zq;
#pragma unroll
        for (int s = 0; s < 8; s++, dst += 2 * (int64_t)n * zstride) st<NTS>(reinterpret_cast<v4f *>(dst), v4f{1.f, 2.f, 3.f, (float)s});
    }
}
// column tiles: N rows x C complex, in-place rmw (y pass shape), 512 threads
template <int N, int C, int NTL, int NTS>
__global__ __launch_bounds__(512) void rmw_cols(float2 *data, int64_t S, int ntile_c, int64_t ntiles, int64_t outer_stride) {
    constexpr int NLD = N * (C / 2) / 512;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        float2 *g = data + (t / ntile_c) * outer_stride + (t % ntile_c) * C;
        v4f r[NLD];
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int e = q * 512 + threadIdx.x, c2 = (e % (C / 2)) * 2, y = e / (C / 2);
            r[q] = ld<NTL, 0>(reinterpret_cast<const v4f *>(g + (int64_t)y * S + c2));
        }
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int e = q * 512 + threadIdx.x, c2 = (e % (C / 2)) * 2, y = e / (C / 2);
            r[q].x += 1.f;
            st<NTS>(reinterpret_cast<v4f *>(g + (int64_t)y * S + c2), r[q]);
        }
    }
}
template <class F>
float timeit(F f, int reps = 3) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; r++) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}
int main() {
    const int n = 2048, pitch_r = 2080, pitch_c = pitch_r / 2;
    const int64_t nfl = (int64_t)n * n * pitch_r;
    float *d, *d2;
    CHECK(hipMalloc(&d, nfl * 4)); CHECK(hipMalloc(&d2, nfl * 4));
    CHECK(hipMemset(d, 0, nfl * 4)); CHECK(hipMemset(d2, 0, nfl * 4));
    const double gb = nfl * 4.0 / 1e9;
    auto rep = [&](const char *name, double bytes_gb, float ms) { printf("%-52s %8.2f ms  %.2f TB/s\n", name, ms, bytes_gb / ms); fflush(stdout); };
    const int grid = 8192;
    rep("rmw rows            plain load, plain store", 2 * gb, timeit([&] { rmw_rows<0, 0><<<grid, 256>>>((v4f *)d, nfl / 4); }));
    rep("rmw rows            nt load,    plain store", 2 * gb, timeit([&] { rmw_rows<1, 0><<<grid, 256>>>((v4f *)d, nfl / 4); }));
    rep("rmw rows            plain load, nt store", 2 * gb, timeit([&] { rmw_rows<0, 1><<<grid, 256>>>((v4f *)d, nfl / 4); }));
    rep("rmw rows            nt load,    nt store", 2 * gb, timeit([&] { rmw_rows<1, 1><<<grid, 256>>>((v4f *)d, nfl / 4); }));
    rep("copy out of place   plain, plain", 2 * gb, timeit([&] { copy_rows<0, 0><<<grid, 256>>>((const v4f *)d, (v4f *)d2, nfl / 4); }));
    rep("copy out of place   nt load, plain store", 2 * gb, timeit([&] { copy_rows<1, 0><<<grid, 256>>>((const v4f *)d, (v4f *)d2, nfl / 4); }));
    rep("copy out of place   plain load, nt store", 2 * gb, timeit([&] { copy_rows<0, 1><<<grid, 256>>>((const v4f *)d, (v4f *)d2, nfl / 4); }));
    rep("copy out of place   nt, nt", 2 * gb, timeit([&] { copy_rows<1, 1><<<grid, 256>>>((const v4f *)d, (v4f *)d2, nfl / 4); }));
    const int64_t ntiles = (int64_t)(n / 16) * (n / 16) * (n / 32);
    const double tgb = (double)n * n * n * 4.0 / 1e9;
    for (int g : {1024, 2048}) {
        printf("tile flush, grid %d\n", g);
        rep("tile write 16x16x32 plain store", tgb, timeit([&] { tile_write<0><<<g, 256>>>(d, n, pitch_r, ntiles); }));
        rep("tile write 16x16x32 nt store", tgb, timeit([&] { tile_write<1><<<g, 256>>>(d, n, pitch_r, ntiles); }));
    }
    {
        constexpr int C = 16; const int ntc = pitch_c / C; const int64_t nt = (int64_t)2 * n * ntc;     // the fused form's y pass: 1024-row half columns
        const double cgb = 2 * gb * (ntc * C) / (double)pitch_c;
        for (int g : {256, 512}) {
            printf("y pass shape (1024 rows x 16 columns), grid %d\n", g);
            rep("cols rmw plain, plain", cgb, timeit([&] { rmw_cols<1024, C, 0, 0><<<g, 512>>>((float2 *)d, pitch_c, ntc, nt, (int64_t)1024 * pitch_c); }));
            rep("cols rmw nt load, plain store", cgb, timeit([&] { rmw_cols<1024, C, 1, 0><<<g, 512>>>((float2 *)d, pitch_c, ntc, nt, (int64_t)1024 * pitch_c); }));
            rep("cols rmw plain load, nt store", cgb, timeit([&] { rmw_cols<1024, C, 0, 1><<<g, 512>>>((float2 *)d, pitch_c, ntc, nt, (int64_t)1024 * pitch_c); }));
            rep("cols rmw nt, nt", cgb, timeit([&] { rmw_cols<1024, C, 1, 1><<<g, 512>>>((float2 *)d, pitch_c, ntc, nt, (int64_t)1024 * pitch_c); }));
        }
    }
    return 0;
}
