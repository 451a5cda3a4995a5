// microbenchmark: floor of an in-place pass over a 2048^3-sized pitched mesh (read 16 B, write it back):
//   rows:    unit-stride rows (the z pass)            - grid-stride float4 read-modify-write
//   cols C:  C adjacent complex columns x N rows tiles (the y / x passes), element stride S - each WG reads a tile
//            (N rows x C*8 B) and writes it back, no LDS, to isolate the memory system from the transform
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#pragma clang diagnostic ignored "-Wunused-value"
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void rmw_rows(float4 *d, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 v = d[i];
        v.x += 1.f;
        d[i] = v;
    }
}
__global__ __launch_bounds__(256) void copy_rows(const float4 *s, float4 *d, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) d[i] = s[i];
}
__global__ __launch_bounds__(256) void read_rows(const float4 *s, float4 *out, int64_t n4) {
    float acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) { float4 v = s[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 1.2345f) out[0] = make_float4(acc, 0, 0, 0);
}
__global__ __launch_bounds__(256) void write_rows(float4 *d, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) d[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
// tile = N rows x C complex; persistent WGs of 512 threads; NLD = N*C/2/512 float4 per thread
template <int N, int C, int MODE>   // MODE 0: rmw, 1: read only
__global__ __launch_bounds__(512) void rmw_cols(float2 *data, int64_t S, int ntile_c, int64_t ntiles, int64_t outer_stride, float4 *out) {
    constexpr int NLD = N * (C / 2) / 512;
    float acc = 0;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        float2 *g = data + (t / ntile_c) * outer_stride + (t % ntile_c) * C;
        float4 r[NLD];
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int e = q * 512 + threadIdx.x;
            const int c2 = (e % (C / 2)) * 2, y = e / (C / 2);
            r[q] = *reinterpret_cast<const float4 *>(g + (int64_t)y * S + c2);
        }
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int e = q * 512 + threadIdx.x;
            const int c2 = (e % (C / 2)) * 2, y = e / (C / 2);
            if (MODE == 0) { r[q].x += 1.f; *reinterpret_cast<float4 *>(g + (int64_t)y * S + c2) = r[q]; }
            else acc += r[q].x;
        }
    }
    if (acc == 1.2345f) out[0] = make_float4(acc, 0, 0, 0);
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
    auto rep = [&](const char *name, double bytes_gb, float ms) { printf("%-44s %8.2f ms  %.2f TB/s\n", name, ms, bytes_gb / ms); };
    for (int grid : {2048, 8192, 32768}) {
        printf("grid %d\n", grid);
        rep("read rows", gb, timeit([&] { read_rows<<<grid, 256>>>((float4 *)d, (float4 *)d2, nfl / 4); }));
        rep("write rows", gb, timeit([&] { write_rows<<<grid, 256>>>((float4 *)d, nfl / 4); }));
        rep("rmw rows in place (r+w)", 2 * gb, timeit([&] { rmw_rows<<<grid, 256>>>((float4 *)d, nfl / 4); }));
        rep("copy rows out of place (r+w)", 2 * gb, timeit([&] { copy_rows<<<grid, 256>>>((float4 *)d, (float4 *)d2, nfl / 4); }));
    }
    {
            for (int grid : {256, 512, 1024}) {
            printf("cols grid %d\n", grid);
            { constexpr int C = 8; const int ntc = pitch_c / C;   /* whole tiles inside the row pitch only */ const int64_t nt = (int64_t)n * ntc;
              rep("y pass C=8  read only", gb * (ntc * C) / (double)pitch_c, timeit([&] { rmw_cols<2048, C, 1><<<grid, 512>>>((float2 *)d, pitch_c, ntc, nt, (int64_t)n * pitch_c, (float4 *)d2); }));
              rep("y pass C=8  rmw", 2 * gb * (ntc * C) / (double)pitch_c, timeit([&] { rmw_cols<2048, C, 0><<<grid, 512>>>((float2 *)d, pitch_c, ntc, nt, (int64_t)n * pitch_c, (float4 *)d2); }));
              rep("x pass C=8  read only", gb * (ntc * C) / (double)pitch_c, timeit([&] { rmw_cols<2048, C, 1><<<grid, 512>>>((float2 *)d, (int64_t)n * pitch_c, ntc, nt, pitch_c, (float4 *)d2); }));
              rep("x pass C=8  rmw", 2 * gb * (ntc * C) / (double)pitch_c, timeit([&] { rmw_cols<2048, C, 0><<<grid, 512>>>((float2 *)d, (int64_t)n * pitch_c, ntc, nt, pitch_c, (float4 *)d2); })); }
            { constexpr int C = 16; const int ntc = pitch_c / C;   /* whole tiles inside the row pitch only */ const int64_t nt = (int64_t)n * ntc;
              rep("y pass C=16 rmw", 2 * gb * (ntc * C) / (double)pitch_c, timeit([&] { rmw_cols<2048, C, 0><<<grid, 512>>>((float2 *)d, pitch_c, ntc, nt, (int64_t)n * pitch_c, (float4 *)d2); }));
              rep("x pass C=16 rmw", 2 * gb * (ntc * C) / (double)pitch_c, timeit([&] { rmw_cols<2048, C, 0><<<grid, 512>>>((float2 *)d, (int64_t)n * pitch_c, ntc, nt, pitch_c, (float4 *)d2); })); }
            { constexpr int C = 32; const int ntc = pitch_c / C;   /* whole tiles inside the row pitch only */ const int64_t nt = (int64_t)n * ntc;
              rep("y pass C=32 rmw", 2 * gb * (ntc * C) / (double)pitch_c, timeit([&] { rmw_cols<2048, C, 0><<<grid, 512>>>((float2 *)d, pitch_c, ntc, nt, (int64_t)n * pitch_c, (float4 *)d2); }));
              rep("x pass C=32 rmw", 2 * gb * (ntc * C) / (double)pitch_c, timeit([&] { rmw_cols<2048, C, 0><<<grid, 512>>>((float2 *)d, (int64_t)n * pitch_c, ntc, nt, pitch_c, (float4 *)d2); })); }
        }
    }
    return 0;
}
