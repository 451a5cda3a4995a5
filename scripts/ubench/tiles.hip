// microbenchmark: in-place read-modify-write of a 2048^3 pitched half-spectrum in tiles of R rows x C complex columns
// (R*C = 16384 complex = one LDS-sized FFT tile), rows taken as G groups of A adjacent rows (R = G*A), groups GS rows apart.
// Tells which tile shapes the memory system sustains, independent of any FFT arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#pragma clang diagnostic ignored "-Wunused-value"
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// a tile: column block cb (C columns), row block: base row rb; rows = rb + g*GS + a
template <int C, int G, int A>
__global__ __launch_bounds__(512) void rmw_tiles(float2 *data, int64_t row_stride, int64_t GS, int ncb, int nrb_inner,
                                                 int64_t ntiles, int64_t plane_stride, int rows_per_plane_tile) {
    constexpr int R = G * A;
    constexpr int NLD = R * (C / 2) / 512;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        // tile order: column block fastest, then row block inside the plane, then plane
        const int cb = (int)(t % ncb);
        const int64_t u = t / ncb;
        const int rb = (int)(u % nrb_inner);
        const int64_t plane = u / nrb_inner;
        float2 *g = data + plane * plane_stride + (int64_t)rb * rows_per_plane_tile * row_stride + (int64_t)cb * C;
        float4 r[NLD];
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int e = q * 512 + threadIdx.x;
            const int c2 = (e % (C / 2)) * 2, y = e / (C / 2);
            const int64_t row = (int64_t)(y / A) * GS + (y % A);
            r[q] = *reinterpret_cast<const float4 *>(g + row * row_stride + c2);
        }
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int e = q * 512 + threadIdx.x;
            const int c2 = (e % (C / 2)) * 2, y = e / (C / 2);
            const int64_t row = (int64_t)(y / A) * GS + (y % A);
            r[q].x += 1.f;
            *reinterpret_cast<float4 *>(g + row * row_stride + c2) = r[q];
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
    const int n = 2048, pitch_c = 1040;
    const int64_t ncplx = (int64_t)n * n * pitch_c;
    float2 *d;
    CHECK(hipMalloc(&d, ncplx * 8 + (1 << 20)));
    CHECK(hipMemset(d, 0, ncplx * 8));
    const int grid = 512;
    auto rep = [&](const char *name, double bytes, float ms) { printf("%-58s %8.2f ms  %.2f TB/s\n", name, ms, bytes / 1e9 / ms); };
    // y-geometry: rows = y index (stride pitch_c), planes = x.  Column blocks: 1024 of the 1040 columns.
#define RUN(C, G, A, GSrows, label)                                                                               \
    {                                                                                                             \
        const int ncb = 1024 / C;                                                                                 \
        const int R = G * A;                                                                                      \
        /* a "row block" of the plane = rows covered by the G groups starting at base rb*A (grouped) or rb*R */   \
        const int nrb = n / R;                                                                                    \
        const int64_t nt = (int64_t)n * nrb * ncb;                                                                \
        const int rpt = (G == 1) ? R : A;                                                                         \
        float ms = timeit([&] { rmw_tiles<C, G, A><<<grid, 512>>>(d, pitch_c, GSrows, ncb, nrb, nt, (int64_t)n * pitch_c, rpt); }); \
        rep(label, 2.0 * nt * R * C * 8, ms);                                                                     \
    }
    RUN(8, 1, 2048, 0, "y: 2048 adjacent rows x 8 cols (64 B)");
    RUN(16, 1, 1024, 0, "y: 1024 adjacent rows x 16 cols (128 B)");
    RUN(32, 1, 512, 0, "y: 512 adjacent rows x 32 cols (256 B)");
    RUN(64, 1, 256, 0, "y: 256 adjacent rows x 64 cols (512 B)");
    RUN(128, 1, 128, 0, "y: 128 adjacent rows x 128 cols (1 KB)");
    RUN(256, 1, 64, 0, "y: 64 adjacent rows x 256 cols (2 KB)");
    RUN(64, 32, 8, 64, "y: 32 groups (64 rows apart) x 8 rows x 64 cols (512 B)");
    RUN(32, 32, 16, 64, "y: 32 groups (64 apart) x 16 rows x 32 cols (256 B)");
    RUN(128, 32, 4, 64, "y: 32 groups (64 apart) x 4 rows x 128 cols (1 KB)");
    RUN(256, 32, 2, 64, "y: 32 groups (64 apart) x 2 rows x 256 cols (2 KB)");
    return 0;
}
