// microbenchmark: achievable read bandwidth for k concurrent double streams (16-B loads), tile per block vs grid-stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int K, int ITER>
__global__ __launch_bounds__(256) void rd_tile(const double *base, int64_t n, int64_t stride, double *out) {
    const int64_t tile0 = (int64_t)blockIdx.x * (512 * ITER);
    double acc = 0;
#pragma unroll
    for (int k = 0; k < ITER; k++) {
        int64_t i = tile0 + k * 512 + 2 * threadIdx.x;
        if (i + 1 < n) {
#pragma unroll
            for (int a = 0; a < K; a++) {
                double2 v = *reinterpret_cast<const double2 *>(base + a * stride + i);
                acc += v.x + v.y;
            }
        }
    }
    if (acc == 1.2345) out[0] = acc;
}
template <int K>
__global__ __launch_bounds__(256) void rd_stride(const double *base, int64_t n, int64_t stride, double *out) {
    double acc = 0;
    for (int64_t i = 2 * ((int64_t)blockIdx.x * 256 + threadIdx.x); i + 1 < n; i += 2 * (int64_t)gridDim.x * 256) {
#pragma unroll
        for (int a = 0; a < K; a++) {
            double2 v = *reinterpret_cast<const double2 *>(base + a * stride + i);
            acc += v.x + v.y;
        }
    }
    if (acc == 1.2345) out[0] = acc;
}
template <class F>
float timeit(F f, int reps = 20) {
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
    const int64_t n = 10000000;
    const int64_t stride = n + 1024 - (n % 1024) + 512;   // arrays like separate hipMallocs (not channel aligned)
    double *d, *out;
    hipMalloc(&d, 5 * stride * 8); hipMalloc(&out, 8);
    hipMemset(d, 0, 5 * stride * 8);
    auto rep = [&](const char *name, int K, float ms) { printf("%-28s K=%d  %.1f us  %.2f TB/s\n", name, K, ms * 1e3, K * n * 8.0 / (ms * 1e-3) / 1e12); };
    int nt4 = (n + 2047) / 2048, nt16 = (n + 8191) / 8192;
    rep("tile2048", 1, timeit([&] { rd_tile<1, 4><<<nt4, 256>>>(d, n, stride, out); }));
    rep("tile2048", 3, timeit([&] { rd_tile<3, 4><<<nt4, 256>>>(d, n, stride, out); }));
    rep("tile2048", 5, timeit([&] { rd_tile<5, 4><<<nt4, 256>>>(d, n, stride, out); }));
    rep("tile8192", 5, timeit([&] { rd_tile<5, 16><<<nt16, 256>>>(d, n, stride, out); }));
    rep("gridstride 2048 blocks", 5, timeit([&] { rd_stride<5><<<2048, 256>>>(d, n, stride, out); }));
    rep("gridstride 1024 blocks", 5, timeit([&] { rd_stride<5><<<1024, 256>>>(d, n, stride, out); }));
    rep("gridstride 4096 blocks", 5, timeit([&] { rd_stride<5><<<4096, 256>>>(d, n, stride, out); }));
    rep("gridstride 2048 blocks", 1, timeit([&] { rd_stride<1><<<2048, 256>>>(d, n, stride, out); }));
    // larger problem: 1e8 elements, 1 stream (beyond the 256 MiB infinity cache)
    double *big; hipMalloc(&big, 800000000ull + 4096);
    hipMemset(big, 0, 800000000ull);
    float ms = timeit([&] { rd_stride<1><<<4096, 256>>>(big, 100000000, 0, out); }, 5);
    printf("1e8 doubles 1 stream gridstride: %.1f us %.2f TB/s\n", ms * 1e3, 8e8 / (ms * 1e-3) / 1e12);
    return 0;
}
