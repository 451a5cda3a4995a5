// microbenchmark: LDS atomic add throughput (no return) for u32 / f32 / u64 / f64 on pseudo-random cell indices of an
// 8192-cell tile, 256 threads per workgroup, 2 workgroups per CU (the tsc_tile_deposit shape)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#pragma clang diagnostic ignored "-Wunused-value"
template <typename T>
__global__ __launch_bounds__(256) void k(T *out, int iters, int spread) {
    __shared__ T tile[8192];
    for (int q = threadIdx.x; q < 8192; q += 256) tile[q] = (T)0;
    __syncthreads();
    unsigned int s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 27; u++) {
            s = s * 1664525u + 1013904223u;
            const int idx = spread ? (s >> 8) & 8191 : ((threadIdx.x * 37 + u * 301) & 8191);
            atomicAdd(&tile[idx], (T)1);
        }
    }
    __syncthreads();
    if (tile[threadIdx.x] == (T)123456789) out[0] = tile[0];
}
template <typename T>
void run(const char *name, int spread) {
    T *out;
    hipMalloc(&out, 64);
    const int iters = 200, grid = 512;
    k<T><<<grid, 256>>>(out, iters, spread);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    k<T><<<grid, 256>>>(out, iters, spread);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double lane_ops = (double)grid * 256 * iters * 27;
    // per CU: 2 WGs -> lane-ops per CU = lane_ops/256
    printf("%-6s %-8s %8.3f ms   %.2f lane-atomics/cycle/CU (2.4 GHz)   %.1f cycles per wave-instruction per CU\n", name,
           spread ? "random" : "strided", ms, lane_ops / 256 / (ms * 1e-3 * 2.4e9), 64.0 / (lane_ops / 256 / (ms * 1e-3 * 2.4e9)));
    hipFree(out);
}
int main() {
    for (int spread = 0; spread < 2; spread++) {
        run<unsigned int>("u32", spread);
        run<float>("f32", spread);
        run<unsigned long long>("u64", spread);
        run<double>("f64", spread);
    }
    return 0;
}
