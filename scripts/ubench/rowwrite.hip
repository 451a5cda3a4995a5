// microbenchmark: tile-shaped mesh writes (16x16 rows of 32 floats per workgroup) for different row pitches
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void tile_write(float *grid, int n, long zstride, int ntz, int nty) {
    const int tzi = blockIdx.x % ntz, tyi = (blockIdx.x / ntz) % nty, txi = blockIdx.x / (ntz * nty);
    const int ox = txi * 16, oy = tyi * 16, oz = tzi * 32;
    for (int q = threadIdx.x; q < 16 * 16 * 16; q += 256) {
        const int zp = q & 15, y = (q >> 4) & 15, x = q >> 8;
        float2 *dst = reinterpret_cast<float2 *>(grid + ((long)(ox + x) * n + (oy + y)) * zstride + oz) + zp;
        *dst = make_float2(1.f, 2.f);
    }
}
__global__ __launch_bounds__(256) void tile_write4(float *grid, int n, long zstride, int ntz, int nty) {
    const int tzi = blockIdx.x % ntz, tyi = (blockIdx.x / ntz) % nty, txi = blockIdx.x / (ntz * nty);
    const int ox = txi * 16, oy = tyi * 16, oz = tzi * 32;
    for (int q = threadIdx.x; q < 16 * 16 * 8; q += 256) {
        const int zp = q & 7, y = (q >> 3) & 15, x = q >> 7;
        float4 *dst = reinterpret_cast<float4 *>(grid + ((long)(ox + x) * n + (oy + y)) * zstride + oz) + zp;
        *dst = make_float4(1.f, 2.f, 3.f, 4.f);
    }
}
int main() {
    const int n = 1024;
    float *g; hipMalloc(&g, (size_t)n * n * (n + 64) * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (long zs : {1024L, 1026L, 1056L}) {
        int ntz = n / 32, nty = n / 16, ntx = n / 16;
        for (int v = 0; v < 2; v++) {
            if (v == 1 && (zs % 4)) continue;
            for (int r = 0; r < 2; r++) {
                hipEventRecord(a);
                for (int i = 0; i < 5; i++) {
                    if (v == 0) tile_write<<<ntz * nty * ntx, 256>>>(g, n, zs, ntz, nty);
                    else tile_write4<<<ntz * nty * ntx, 256>>>(g, n, zs, ntz, nty);
                }
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
            printf("zstride %ld %s: %.3f ms  %.2f TB/s\n", zs, v ? "float4" : "float2", ms, (double)n * n * n * 4 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
