// Sparse gather of record lines, as hod_exact / hod_emit / the filter's stage 2 issue them: of N records of RB bytes a
// fraction `dens` (ascending, shuffled inside groups of 2048 like the filter's queue) is visited; a thread reads LB bytes
// of its record.  Prints the rate in records/s and the bytes/s actually requested.  (what bounds the HOD
// kernels of a dense tracer mix.)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int LB>   // bytes read per record: 64 or 128
__global__ __launch_bounds__(256) void gather(const char *__restrict__ rec, int rb, const unsigned int *__restrict__ idx, int n,
                                              double *__restrict__ out) {
    double acc = 0;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const double4 *p = reinterpret_cast<const double4 *>(rec + (size_t)idx[j] * rb);
#pragma unroll
        for (int q = 0; q < LB / 32; q++) {
            const double4 v = p[q];
            acc += v.x + v.y + v.z + v.w;
        }
    }
    if (acc == 1.2345) out[0] = acc;
}

int main() {
    const int N = 10000000;
    for (int rb : {128, 192}) {
        char *rec;
        CK(hipMalloc(&rec, (size_t)N * rb));
        CK(hipMemset(rec, 0, (size_t)N * rb));
        double *out;
        CK(hipMalloc(&out, 8));
        for (double dens : {0.01, 0.05, 0.1, 0.27, 0.5, 1.0}) {
            std::mt19937 rng(7);
            std::vector<unsigned int> idx;
            std::uniform_real_distribution<double> U(0, 1);
            for (int g = 0; g < N; g += 2048) {
                const size_t b = idx.size();
                for (int i = g; i < std::min(N, g + 2048); i++)
                    if (U(rng) < dens) idx.push_back(i);
                std::shuffle(idx.begin() + b, idx.end(), rng);
            }
            unsigned int *d;
            CK(hipMalloc(&d, idx.size() * 4));
            CK(hipMemcpy(d, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
            for (int lb : {64, 128}) {
                for (int grid : {1024, 4096}) {
                    hipEvent_t e0, e1;
                    CK(hipEventCreate(&e0));
                    CK(hipEventCreate(&e1));
                    float best = 1e9;
                    for (int rep = 0; rep < 5; rep++) {
                        CK(hipEventRecord(e0));
                        if (lb == 64) gather<64><<<grid, 256>>>(rec, rb, d, (int)idx.size(), out);
                        else gather<128><<<grid, 256>>>(rec, rb, d, (int)idx.size(), out);
                        CK(hipEventRecord(e1));
                        CK(hipEventSynchronize(e1));
                        float ms;
                        CK(hipEventElapsedTime(&ms, e0, e1));
                        best = std::min(best, ms);
                    }
                    printf("rec %3d B  density %.2f  read %3d B  grid %4d: %8zu records  %7.1f us  %.2e rec/s  %.2f TB/s requested\n", rb,
                           dens, lb, grid, idx.size(), best * 1e3, idx.size() / (best * 1e-3), idx.size() * (double)lb / (best * 1e-3) / 1e12);
                }
            }
            CK(hipFree(d));
        }
        CK(hipFree(rec));
    }
    return 0;
}
