// microbenchmark: scattered writes of aligned runs of R bytes (what a multisplit pass issues): every run lands at a hashed,
// R-aligned (or, `shift` != 0, misaligned by that many bytes) place of a 2-GiB buffer, written by R/16 adjacent lanes with
// one 16-B store each.  Also: the same run split into two stores of R/2 issued one after the other by the same lanes
// ("halves": does the L2 merge them into one line write?), and a streaming read of 1.2 GB running beside the scatter.
// (TSC list build: why the scatter passes write whole 128-B lines.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// run r -> slot: a multiplicative hash permutation of [0, nruns) (nruns a power of two)
__device__ __forceinline__ uint64_t perm(uint64_t r, uint64_t mask) { return (r * 0x9E3779B97F4A7C15ull >> 20 ^ r * 2654435761ull) & mask; }

template <int R, int HALVES>
__global__ __launch_bounds__(256) void scatter(char *__restrict__ buf, uint64_t nruns, int shift, const float4 *__restrict__ src,
                                               int64_t nsrc, float *__restrict__ sink) {
    constexpr int LPR = R / 16;              // lanes per run
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * 256;
    float acc = 0.f;
    for (uint64_t i = t; i < nruns * LPR; i += nthreads) {
        const uint64_t run = i / LPR, lane = i % LPR;
        // a bijection: odd multiplier mod 2^k
        const uint64_t slot = (run * 0x9E3779B97F4A7C15ull) & (nruns - 1);
        float4 v = make_float4((float)i, 1.f, 2.f, 3.f);
        if (src) {   // a streaming read beside the scatter (what the pass reads): one float4 per 16-B written
            const float4 s = src[i % nsrc];
            v.y = s.x + s.y + s.z + s.w;
        }
        char *dst = buf + slot * R + shift;
        if (HALVES) {
            // lanes [0, LPR/2) write the first half now, then the same lanes write the second half: two partial-line stores
            if (lane < LPR / 2) {
                *reinterpret_cast<float4 *>(dst + lane * 16) = v;
                __builtin_amdgcn_s_waitcnt(0);   // keep them apart in the instruction stream
                *reinterpret_cast<float4 *>(dst + R / 2 + lane * 16) = v;
            }
        } else {
            *reinterpret_cast<float4 *>(dst + lane * 16) = v;
        }
        acc += v.y;
    }
    if (acc == 1.2345f) sink[0] = acc;
}

template <int R, int HALVES>
void run(char *buf, size_t bytes, int shift, const float4 *src, int64_t nsrc, float *sink, const char *tag) {
    uint64_t nruns = 1;
    while (nruns * 2 * R <= bytes - 4096) nruns *= 2;
    float best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        CK(hipEventRecord(a));
        scatter<R, HALVES><<<256 * 16, 256>>>(buf, nruns, shift, src, nsrc, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    const double gb = (double)nruns * R * (HALVES ? 1.0 : 1.0) / 1e9;
    printf("R %5d B  shift %3d  %-7s %-6s %8.3f ms  %7.1f GB written  %6.2f TB/s\n", R, shift, HALVES ? "halves" : "whole", tag, best, gb,
           gb / best);
    fflush(stdout);
}


// multisplit-like appends: workgroup w keeps B open regions (one per bucket) and appends RB bytes to each of them per
// round - the same line is completed by 128 / RB consecutive rounds of the same workgroup (does the L2 merge them?)
template <int RB>
__global__ __launch_bounds__(256) void append(char *__restrict__ buf, int B, int rounds, size_t region) {
    constexpr int LPR = RB / 16;
    const int g = threadIdx.x / LPR, lane = threadIdx.x % LPR, ng = 256 / LPR;
    for (int r = 0; r < rounds; r++)
        for (int b = g; b < B; b += ng) {
            char *dst = buf + ((size_t)blockIdx.x * B + b) * region + (size_t)r * RB + lane * 16;
            *reinterpret_cast<float4 *>(dst) = make_float4((float)r, 1.f, 2.f, 3.f);
        }
}
template <int RB>
void run_append(char *buf, size_t bytes, int B, int nwg) {
    const size_t region = (bytes / ((size_t)nwg * B)) & ~(size_t)1023;
    const int rounds = (int)(region / RB);
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        CK(hipEventRecord(a));
        append<RB><<<nwg, 256>>>(buf, B, rounds, region);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    const double gb = (double)nwg * B * rounds * RB / 1e9;
    printf("append %4d B per round, %5d buckets x %4d workgroups (region %6zu B): %8.3f ms  %6.2f TB/s\n", RB, B, nwg, region, best,
           gb / best);
    fflush(stdout);
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    char *buf;
    CK(hipMalloc(&buf, bytes + 4096));
    CK(hipMemset(buf, 0, bytes));
    float *sink;
    CK(hipMalloc(&sink, 64));
    const int64_t nsrc = 75000000;   // 1.2 GB
    float4 *src;
    CK(hipMalloc(&src, nsrc * 16));
    CK(hipMemset(src, 0, nsrc * 16));
    for (int withread = 0; withread < 2; withread++) {
        const float4 *s = withread ? src : nullptr;
        const char *tag = withread ? "+read" : "";
        run<16, 0>(buf, bytes, 0, s, nsrc, sink, tag);
        run<32, 0>(buf, bytes, 0, s, nsrc, sink, tag);
        run<64, 0>(buf, bytes, 0, s, nsrc, sink, tag);
        run<128, 0>(buf, bytes, 0, s, nsrc, sink, tag);
        run<256, 0>(buf, bytes, 0, s, nsrc, sink, tag);
        run<512, 0>(buf, bytes, 0, s, nsrc, sink, tag);
        run<1024, 0>(buf, bytes, 0, s, nsrc, sink, tag);
        run<128, 0>(buf, bytes, 32, s, nsrc, sink, tag);
        run<128, 0>(buf, bytes, 64, s, nsrc, sink, tag);
        run<256, 0>(buf, bytes, 64, s, nsrc, sink, tag);
        run<512, 0>(buf, bytes, 64, s, nsrc, sink, tag);
        run<128, 1>(buf, bytes, 0, s, nsrc, sink, tag);
        run<256, 1>(buf, bytes, 0, s, nsrc, sink, tag);
    }
    for (int nwg : {256, 512, 1024})
        for (int B : {128, 256, 512, 1024}) {
            run_append<32>(buf, bytes, B, nwg);
            run_append<64>(buf, bytes, B, nwg);
            run_append<128>(buf, bytes, B, nwg);
            run_append<256>(buf, bytes, B, nwg);
        }
    return 0;
}
