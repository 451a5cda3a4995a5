// Exclusive prefix sum of 32-bit counters into 64-bit offsets, used for tile lists (tsc.hip) and cell lists
// (pairs.hip).  Three small launches (block sums, scan of block sums, local scan + offset); the counters are
// optionally reset to zero for a following fill pass.  n up to 2^31.
#include "common.hpp"

namespace abacus {

namespace {
constexpr int SB = 1024;             // threads per block
constexpr int ITEMS = 4;             // counters per thread
constexpr int CHUNK = SB * ITEMS;    // counters per block

__device__ __forceinline__ unsigned long long block_scan_incl(unsigned long long v, unsigned long long *wave_tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        unsigned long long t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    if (lane == 63) wave_tot[wave] = v;
    __syncthreads();
    if (wave == 0) {
        unsigned long long w = lane < SB / 64 ? wave_tot[lane] : 0;
#pragma unroll
        for (int off = 1; off < SB / 64; off <<= 1) {
            unsigned long long t = __shfl_up(w, off, 64);
            if (lane >= off) w += t;
        }
        if (lane < SB / 64) wave_tot[lane] = w;   // inclusive totals of waves 0..lane
    }
    __syncthreads();
    if (wave > 0) v += wave_tot[wave - 1];
    return v;
}

__global__ __launch_bounds__(SB) void scan_block_sums(const unsigned int *__restrict__ c, int64_t n,
                                                      unsigned long long *__restrict__ bsum) {
    __shared__ unsigned long long wt[SB / 64];
    const int64_t base = (int64_t)blockIdx.x * CHUNK + (int64_t)threadIdx.x * ITEMS;
    unsigned long long s = 0;
#pragma unroll
    for (int q = 0; q < ITEMS; q++)
        if (base + q < n) s += c[base + q];
    s = block_scan_incl(s, wt);
    if (threadIdx.x == SB - 1) bsum[blockIdx.x] = s;
}

__global__ __launch_bounds__(SB) void scan_top(unsigned long long *__restrict__ bsum, int nblocks,
                                               int64_t *__restrict__ total_out) {
    __shared__ unsigned long long wt[SB / 64];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblocks; b0 += SB) {
        const int i = b0 + threadIdx.x;
        const unsigned long long v = i < nblocks ? bsum[i] : 0;
        const unsigned long long incl = block_scan_incl(v, wt);
        const unsigned long long c0 = carry;
        if (i < nblocks) bsum[i] = c0 + incl - v;   // exclusive
        __syncthreads();
        if (threadIdx.x == SB - 1) carry = c0 + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = (int64_t)carry;
}

__global__ __launch_bounds__(SB) void scan_apply(unsigned int *__restrict__ c, int64_t n,
                                                 const unsigned long long *__restrict__ bsum,
                                                 int64_t *__restrict__ out, int zero_counters) {
    __shared__ unsigned long long wt[SB / 64];
    const int64_t base = (int64_t)blockIdx.x * CHUNK + (int64_t)threadIdx.x * ITEMS;
    unsigned int v[ITEMS];
    unsigned long long s = 0;
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {
        v[q] = base + q < n ? c[base + q] : 0u;
        s += v[q];
    }
    unsigned long long run = block_scan_incl(s, wt) - s + bsum[blockIdx.x];
#pragma unroll
    for (int q = 0; q < ITEMS; q++)
        if (base + q < n) {
            out[base + q] = (int64_t)run;
            run += v[q];
            if (zero_counters) c[base + q] = 0u;
        }
}
}  // namespace

// out[i] = sum_{j<i} counters[j] for i in [0, n]; out[n] = total.  `scratch` must hold ceil(n/4096)+1 u64.
int exclusive_scan_u32(unsigned int *counters, int64_t n, int64_t *out, DevBuf &scratch, int zero_counters) {
    const int nblocks = (int)ceil_div(n > 0 ? n : 1, CHUNK);
    ABACUS_TRY(scratch.reserve((size_t)(nblocks + 1) * sizeof(unsigned long long)));
    unsigned long long *bsum = scratch.as<unsigned long long>();
    ABACUS_LAUNCH("scan_block_sums", scan_block_sums, dim3(nblocks), dim3(SB), 0, counters, n, bsum);
    ABACUS_LAUNCH("scan_top", scan_top, dim3(1), dim3(SB), 0, bsum, nblocks, out + n);
    ABACUS_LAUNCH("scan_apply", scan_apply, dim3(nblocks), dim3(SB), 0, counters, n, bsum, out, zero_counters);
    return 0;
}

}  // namespace abacus
