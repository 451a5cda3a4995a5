// (k, mu) / multipole histogram arguments and the bin search shared by spectrum_bin (power.hip) and the fused last FFT
// pass + binning (xbin.hip).  Included at file scope.
#pragma once

namespace abacus {

constexpr int MAX_POLES = 8;     // requested multipoles

struct BinArgs {
    int Nk, Nmu, Np;          // Np = number of requested poles with ell != 0 (ell = 0 comes from the wedges)
    const float *kedges2;     // (Nk+1) f32((kedges/dk)^2)  (:217)
    const float *muedges2;    // (Nmu+1) f32(muedges^2)     (:218)
    const float *h_edges2;    // HOST copy: kedges2 then muedges2 (key of fft_x_bin's cached geometry descriptor), or null
    float polecoef[MAX_POLES][6];   // (2l+1) * P_l as a polynomial in mu^2: sum_m c[m] * (mu^2)^m
    int poledeg[MAX_POLES];         // l/2
    int dbg;                        // ablation switches (ABACUS_DBG): 1 skip binning, 2 skip staging
    unsigned long long *g_cnt;      // (Nk*Nmu)
    double *g_sum, *g_ksum;         // (Nk*Nmu)
    double *g_pole;                 // (Np*Nk)
};

// number of edges[1..N] strictly below v  ==  the bin the reference's `while v > edges[b+1]: b += 1` stops at
static __device__ __noinline__ int lower_bin(const float *edges, int N, float v) {
    int lo = 0, hi = N;   // answer in [lo, hi]
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (v > edges[mid + 1]) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

}  // namespace abacus
