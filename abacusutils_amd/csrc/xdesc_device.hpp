// The cached geometry descriptor of a (mesh size, k edges, mu edges) set - which bin a mode falls into, N_mode and sum |k| per
// bin - as the kernels that bin straight from LDS read it (xbin.hip: fft_x_bin2 for the power-of-two meshes; gfft.hip:
// gfft_x_bin for the mixed-radix ones).  Built and validated over every mode of the mesh by xbin.hip (xdesc_get).
// Included at file scope inside the including file's anonymous namespace.
#pragma once

struct XDesc {
    const unsigned int *lut;   // (ncell) eb << 22 | min(T[eb], 2^22 - 1)
    const int *U;              // (kzlen, ustride): largest kmag2 with mu2 > muedges2[m + 1], m = 0 .. Nmu-2; -1: none
    int ncell, sh, off, ustride;
    int vtop;                  // T[Nk]: every kmag2 above it lies beyond the last edge
    const unsigned long long *cnt;   // (Nk * Nmu) N_mode
    const double *ksum;              // (Nk * Nmu) sum of w * sqrt(kmag2)
};

constexpr int XD_USTRIDE = 8;            // mu thresholds kept per kz (Nmu <= 8 on this path)
constexpr int XD_TMASK = 0x3fffff;

// vf1 = max(f32(kmag2), 1): kmag2 = 0 (the DC mode) shares the cell of kmag2 = 1; lut0 = lut - off; the table covers every
// kmag2 of the mesh, so the index needs no clamp
__device__ __forceinline__ int xd_eb(const unsigned int *lut0, int sh, int v, float vf1) {
    const unsigned int w = lut0[__float_as_uint(vf1) >> sh];
    return (int)(w >> 22) + (v > (int)(w & XD_TMASK) ? 1 : 0);
}

