// Counter-based random numbers shared by hod.hip (`reseed`) and prepare.hip (`prepare_sim`'s draws on the device).
// Philox4x32-10: the value of an object depends only on (seed, stream, object index), never on the launch geometry, so a
// catalogue sharded over GPUs draws the same numbers as the unsharded one when the caller passes its global index offset.
// Pinned by the generator's published known-answer vectors through the oracle's restatement (tests/test_oracle_reseed.py).
#pragma once

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned int hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const unsigned int hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}
__device__ __forceinline__ float u01(unsigned int x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }   // [0, 1)

// Transforms of the uniforms, REPRODUCIBLE bit for bit on any IEEE-754 machine (the CPU oracle restates them,
// oracle/abacus_oracle.c, and tests/test_reseed_gpu.py holds the device stream to it): float64 + - * / sqrt only, in a
// fixed order (-ffp-contract=off), no library log / sin / cos whose last bits differ between libm and ocml.  Accuracy
// ~1e-12, far inside the float32 the draws are rounded to.
//   det_log(x), x > 0 finite: x = m 2^e with m in [sqrt(1/2), sqrt 2); log m = 2 atanh(s), s = (m-1)/(m+1), odd series to s^15
__device__ __forceinline__ double det_log(double x) {
    long long bits = __double_as_longlong(x);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    double m = __longlong_as_double((bits & 0x000fffffffffffffll) | 0x3ff0000000000000ll);   // [1, 2)
    if (m > 1.4142135623730951) m *= 0.5, e += 1;
    const double s = (m - 1.0) / (m + 1.0), z = s * s;
    double p = 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    return 2.0 * s * p + (double)e * 0.6931471805599453;
}
//   sin and cos of 2 pi t, t in [0, 1): quadrant q = floor(4 t), angle a = (4 t - q) pi/2 in [0, pi/2), Taylor series to a^19 / a^18
__device__ __forceinline__ void det_sincos2pi(double t, double *sn, double *cs) {
    const double t4 = 4.0 * t;
    const int q = (int)t4;
    const double a = (t4 - (double)q) * 1.5707963267948966, z = a * a;
    double ps = -1.0 / 121645100408832000.0;   // -1/19!
    ps = ps * z + 1.0 / 355687428096000.0;     //  1/17!
    ps = ps * z - 1.0 / 1307674368000.0;       // -1/15!
    ps = ps * z + 1.0 / 6227020800.0;          //  1/13!
    ps = ps * z - 1.0 / 39916800.0;            // -1/11!
    ps = ps * z + 1.0 / 362880.0;              //  1/9!
    ps = ps * z - 1.0 / 5040.0;                // -1/7!
    ps = ps * z + 1.0 / 120.0;                 //  1/5!
    ps = ps * z - 1.0 / 6.0;                   // -1/3!
    ps = ps * z + 1.0;
    const double s0 = a * ps;
    double pc = 1.0 / 6402373705728000.0;      //  1/18!
    pc = pc * z - 1.0 / 20922789888000.0;      // -1/16!
    pc = pc * z + 1.0 / 87178291200.0;         //  1/14!
    pc = pc * z - 1.0 / 479001600.0;           // -1/12!
    pc = pc * z + 1.0 / 3628800.0;             //  1/10!
    pc = pc * z - 1.0 / 40320.0;               // -1/8!
    pc = pc * z + 1.0 / 720.0;                 //  1/6!
    pc = pc * z - 1.0 / 24.0;                  // -1/4!
    pc = pc * z + 0.5;                         //  1/2!
    const double c0 = 1.0 - z * pc;
    switch (q & 3) {
        case 0: *sn = s0, *cs = c0; break;
        case 1: *sn = c0, *cs = -s0; break;
        case 2: *sn = -s0, *cs = -c0; break;
        default: *sn = -c0, *cs = s0; break;
    }
}
// 53-bit uniform in [0, 1) from two words (the construction NumPy's generators use for float64)
__device__ __forceinline__ double u53(unsigned int a, unsigned int b) {
    return (double)(((unsigned long long)(a >> 5) << 26) | (unsigned long long)(b >> 6)) * 1.1102230246251565e-16;
}
