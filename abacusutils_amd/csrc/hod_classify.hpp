// Float32 INTERVAL evaluation of the HOD marker chains (gen_cent pass 1, hod/GRAND_HOD.py:213-252; gen_sats pass 1,
// :957-1088): every marker of an object is enclosed in [lo, hi] by float32 arithmetic with outward slack, and the
// reference's decision `random <= marker` is taken from the enclosure whenever the random lies outside every band;
// only the objects whose random falls INSIDE a band (a few 1e-4 of the candidates) are evaluated with the reference's
// float64 erfc / erf / exp / log10 / pow chain.  The enclosure must contain the value the REFERENCE COMPUTES, not the
// mathematical one: `0.5 * (1 + erf(u))` (N_cen_QSO, the ELG Phi) carries the absolute rounding error of `1 + erf(u)`,
// which is added to its band.
//
// Compiled by hipcc into hod.hip (device) and by g++ into tests/native/classify_host.cpp (host), so that the logic is
// fuzzed on the CPU against the oracle's exact decisions; the libm / ocml float functions only need to be accurate to
// the 1e-5 relative slack they are given here (both are within a few ulp = 1e-6).
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/abacus_hip.h"

#if defined(__HIPCC__)
#define CLS_HD __host__ __device__ __forceinline__
#else
#define CLS_HD static inline
#endif

namespace abacus_cls {

// 10**x of values that do not depend on the particle (all assembly-bias coefficients of a tracer zero:
// `logM1 + 0*d + 0*f == logM1` exactly), evaluated once on the host with libm's pow
struct SatPre {
    int L_const, E_const, Q_const;
    int pad;
    double L_M1, L_Mcut, E_M1, E_Mcut, E_M1_EL, E_M1_EE, Q_M1, Q_Mcut;
};

struct Iv {
    float lo, hi;
};

constexpr float CLS_R = 2.5e-7f;     // outward slack per float32 operation (one rounding is 6e-8)
constexpr float CLS_F = 1e-5f;       // relative slack of erfcf / expf / exp10f / powf results
constexpr float CLS_TINY = 1e-36f;

// exp / exp10 / log10 / pow: on the device the hardware's exp2 and log2 (v_exp_f32, v_log_f32: 1 ulp) instead of the ocml
// library functions (25 - 100 instructions each: two thirds of the satellite classifier).  Relative errors, all inside the
// slack the enclosures already carry: exp10(x) |x| 2.3e-7 + 1 ulp (|x| <= 17: 4e-6 of the 2e-5 allotted); exp(-a) a 1e-7
// (a <= 87: 9e-6 of 2e-5); log10 1.5e-7 |l| absolute (1e-6 |l| allotted); pow(y, a) a |log2 y| 9e-8 (a |log2 y| 6.9e-7 allotted).
#if defined(__HIP_DEVICE_COMPILE__)
CLS_HD float cls_exp10(float x) { return __builtin_amdgcn_exp2f(x * 3.3219281f); }
CLS_HD float cls_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950f); }
CLS_HD float cls_log10(float x) { return __builtin_amdgcn_logf(x) * 0.30103000f; }
CLS_HD float cls_pow(float y, float a) { return __builtin_amdgcn_exp2f(a * __builtin_amdgcn_logf(y)); }
#else
CLS_HD float cls_exp10(float x) { return exp10f(x); }
CLS_HD float cls_exp(float x) { return expf(x); }
CLS_HD float cls_log10(float x) { return log10f(x); }
CLS_HD float cls_pow(float y, float a) { return powf(y, a); }
#endif

CLS_HD Iv iv(float lo, float hi) {
    Iv r;
    r.lo = lo, r.hi = hi;
    return r;
}
CLS_HD Iv widen(Iv a) { return iv(a.lo - fabsf(a.lo) * CLS_R - CLS_TINY, a.hi + fabsf(a.hi) * CLS_R + CLS_TINY); }
CLS_HD Iv from_double(double x) {   // the float nearest to x is within 6e-8 |x|
    const float f = (float)x;
    return iv(f - fabsf(f) * CLS_R, f + fabsf(f) * CLS_R);
}
CLS_HD Iv iadd(Iv a, Iv b) { return widen(iv(a.lo + b.lo, a.hi + b.hi)); }
CLS_HD Iv isub(Iv a, Iv b) { return widen(iv(a.lo - b.hi, a.hi - b.lo)); }
CLS_HD Iv imul(Iv a, Iv b) {
    const float p0 = a.lo * b.lo, p1 = a.lo * b.hi, p2 = a.hi * b.lo, p3 = a.hi * b.hi;
    return widen(iv(fminf(fminf(p0, p1), fminf(p2, p3)), fmaxf(fmaxf(p0, p1), fmaxf(p2, p3))));
}
// product of two quantities that are non-negative by construction (occupation factors, weights, powers): two multiplies
// instead of four and their min / max (a lower end that the widening pushed below zero counts as zero)
CLS_HD Iv imul_pos(Iv a, Iv b) { return widen(iv(fmaxf(a.lo, 0.f) * fmaxf(b.lo, 0.f), a.hi * b.hi)); }
CLS_HD Iv iscale(Iv a, float c) {   // c: a float constant known to 6e-8 (its error is inside the widening)
    return widen(c >= 0.f ? iv(a.lo * c, a.hi * c) : iv(a.hi * c, a.lo * c));
}
// base + A d + B f + C s: float32 coefficients (host-converted from the float64 parameters: 6e-8 each) and float64
// per-object values, as a float32 value with an error bound
CLS_HD Iv affine(float b, float A, double d, float B, double f, float C, double s) {
    const float t1 = A * (float)d, t2 = B * (float)f, t3 = C * (float)s;
    const float v = ((b + t1) + t2) + t3;
    const float e = 1e-6f * (fabsf(b) + fabsf(t1) + fabsf(t2) + fabsf(t3)) + CLS_TINY;
    return iv(v - e, v + e);
}
CLS_HD Iv ilog10(double M) {   // log10 of a positive float64; non-positive / NaN masses give a NaN band (-> exact path)
    const float l = cls_log10((float)M);
    const float e = fabsf(l) * 1e-6f + 2e-7f;
    return iv(l - e, l + e);
}
// The transcendental of an enclosure is evaluated ONCE, at the end that gives the upper value; the lower value follows
// from a derivative bound over the (1e-6-wide) argument band - half the erfcf / exp10f / expf / powf calls, which are
// most of the classifier's instructions.
//
// 0.5 * erfc(t), decreasing in t; `abs_slack`: absolute error of the reference's own float64 evaluation
// (0 for the erfc form, 3e-16 for 0.5 * (1 + erf(u))).  Lower end: erfc(t.lo) - erfc(t.hi) = |erfc'(x)| (t.hi - t.lo)
// for some x in the band, and |erfc'(x)| = 2/sqrt(pi) exp(-x^2) <= erfc(x) (2 max(x, 0) + 1.5) <= erfc(t.lo) (2 max(t.hi, 0) + 1.5)
// (x >= 0: the Mills-ratio bound erfc(x) > 2/sqrt(pi) exp(-x^2) / (x + sqrt(x^2 + 2)); x < 0: erfc >= 1, |erfc'| <= 1.13)
CLS_HD Iv half_erfc(Iv t, float abs_slack) {
    const float hi = 0.5f * erfcf(t.lo);
    const float w = (t.hi - t.lo) * (1.f + 1e-6f) + CLS_TINY;
    const float lo = hi * fmaxf(1.f - w * (2.f * fmaxf(t.hi, 0.f) + 1.5f), 0.f);
    return iv(fmaxf(lo * (1.f - CLS_F) - abs_slack - CLS_TINY, 0.f), hi * (1.f + CLS_F) + abs_slack + CLS_TINY);
}
// 10**x: 10**x.lo = 10**x.hi * 10**-(x.hi - x.lo) >= 10**x.hi * (1 - ln(10) (x.hi - x.lo))
CLS_HD Iv iexp10(Iv x) {
    const float hi = cls_exp10(x.hi);
    const float w = (x.hi - x.lo) * (1.f + 1e-6f) + CLS_TINY;
    return iv(hi * fmaxf(1.f - 2.302586f * w, 0.f) * (1.f - 2.f * CLS_F), hi * (1.f + 2.f * CLS_F) + CLS_TINY);
}
// exp(-a): a in [alo, ahi], 0 <= alo; exp(-ahi) = exp(-alo) exp(-(ahi - alo)) >= exp(-alo) (1 - (ahi - alo))
CLS_HD Iv iexp_neg(float alo, float ahi) {
    const float hi = cls_exp(-alo);
    const float w = (ahi - alo) * (1.f + 1e-6f) + CLS_TINY;
    return iv(hi * fmaxf(1.f - w, 0.f) * (1.f - 2.f * CLS_F), hi * (1.f + 2.f * CLS_F) + CLS_TINY);
}
CLS_HD Iv fconst(float c) { return iv(c - fabsf(c) * CLS_R, c + fabsf(c) * CLS_R); }   // a host-converted float64 constant

// Every parameter the classifier reads, converted to float32 ON THE HOST: kernel arguments live in scalar registers,
// whereas a float64 -> float32 conversion inside the kernel is a vector instruction whose (uniform) result occupies a
// vector register for the whole candidate loop - sixty of them cost the kernel half its occupancy.
struct ClsTracer {
    float lc0, Ac, Bc, Cc;          // logM_cut' = lc0 + Ac deltac + Bc fenv + Cc shear
    float l10, As, Bs, Cs;          // logM1'    = l10 + As deltac + Bs fenv + Cs shear
    float inv_s, ic, kappa, alpha;  // 1 / (1.41421356 sigma), incompleteness, kappa, alpha
    float s[4];                     // s, s_v, s_p, s_r
    float M1, Mcut;                 // particle-independent 10**logM1, 10**logM_cut (SatPre), used when `is_const`
    int is_const, alpha_is_one;
};
struct ClsConst {
    ClsTracer L, E, Q;
    float E_K, E_h, E_gs, E_As;                   // 2 (p_max - 1/Q) 0.39894.../sigma * ic;  1 / (2 sigma^2);  gamma / sigma / sqrt(2);  A_s
    float E_l10_EL, E_l10_EE, E_alpha_EL, E_alpha_EE, E_M1_EL, E_M1_EE;
    int E_alpha_EL_is_one, E_alpha_EE_is_one;
    int want_LRG, want_ELG, want_QSO, enable_ranks;
};

static inline void make_cls_const(const abacus_hod_params &p, const SatPre &pre, ClsConst &c) {
    auto tr = [](ClsTracer &t, double lc0, double Ac, double Bc, double Cc, double l10, double As, double Bs, double Cs,
                 double sigma, double ic, double kappa, double alpha, double s0, double s1, double s2, double s3, double M1,
                 double Mcut, int is_const) {
        t.lc0 = (float)lc0, t.Ac = (float)Ac, t.Bc = (float)Bc, t.Cc = (float)Cc;
        t.l10 = (float)l10, t.As = (float)As, t.Bs = (float)Bs, t.Cs = (float)Cs;
        t.inv_s = (float)(1.0 / (1.41421356 * sigma)), t.ic = (float)ic, t.kappa = (float)kappa, t.alpha = (float)alpha;
        t.s[0] = (float)s0, t.s[1] = (float)s1, t.s[2] = (float)s2, t.s[3] = (float)s3;
        t.M1 = (float)M1, t.Mcut = (float)Mcut, t.is_const = is_const, t.alpha_is_one = alpha == 1.0;
    };
    tr(c.L, p.L_logM_cut, p.L_Acent, p.L_Bcent, 0.0, p.L_logM1, p.L_Asat, p.L_Bsat, 0.0, p.L_sigma, p.L_ic, p.L_kappa,
       p.L_alpha, p.L_s, p.L_s_v, p.L_s_p, p.L_s_r, pre.L_M1, pre.L_Mcut, pre.L_const);
    tr(c.E, p.E_logM_cut, p.E_Acent, p.E_Bcent, p.E_Ccent, p.E_logM1, p.E_Asat, p.E_Bsat, p.E_Csat, p.E_sigma, p.E_ic,
       p.E_kappa, p.E_alpha, p.E_s, p.E_s_v, p.E_s_p, p.E_s_r, pre.E_M1, pre.E_Mcut, pre.E_const);
    tr(c.Q, p.Q_logM_cut, p.Q_Acent, p.Q_Bcent, 0.0, p.Q_logM1, p.Q_Asat, p.Q_Bsat, 0.0, p.Q_sigma, p.Q_ic, p.Q_kappa,
       p.Q_alpha, p.Q_s, p.Q_s_v, p.Q_s_p, p.Q_s_r, pre.Q_M1, pre.Q_Mcut, pre.Q_const);
    c.E_K = (float)(2.0 * (p.E_p_max - 1.0 / p.E_Q) * 0.3989422804014327 / p.E_sigma * p.E_ic);
    c.E_h = (float)(0.5 / (p.E_sigma * p.E_sigma));
    c.E_gs = (float)(p.E_gamma / p.E_sigma / 1.4142135623730951);
    c.E_As = (float)p.E_A_s;
    c.E_l10_EL = (float)p.E_logM1_EL, c.E_l10_EE = (float)p.E_logM1_EE;
    c.E_alpha_EL = (float)p.E_alpha_EL, c.E_alpha_EE = (float)p.E_alpha_EE;
    c.E_M1_EL = (float)pre.E_M1_EL, c.E_M1_EE = (float)pre.E_M1_EE;
    c.E_alpha_EL_is_one = p.E_alpha_EL == 1.0, c.E_alpha_EE_is_one = p.E_alpha_EE == 1.0;
    c.want_LRG = p.want_LRG, c.want_ELG = p.want_ELG, c.want_QSO = p.want_QSO, c.enable_ranks = p.enable_ranks;
}

// decision of the chain `r <= m1 -> 1; r <= m2 -> 2; r <= m3 -> 3; else 0` from enclosures; -1 = inside a band
CLS_HD int pick_iv(double r, Iv m1, Iv m2, Iv m3) {
    if (r <= (double)m1.lo) return 1;
    if (!(r > (double)m1.hi)) return -1;
    if (r <= (double)m2.lo) return 2;
    if (!(r > (double)m2.hi)) return -1;
    if (r <= (double)m3.lo) return 3;
    if (!(r > (double)m3.hi)) return -1;
    return 0;
}

// centrals: cent_decide (hod.hip) = hod/GRAND_HOD.py:213-252
CLS_HD int cent_classify(const ClsConst &c, double mass, double multis, double r, double deltac, double fenv, double shear) {
    const Iv lM = ilog10(mass), mu = from_double(multis);
    Iv m1 = iv(0.f, 0.f);
    if (c.want_LRG) {
        const Iv lc = affine(c.L.lc0, c.L.Ac, deltac, c.L.Bc, fenv, 0.f, 0.0);
        const Iv t = iscale(isub(lc, lM), c.L.inv_s);
        m1 = imul_pos(iscale(half_erfc(t, 0.f), c.L.ic), mu);
    }
    Iv m2 = m1;
    if (c.want_ELG) {
        const Iv lc = affine(c.E.lc0, c.E.Ac, deltac, c.E.Bc, fenv, c.E.Cc, shear);
        const Iv d = isub(lM, lc);
        const float alo = (d.lo <= 0.f && d.hi >= 0.f) ? 0.f : fminf(fabsf(d.lo), fabsf(d.hi)), ahi = fmaxf(fabsf(d.lo), fabsf(d.hi));
        const float h = c.E_h;
        const Iv phi = iexp_neg((alo * alo) * h * (1.f - 4.f * CLS_R), (ahi * ahi) * h * (1.f + 4.f * CLS_R));
        const Iv y = iscale(d, c.E_gs);                              // gamma (logM - logM_cut) / sigma / sqrt(2)
        const Iv Phi = half_erfc(iv(-y.hi, -y.lo), 3e-16f);          // 0.5 (1 + erf(y)) = 0.5 erfc(-y)
        m2 = iadd(m1, imul(iscale(imul_pos(phi, Phi), c.E_K), mu));      // E_K = 2 (p_max - 1/Q) ...: sign not guaranteed
    }
    Iv m3 = m2;
    if (c.want_QSO) {
        const Iv lc = affine(c.Q.lc0, c.Q.Ac, deltac, c.Q.Bc, fenv, 0.f, 0.0);
        const Iv t = iscale(isub(lc, lM), c.Q.inv_s);
        m3 = iadd(m2, imul_pos(iscale(half_erfc(t, 3e-16f), c.Q.ic), mu));
    }
    return pick_iv(r, m1, m2, m3);
}

// ((M - kappa M_cut) / M1)^alpha, 0 when M - kappa M_cut < 0 (n_sat_LRG_modified :28-29, N_sat_generic :48-49);
// `ok` is cleared when the enclosure cannot be formed (alpha < 0)
CLS_HD Iv plaw_iv(Iv M, float kappa, Iv Mcut, Iv M1, float alpha, int alpha_is_one, bool &ok) {
    const Iv x = isub(M, imul(fconst(kappa), Mcut));
    if (x.hi < 0.f) return iv(0.f, 0.f);
    const float ylo = fmaxf(x.lo, 0.f) / M1.hi * (1.f - CLS_R), yhi = x.hi / M1.lo * (1.f + CLS_R) + CLS_TINY;
    if (alpha_is_one) return iv(x.lo < 0.f ? 0.f : ylo, yhi);
    if (!(alpha >= 0.f)) {
        ok = false;
        return iv(0.f, 0.f);
    }
    // powf of a value known to 3e-7 with an exponent known to 6e-8: relative error alpha * 3e-7 + |ln y| * 6e-8 * alpha,
    // |ln y| <= ln(2) (|binary exponent of y| + 1).  One powf, at yhi: ylo^alpha = yhi^alpha (1 - u)^alpha with
    // u = 1 - ylo / yhi in [0, 1], and (1 - u)^alpha >= 1 - max(alpha, 1) u (Bernoulli for alpha >= 1; (1 - u)^alpha >= 1 - u below)
    int e2;
    (void)frexpf(fmaxf(yhi, 1e-30f), &e2);
    const float s = 2.f * CLS_F + alpha * 1e-6f * (1.f + 0.6932f * (float)(abs(e2) + 1));
    const float v = cls_pow(yhi, alpha);
    if (x.lo < 0.f) return iv(0.f, v * (1.f + s) + CLS_TINY);
    const float u = (yhi - ylo) / yhi * (1.f + 1e-6f) + 1e-7f;
    return iv(v * fmaxf(1.f - fmaxf(alpha, 1.f) * u, 0.f) * (1.f - s), v * (1.f + s) + CLS_TINY);
}

CLS_HD Iv dec_iv(const float s[4], double r, double rv, double rp, double rr) {
    const float t0 = s[0] * (float)r, t1 = s[1] * (float)rv, t2 = s[2] * (float)rp, t3 = s[3] * (float)rr;
    const float v = (((1.f + t0) + t1) + t2) + t3;
    const float e = 1e-6f * (1.f + fabsf(t0) + fabsf(t1) + fabsf(t2) + fabsf(t3));
    return iv(v - e, v + e);
}

// satellites: sat_decide (hod.hip) = hod/GRAND_HOD.py:957-1088
CLS_HD int sat_classify(const ClsConst &c, double hmass, double weights, double r, double d, double f, double sh, double rk,
                        double rkv, double rkp, double rkr, int keep_cent) {
    bool ok = true;
    const Iv w = from_double(weights), M = from_double(hmass);
    Iv m1 = iv(0.f, 0.f);
    if (c.want_LRG) {
        const ClsTracer &T = c.L;
        const Iv lc = affine(T.lc0, T.Ac, d, T.Bc, f, 0.f, 0.0);
        const Iv M1 = T.is_const ? fconst(T.M1) : iexp10(affine(T.l10, T.As, d, T.Bs, f, 0.f, 0.0));
        const Iv Mcut = T.is_const ? fconst(T.Mcut) : iexp10(lc);
        const Iv t = iscale(isub(lc, ilog10(hmass)), T.inv_s);
        Iv term = imul_pos(imul_pos(plaw_iv(M, T.kappa, Mcut, M1, T.alpha, T.alpha_is_one, ok), half_erfc(t, 0.f)), iscale(w, T.ic));
        if (c.enable_ranks) term = imul(term, dec_iv(T.s, rk, rkv, rkp, rkr));
        m1 = term;
    }
    Iv m2 = m1;
    if (c.want_ELG) {
        const ClsTracer &T = c.E;
        const Iv lc = affine(T.lc0, T.Ac, d, T.Bc, f, T.Cc, sh);
        float alpha = T.alpha;
        int a1 = T.alpha_is_one;
        Iv M1;
        if (keep_cent == 1) {   // ELG conformity (:1006-1035); these branches carry no Csat term
            M1 = T.is_const ? fconst(c.E_M1_EL) : iexp10(affine(c.E_l10_EL, T.As, d, T.Bs, f, 0.f, 0.0));
            alpha = c.E_alpha_EL, a1 = c.E_alpha_EL_is_one;
        } else if (keep_cent == 2) {
            M1 = T.is_const ? fconst(c.E_M1_EE) : iexp10(affine(c.E_l10_EE, T.As, d, T.Bs, f, 0.f, 0.0));
            alpha = c.E_alpha_EE, a1 = c.E_alpha_EE_is_one;
        } else {
            M1 = T.is_const ? fconst(T.M1) : iexp10(affine(T.l10, T.As, d, T.Bs, f, T.Cs, sh));
        }
        const Iv Mcut = T.is_const ? fconst(T.Mcut) : iexp10(lc);
        Iv term = imul(imul(plaw_iv(M, T.kappa, Mcut, M1, alpha, a1, ok), fconst(c.E_As)), iscale(w, T.ic));
        if (c.enable_ranks) term = imul(term, dec_iv(T.s, rk, rkv, rkp, rkr));
        m2 = iadd(m1, term);
    }
    Iv m3 = m2;
    if (c.want_QSO) {
        const ClsTracer &T = c.Q;
        const Iv lc = affine(T.lc0, T.Ac, d, T.Bc, f, 0.f, 0.0);
        const Iv M1 = T.is_const ? fconst(T.M1) : iexp10(affine(T.l10, T.As, d, T.Bs, f, 0.f, 0.0));
        const Iv Mcut = T.is_const ? fconst(T.Mcut) : iexp10(lc);
        Iv term = imul_pos(plaw_iv(M, T.kappa, Mcut, M1, T.alpha, T.alpha_is_one, ok), iscale(w, T.ic));
        if (c.enable_ranks) term = imul(term, dec_iv(T.s, rk, rkv, rkp, rkr));
        m3 = iadd(m2, term);
    }
    if (!ok) return -1;
    return pick_iv(r, m1, m2, m3);
}

}  // namespace abacus_cls
