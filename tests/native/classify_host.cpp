// Host build of abacusutils_amd/csrc/hod_classify.hpp (g++): lets the CPU suite fuzz the float32 interval classifier of the
// fused HOD kernel against the oracle's exact keep masks (tests/test_hod_classify.py).  Test infrastructure only.
#include <cmath>
#include <cstring>

#include "../../abacusutils_amd/csrc/hod_classify.hpp"

using namespace abacus_cls;

static SatPre make_pre(const abacus_hod_params *p) {   // as abacus_hod_populate_async builds it
    SatPre pre;
    memset(&pre, 0, sizeof pre);
    pre.L_const = p->L_Acent == 0 && p->L_Asat == 0 && p->L_Bcent == 0 && p->L_Bsat == 0;
    pre.E_const = p->E_Acent == 0 && p->E_Asat == 0 && p->E_Bcent == 0 && p->E_Bsat == 0 && p->E_Ccent == 0 && p->E_Csat == 0;
    pre.Q_const = p->Q_Acent == 0 && p->Q_Asat == 0 && p->Q_Bcent == 0 && p->Q_Bsat == 0;
    pre.L_M1 = pow(10.0, p->L_logM1), pre.L_Mcut = pow(10.0, p->L_logM_cut);
    pre.E_M1 = pow(10.0, p->E_logM1), pre.E_Mcut = pow(10.0, p->E_logM_cut);
    pre.E_M1_EL = pow(10.0, p->E_logM1_EL), pre.E_M1_EE = pow(10.0, p->E_logM1_EE);
    pre.Q_M1 = pow(10.0, p->Q_logM1), pre.Q_Mcut = pow(10.0, p->Q_logM_cut);
    return pre;
}

extern "C" {

void cls_cent(const abacus_hod_params *p, int64_t n, const double *mass, const double *multis, const double *randoms,
              const double *deltac, const double *fenv, const double *shear, int8_t *out) {
    ClsConst c;
    make_cls_const(*p, make_pre(p), c);
#pragma omp parallel for
    for (int64_t i = 0; i < n; i++)
        out[i] = (int8_t)cent_classify(c, mass[i], multis[i], randoms[i], deltac ? deltac[i] : 0.0, fenv ? fenv[i] : 0.0,
                                       (shear && p->want_ELG) ? shear[i] : 0.0);
}

void cls_sat(const abacus_hod_params *p, int64_t n, const double *hmass, const double *weights, const double *randoms,
             const double *d, const double *f, const double *sh, const double *rk, const double *rkv, const double *rkp,
             const double *rkr, const int8_t *keep_cent, int8_t *out) {
    ClsConst c;
    make_cls_const(*p, make_pre(p), c);
    const bool ranks = p->enable_ranks != 0;
#pragma omp parallel for
    for (int64_t i = 0; i < n; i++)
        out[i] = (int8_t)sat_classify(c, hmass[i], weights[i], randoms[i], d ? d[i] : 0.0, f ? f[i] : 0.0,
                                      (sh && p->want_ELG) ? sh[i] : 0.0, ranks ? rk[i] : 1.0, ranks ? rkv[i] : 1.0,
                                      ranks ? rkp[i] : 1.0, ranks ? rkr[i] : 1.0, keep_cent ? keep_cent[i] : 0);
}

}  // extern "C"
