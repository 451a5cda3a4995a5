#include "../../include/abacus_hip.h"
#include "common.hpp"
using namespace abacus;
extern "C" {
int abacus_field_fft(float *, int64_t, const float *, double, int, int, const float *, int, void *) { return fail("not built yet"); }
int abacus_pk_from_deltak(const void *, const void *, int, double, const double *, int, const double *, int, const int64_t *, int, float *, int64_t *, float *, int64_t *, float *) { return fail("not built yet"); }
int abacus_power_from_particles(float *, int64_t, const float *, float *, int64_t, const float *, double, int, int, const float *, int, const double *, int, const double *, int, const int64_t *, int, float *, int64_t *, float *, int64_t *, float *) { return fail("not built yet"); }
int abacus_power_from_particles_dev(float *, int64_t, const float *, float *, int64_t, const float *, double, int, int, const float *, int, const double *, int, const double *, int, const int64_t *, int, float *, int64_t *, float *, int64_t *, float *) { return fail("not built yet"); }
int abacus_power_release(void) { return 0; }
}
