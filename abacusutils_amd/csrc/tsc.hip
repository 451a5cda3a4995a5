#include "../../include/abacus_hip.h"
#include "common.hpp"
using namespace abacus;
extern "C" {
int abacus_tsc_deposit(void *, int64_t, const void *, int, void *, int, int, int, int, double, double, int) { return fail("abacus_tsc_deposit: not built yet"); }
int abacus_tsc_deposit_dev(float *, int64_t, const float *, float *, int, int, int, double, double, int, int, int) { return fail("not built yet"); }
int abacus_cic_deposit(const void *, int64_t, const void *, int, float *, int, int, int, double) { return fail("not built yet"); }
int abacus_partition(const void *, int64_t, const void *, int, int, double, int, void *, int64_t *, void *) { return fail("not built yet"); }
}
