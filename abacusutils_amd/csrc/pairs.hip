#include "../../include/abacus_hip.h"
#include "common.hpp"
using namespace abacus;
extern "C" {
int abacus_paircount(int, const float *, const float *, const float *, int64_t, const float *, const float *, const float *, int64_t, float, const float *, int, float, int, float, int, uint64_t *) { return fail("not built yet"); }
}
