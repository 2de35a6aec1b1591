// stubs.hip — entry points declared in aps.h whose kernels are not written yet.  Each returns
// APS_E_INTERNAL with a clear message (never a silent fallback).  Shrinks as kernels land.
#include "aps_internal.h"
using namespace aps;
#define APS_STUB(name, ...) \
    int name(__VA_ARGS__) { return guarded([&] { fail(APS_E_INTERNAL, #name ": not implemented in this build"); }); }
extern "C" {
APS_STUB(aps_knn_global, const float*, int64_t, int64_t, const float*, int64_t, int64_t, int, int, int, uint32_t*, float*, int64_t)
APS_STUB(aps_global_filter, const uint32_t*, const float*, int64_t, int, int64_t, int, const uint32_t*, const uint32_t*, int, float, int64_t*, uint32_t*, uint32_t*, int64_t, int64_t*)
APS_STUB(aps_hamming_2nn, const uint8_t*, int64_t, int64_t, const uint8_t*, int64_t, int64_t, int, int, uint32_t*, float*, float*)
}
