// Shared runtime plumbing of libabacus_hip.so: error reporting, the library stream, device buffers and the
// per-kernel event profiler behind abacus_profile_* (include/abacus_hip.h).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>

namespace abacus {

int fail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int ensure_init();
hipStream_t stream();
// Scratch blocks for the per-call temporaries of the host-array entry points (prepare.hip, staging.hip): hipMalloc + hipFree per
// temporary cost those calls more than their kernels (abacus_prepare_randoms: 5 pairs, 7 ms of a 0.1-ms kernel).  A released
// block is kept for the next call; abacus_scratch_release() frees what is not in use.
int scratch_acquire(void **out, size_t bytes);
void scratch_release(void *p);
int scratch_trim_idle();
// the mesh buffers the P(k) entry points keep between calls (power.hip: up to four fields, 35 GB each at 2048^3; fft.hip: the second
// mesh of the out-of-place passes) hold nothing between calls: given back when an allocation of the caller's fails (abacus_malloc)
int power_trim_caches();
int fft_trim_scratch();

#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return ::abacus::fail("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);      \
    } while (0)

// Every extern "C" entry point starts with ABACUS_ENTER(): the library keeps one stream and reusable scratch
// allocations, so concurrent callers (ctypes releases the GIL) are serialised here; nested entry points re-enter.
std::recursive_mutex &api_mutex();
#define ABACUS_ENTER()                                                              \
    std::lock_guard<std::recursive_mutex> abacus_api_guard_(::abacus::api_mutex()); \
    ABACUS_TRY(::abacus::ensure_init())

#define ABACUS_TRY(expr)        \
    do {                        \
        int r_ = (expr);        \
        if (r_ != 0) return r_; \
    } while (0)

// Diagnostic options (abacus_set_option, include/abacus_hip.h): comparator paths the parity tests and the A/B scripts
// switch on explicitly - nothing in the library reads the environment.  0 when never set.
int option(const char *name);

// profiler hooks: no-ops unless abacus_profile_enable(1)
void prof_begin(const char *name);
void prof_end(const char *name);
bool prof_enabled();

// Launch a kernel on the library stream, bracketed by profiler events when profiling is on.
#define ABACUS_LAUNCH(name, kernel, grid, block, shmem, ...)                                     \
    do {                                                                                         \
        ::abacus::prof_begin(name);                                                              \
        kernel<<<grid, block, shmem, ::abacus::stream()>>>(__VA_ARGS__);                         \
        ::abacus::prof_end(name);                                                                \
        HIP_TRY(hipGetLastError());                                                              \
    } while (0)

// growable device allocation (never shrinks); contents are NOT preserved across grow()
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t nbytes) {
        if (nbytes <= cap) return 0;
        if (p) HIP_TRY(hipFree(p));
        p = nullptr;
        cap = 0;
        if (hipMalloc(&p, nbytes) != hipSuccess) {   // idle scratch blocks may be what stands in the way
            (void)hipGetLastError();
            p = nullptr;
            ABACUS_TRY(scratch_trim_idle());
            HIP_TRY(hipMalloc(&p, nbytes));
        }
        cap = nbytes;
        return 0;
    }
    int release() {
        if (p) HIP_TRY(hipFree(p));
        p = nullptr;
        cap = 0;
        return 0;
    }
    template <class T>
    T *as() const {
        return static_cast<T *>(p);
    }
};

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Orders the LDS traffic of ONE wave: its lanes' earlier LDS writes are visible to its lanes' later LDS reads (the LDS
// executes a wave's instructions in order; the fences keep the compiler from moving accesses across).  Lets a wave that
// owns a private LDS region work without workgroup barriers.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace abacus
