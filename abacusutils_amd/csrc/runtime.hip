// Runtime entry points of include/abacus_hip.h: device selection, the library stream, raw device memory,
// HIP-event timers and the per-kernel profiler.
#include <cstring>
#include <map>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/abacus_hip.h"
#include <algorithm>

#include "common.hpp"

namespace abacus {

static thread_local std::string g_err;
static int g_device = 0;
static bool g_inited = false;
static hipStream_t g_stream = nullptr;
static bool g_own_stream = false;
static std::mutex g_mu;

int fail(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}

std::recursive_mutex &api_mutex() {
    static std::recursive_mutex m;
    return m;
}

int ensure_init() {
    // HIP's current device is a per-thread setting: every thread that enters the library is bound to the library's device once
    // (a helper thread of a rank with LOCAL_RANK != 0 would otherwise work on device 0)
    static thread_local bool t_bound = false;
    if (g_inited) {
        if (!t_bound) {
            HIP_TRY(hipSetDevice(g_device));
            t_bound = true;
        }
        return 0;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_inited) return 0;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail("no HIP device available (%s); libabacus_hip.so has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    HIP_TRY(hipSetDevice(g_device));
    if (!g_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
        g_own_stream = true;
    }
    t_bound = true;
    g_inited = true;
    return 0;
}

hipStream_t stream() { return g_stream; }

namespace {
struct ScratchBlock {
    void *p;
    size_t cap;
    bool used;
};
std::vector<ScratchBlock> g_scratch_blocks;
size_t scratch_idle_bytes() {
    size_t b = 0;
    for (const ScratchBlock &k : g_scratch_blocks)
        if (!k.used) b += k.cap;
    return b;
}
int scratch_trim() {
    for (size_t i = 0; i < g_scratch_blocks.size();) {
        if (!g_scratch_blocks[i].used) {
            HIP_TRY(hipFree(g_scratch_blocks[i].p));
            g_scratch_blocks.erase(g_scratch_blocks.begin() + (long)i);
        } else i++;
    }
    return 0;
}
}  // namespace

// smallest idle block that holds `bytes` without wasting more than half of itself; otherwise a new allocation (idle blocks
// beyond 8 GiB are freed first)
int scratch_acquire(void **out, size_t bytes) {
    bytes = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
    int best = -1;
    for (size_t i = 0; i < g_scratch_blocks.size(); i++) {
        const ScratchBlock &k = g_scratch_blocks[i];
        if (!k.used && k.cap >= bytes && k.cap <= 2 * bytes + (1u << 20) && (best < 0 || k.cap < g_scratch_blocks[(size_t)best].cap)) best = (int)i;
    }
    if (best >= 0) {
        g_scratch_blocks[(size_t)best].used = true;
        *out = g_scratch_blocks[(size_t)best].p;
        return 0;
    }
    if (scratch_idle_bytes() > ((size_t)8 << 30)) ABACUS_TRY(scratch_trim());
    void *q = nullptr;
    if (hipMalloc(&q, bytes) != hipSuccess) {
        (void)hipGetLastError();
        ABACUS_TRY(scratch_trim());          // out of memory with idle blocks around: give them back and try once more
        HIP_TRY(hipMalloc(&q, bytes));
    }
    g_scratch_blocks.push_back(ScratchBlock{q, bytes, true});
    *out = q;
    return 0;
}
// Blocks above 1 GiB (a caller's whole mesh: 34 GB at 2048^3): at most ONE is kept idle - repeated tsc_parallel / get_field calls
// on host arrays take their device grid again and again (a hipMalloc, a hipFree and its device synchronise per call otherwise),
// while hipFFT work areas, RCCL buffers or another framework in the process must not run out of memory beside a pile of
// them.  scratch_acquire trims on memory pressure, abacus_power_release / abacus_scratch_release give the kept one back.
void scratch_release(void *p) {
    for (size_t i = 0; i < g_scratch_blocks.size(); i++) {
        ScratchBlock &k = g_scratch_blocks[i];
        if (k.p != p) continue;
        k.used = false;
        if (k.cap <= ((size_t)1 << 30)) return;
        for (size_t j = 0; j < g_scratch_blocks.size();) {       // the large block released last stays
            ScratchBlock &o = g_scratch_blocks[j];
            if (j != i && !o.used && o.cap > ((size_t)1 << 30)) {
                (void)hipFree(o.p);
                g_scratch_blocks.erase(g_scratch_blocks.begin() + (long)j);
                if (j < i) i--;
            } else j++;
        }
        return;
    }
}
int scratch_trim_idle() { return scratch_trim(); }

// ---- profiler -------------------------------------------------------------------------------------------
struct ProfEntry {
    double total_ms = 0;
    int64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};
static bool g_prof = false;
static std::map<std::string, ProfEntry> g_prof_map;
static std::vector<hipEvent_t> g_event_pool;
static hipEvent_t g_prof_open = nullptr;
static std::string g_prof_only;   // when set, only this kernel is bracketed (two event records cost a few us per launch)

static hipEvent_t pool_get() {
    if (!g_event_pool.empty()) {
        hipEvent_t e = g_event_pool.back();
        g_event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

bool prof_enabled() { return g_prof; }

void prof_begin(const char *name) {
    if (!g_prof) return;
    if (!g_prof_only.empty() && g_prof_only != name) return;
    g_prof_open = pool_get();
    (void)hipEventRecord(g_prof_open, g_stream);
}

void prof_end(const char *name) {
    if (!g_prof || !g_prof_open) return;
    hipEvent_t stop = pool_get();
    (void)hipEventRecord(stop, g_stream);
    g_prof_map[name].pending.emplace_back(g_prof_open, stop);
    g_prof_open = nullptr;
}

static void prof_drain() {
    for (auto &kv : g_prof_map) {
        for (auto &pr : kv.second.pending) {
            float ms = 0;
            if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                kv.second.total_ms += ms;
                kv.second.launches += 1;
            }
            g_event_pool.push_back(pr.first);
            g_event_pool.push_back(pr.second);
        }
        kv.second.pending.clear();
    }
}

namespace {
std::map<std::string, int> &option_table() {
    static std::map<std::string, int> t;
    return t;
}
}  // namespace

int option(const char *name) {
    std::lock_guard<std::recursive_mutex> guard(api_mutex());
    auto it = option_table().find(name);
    return it == option_table().end() ? 0 : it->second;
}

}  // namespace abacus

using namespace abacus;

extern "C" {

int abacus_set_option(const char *name, int value) {
    if (!name) return abacus::fail("abacus_set_option: null name");
    static const char *known[] = {"dbg", "dbg_fft", "dbg_tsc", "fft_fuse_small", "fft_hipfft", "fft_inplace", "gfft_dbg", "gfft_nofixed", "fft_nofuse", "hod_eblock", "hod_f64filter",
                                  "hod_nobalance", "hod_nocls", "hod_noindex", "hod_nokeys", "hod_nolazy", "hod_norec", "hod_one_stage", "hod_pipe", "hod_sbtiles", "pairs_countsort", "pairs_gen", "pairs_noblocks", "pairs_nolut", "pk_nobatch", "pk_noxbin", "pk_noxbin_cross", "pk_noxbin_inter", "pk_xbin_gen", "pk_xbin_pairs", "prep_columnwise", "slab_nocompact", "slab_nofuse", "slab_nopackfuse", "slab_nounpackfuse", "tsc_acc64", "tsc_atomic", "tsc_lines_cfg1", "tsc_lines_clk", "tsc_lines_sync", "tsc_noshare", "tsc_noslabfast", "tsc_oldlists"};
    bool ok = false;
    for (const char *k : known) ok = ok || !strcmp(k, name);
    if (!ok) return abacus::fail("abacus_set_option: unknown option '%s'", name);
    std::lock_guard<std::recursive_mutex> guard(abacus::api_mutex());
    abacus::option_table()[name] = value;
    return 0;
}

int abacus_get_option(const char *name) { return name ? abacus::option(name) : 0; }


const char *abacus_last_error(void) { return g_err.c_str(); }

int abacus_device_count(int *n) {
    HIP_TRY(hipGetDeviceCount(n));
    return 0;
}

int abacus_set_device(int device) {
    if (g_inited && device != g_device) return fail("abacus_set_device(%d) after initialisation on device %d", device, g_device);
    g_device = device;
    return 0;
}

int abacus_device_name(char *buf, int len) {
    ABACUS_ENTER();
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, g_device));
    snprintf(buf, len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}

int abacus_device_sync(void) {
    ABACUS_ENTER();
    HIP_TRY(hipStreamSynchronize(g_stream));
    return 0;
}

void *abacus_get_stream(void) {
    if (ensure_init() != 0) return nullptr;
    return (void *)g_stream;
}

int abacus_set_stream(void *s) {
    if (g_inited && g_own_stream && g_stream) {
        HIP_TRY(hipStreamSynchronize(g_stream));
        HIP_TRY(hipStreamDestroy(g_stream));
    }
    g_stream = (hipStream_t)s;
    g_own_stream = false;
    return ensure_init();
}

int abacus_malloc(void **dptr, uint64_t nbytes) {
    ABACUS_ENTER();
    if (hipMalloc(dptr, nbytes ? nbytes : 1) != hipSuccess) {
        // idle scratch blocks of this library (one mesh-sized block may be kept between calls, scratch_release) must not stand
        // between a caller and memory it asks for: give them back and try once more
        (void)hipGetLastError();
        ABACUS_TRY(scratch_trim());
        if (hipMalloc(dptr, nbytes ? nbytes : 1) != hipSuccess) {   // ... nor the P(k) context's idle meshes (140 GB after a 2048^3 interlaced cross power)
            (void)hipGetLastError();
            ABACUS_TRY(power_trim_caches());
            HIP_TRY(hipMalloc(dptr, nbytes ? nbytes : 1));
        }
    }
    return 0;
}
int abacus_free(void *dptr) {
    if (dptr) HIP_TRY(hipFree(dptr));
    return 0;
}
int abacus_memcpy_h2d(void *dst, const void *src, uint64_t nbytes) {
    ABACUS_ENTER();
    HIP_TRY(hipMemcpyAsync(dst, src, nbytes, hipMemcpyHostToDevice, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    return 0;
}
int abacus_memcpy_d2h(void *dst, const void *src, uint64_t nbytes) {
    ABACUS_ENTER();
    HIP_TRY(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToHost, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    return 0;
}
int abacus_memset(void *dptr, int value, uint64_t nbytes) {
    ABACUS_ENTER();
    HIP_TRY(hipMemsetAsync(dptr, value, nbytes, g_stream));
    return 0;
}

// page-locked host memory: a device-to-host copy into it is one DMA at link speed (a pageable destination is copied
// through a staging buffer, and a fresh NumPy allocation is page-faulted in on top of that)
int abacus_host_alloc(void **hptr, uint64_t nbytes) {
    ABACUS_ENTER();
    HIP_TRY(hipHostMalloc(hptr, nbytes ? nbytes : 1, hipHostMallocDefault));
    return 0;
}
int abacus_host_free(void *hptr) {
    if (hptr) HIP_TRY(hipHostFree(hptr));
    return 0;
}

namespace {
// sum over i of word[i] * (2 i + 1) mod 2^64: a position-dependent checksum of a column of 8-byte values (any in-place
// edit, a permutation of rows included, changes it)
__global__ void poshash_u64(const unsigned long long *__restrict__ a, int64_t n, unsigned long long *__restrict__ out) {
    unsigned long long acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        acc += a[i] * (2ull * (unsigned long long)i + 1ull);
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}
DevBuf g_hash_scratch;
}  // namespace

int abacus_poshash_u64(const void *dptr, int64_t n, uint64_t *out) {
    ABACUS_ENTER();
    if (!out || (n > 0 && !dptr)) return fail("abacus_poshash_u64: null argument");
    ABACUS_TRY(g_hash_scratch.reserve(8));
    HIP_TRY(hipMemsetAsync(g_hash_scratch.p, 0, 8, g_stream));
    if (n > 0) {
        const int grid = (int)std::min<int64_t>(std::max<int64_t>((n + 255) / 256, 1), 256 * 8);
        ABACUS_LAUNCH("poshash", poshash_u64, dim3(grid), dim3(256), 0, (const unsigned long long *)dptr, n,
                      g_hash_scratch.as<unsigned long long>());
    }
    HIP_TRY(hipMemcpyAsync(out, g_hash_scratch.p, 8, hipMemcpyDeviceToHost, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    return 0;
}

int abacus_event_create(void **ev) {
    ABACUS_ENTER();
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    *ev = (void *)e;
    return 0;
}
int abacus_event_record(void *ev) {
    HIP_TRY(hipEventRecord((hipEvent_t)ev, g_stream));
    return 0;
}
int abacus_event_elapsed_ms(void *start, void *stop, float *ms) {
    HIP_TRY(hipEventSynchronize((hipEvent_t)stop));
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return 0;
}
int abacus_event_destroy(void *ev) {
    HIP_TRY(hipEventDestroy((hipEvent_t)ev));
    return 0;
}

int abacus_profile_enable(int on) {
    ABACUS_ENTER();
    if (!on && g_prof) prof_drain();
    g_prof = on != 0;
    return 0;
}
int abacus_scratch_release(void) {
    ABACUS_ENTER();
    return scratch_trim();
}
int abacus_profile_select(const char *name) {
    g_prof_only = name ? name : "";
    return 0;
}

int abacus_profile_reset(void) {
    prof_drain();
    g_prof_map.clear();
    return 0;
}
int abacus_profile_get(const char **names, double *total_ms, int64_t *launches, int cap) {
    prof_drain();
    int i = 0;
    for (auto &kv : g_prof_map) {
        if (i < cap) {
            names[i] = kv.first.c_str();  // stable until the next abacus_profile_reset
            total_ms[i] = kv.second.total_ms;
            launches[i] = kv.second.launches;
        }
        i++;
    }
    return i;
}

}  // extern "C"
