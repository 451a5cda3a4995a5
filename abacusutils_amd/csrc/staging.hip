// Device side of AbacusHOD.staging() (abacusnbody/hod/abacus_hod.py:253-704), the step in front of the population
// kernels.  Three pieces of it are data-parallel work on the catalogue rather than file I/O:
//   abacus_argsort_i64      the halo sort by id that conformity relies on (:566-585, `np.argsort(hid)`)
//   abacus_searchsorted_i64 particle -> host halo index, `_searchsorted_parallel` (:588: np.searchsorted, side='left')
//   abacus_fenv_rank        `calc_fenv_opt` (:1961-1970): per halo-mass bin, the rank of the halo's environment mass among
//                           the halos of the bin, rescaled to [-0.5, 0.5]
// Sorting is rocPRIM's radix sort through hipCUB (stable, so equal keys keep their input order: np.argsort's order for
// the distinct ids of a catalogue; ties of Menv inside a mass bin - where the reference's quicksort order is unspecified -
// come out in input order).  Everything else is hand-written.  Host arrays in / out: staging runs once per catalogue.
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <vector>

#include "../../include/abacus_hip.h"
#include "common.hpp"

using namespace abacus;

namespace {

__global__ void iota_u32(unsigned int *v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] = (unsigned int)i;
}
// order-preserving unsigned keys
__global__ void key_i64(const int64_t *__restrict__ a, unsigned long long *__restrict__ k, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        k[i] = (unsigned long long)a[i] ^ 0x8000000000000000ull;
}
__global__ void key_f64(const double *__restrict__ a, unsigned long long *__restrict__ k, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long u = (unsigned long long)__double_as_longlong(a[i]);
        k[i] = (u >> 63) ? ~u : (u | 0x8000000000000000ull);   // NaNs (positive payload) sort last, like NumPy
    }
}
__global__ void widen_u32(const unsigned int *__restrict__ a, int64_t *__restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (int64_t)a[i];
}

// np.searchsorted(a, q, side='left'): the first index with a[idx] >= q
__global__ void searchsorted_left(const int64_t *__restrict__ a, int64_t n, const int64_t *__restrict__ q, int64_t m,
                                  int64_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = q[i];
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (a[mid] < v) lo = mid + 1;
            else hi = mid;
        }
        out[i] = lo;
    }
}

// mass bin of every halo: b with mbins[b] < M < mbins[b + 1] (both strict, :1965); 0xffff: in no bin
__global__ void mass_bin(const double *__restrict__ M, int64_t n, const double *__restrict__ mbins, int nb,
                         unsigned short *__restrict__ bin) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double m = M[i];
        int lo = 0, hi = nb + 1;   // first edge >= m
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (mbins[mid] < m) lo = mid + 1;
            else hi = mid;
        }
        // edges [0, lo) are < m; m is inside bin lo - 1 iff 1 <= lo <= nb and mbins[lo] > m (m == an edge: no bin)
        unsigned short b = 0xffffu;
        if (lo >= 1 && lo <= nb && mbins[lo] > m) b = (unsigned short)(lo - 1);
        bin[i] = b;
    }
}
__global__ void gather_bin(const unsigned short *__restrict__ bin, const unsigned int *__restrict__ idx, unsigned short *__restrict__ out,
                           int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = bin[idx[i]];
}
// after the two stable sorts the halos are grouped by bin, ascending Menv inside a bin: the first position of every bin
__global__ void bin_starts(const unsigned short *__restrict__ sorted_bin, int64_t n, int64_t *__restrict__ start, int64_t *__restrict__ count) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned short b = sorted_bin[i];
        if (b == 0xffffu) continue;
        if (i == 0 || sorted_bin[i - 1] != b) start[b] = i;
        if (i == n - 1 || sorted_bin[i + 1] != b) count[b] = i + 1;   // end (exclusive) for now
    }
}
__global__ void write_rank(const unsigned short *__restrict__ sorted_bin, const unsigned int *__restrict__ idx, int64_t n,
                           const int64_t *__restrict__ start, const int64_t *__restrict__ end, double *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned short b = sorted_bin[i];
        double r = 0.0;
        if (b != 0xffffu) {
            const int64_t N = end[b] - start[b];
            if (N > 1) r = (double)(i - start[b]) / (double)(N - 1) - 0.5;   // new_fenv_rank / (Nmask - 1) - 0.5 (:1969)
        }
        out[idx[i]] = r;
    }
}

int grid_for(int64_t n) { return (int)std::min<int64_t>(std::max<int64_t>(ceil_div(n, 256), 1), 256 * 32); }

struct Tmp {   // device allocations of one call, released on every exit path
    std::vector<void *> p;
    ~Tmp() {
        for (void *q : p)
            if (q) scratch_release(q);     // kept for the next call (runtime.hip), not freed
    }
    template <class T>
    int alloc(T **out, size_t count) {
        void *q = nullptr;
        ABACUS_TRY(scratch_acquire(&q, std::max<size_t>(count, 1) * sizeof(T)));
        p.push_back(q);
        *out = static_cast<T *>(q);
        return 0;
    }
};

// stable ascending sort of 64-bit keys carrying their input positions; `bits`: significant key bits
template <class K>
int sort_pairs(Tmp &t, const K *keys_in, K *keys_out, const unsigned int *val_in, unsigned int *val_out, int64_t n, int bits) {
    size_t bytes = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, keys_in, keys_out, val_in, val_out, (int)n, 0, bits, stream()));
    char *tmp;
    ABACUS_TRY(t.alloc(&tmp, bytes));
    prof_begin("staging_radix_sort");
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(tmp, bytes, keys_in, keys_out, val_in, val_out, (int)n, 0, bits, stream());
    prof_end("staging_radix_sort");
    HIP_TRY(e);
    return 0;
}

}  // namespace

namespace abacus {
// stable ascending sort of 16-bit keys carrying 32-bit values (the mass-bin index of the HOD filter keys, hod.hip)
int sort_pairs_u16(const unsigned short *keys_in, unsigned short *keys_out, const unsigned int *val_in, unsigned int *val_out,
                   int64_t n, DevBuf &tmp) {
    if (n <= 0) return 0;
    if (n > 0x7fffffff) return fail("sort_pairs_u16: too many elements");
    size_t bytes = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, keys_in, keys_out, val_in, val_out, (int)n, 0, 16, stream()));
    ABACUS_TRY(tmp.reserve(std::max<size_t>(bytes, 16)));
    prof_begin("hod_index_sort");
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, keys_in, keys_out, val_in, val_out, (int)n, 0, 16, stream());
    prof_end("hod_index_sort");
    HIP_TRY(e);
    return 0;
}
}  // namespace abacus

extern "C" {

int abacus_argsort_i64(const int64_t *keys, int64_t n, int64_t *order) {
    ABACUS_ENTER();
    if (n < 0 || (n > 0 && (!keys || !order))) return fail("abacus_argsort_i64: null argument");
    if (n >= ((int64_t)1 << 31)) return fail("abacus_argsort_i64: more than 2^31 - 1 keys");
    if (n == 0) return 0;
    Tmp t;
    int64_t *d_in;
    unsigned long long *k0, *k1;
    unsigned int *v0, *v1;
    ABACUS_TRY(t.alloc(&d_in, n));
    ABACUS_TRY(t.alloc(&k0, n));
    ABACUS_TRY(t.alloc(&k1, n));
    ABACUS_TRY(t.alloc(&v0, n));
    ABACUS_TRY(t.alloc(&v1, n));
    HIP_TRY(hipMemcpyAsync(d_in, keys, n * 8, hipMemcpyHostToDevice, stream()));
    ABACUS_LAUNCH("staging_keys", key_i64, dim3(grid_for(n)), dim3(256), 0, d_in, k0, n);
    ABACUS_LAUNCH("staging_iota", iota_u32, dim3(grid_for(n)), dim3(256), 0, v0, n);
    ABACUS_TRY(sort_pairs(t, k0, k1, v0, v1, n, 64));
    ABACUS_LAUNCH("staging_widen", widen_u32, dim3(grid_for(n)), dim3(256), 0, v1, d_in, n);
    HIP_TRY(hipMemcpyAsync(order, d_in, n * 8, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_searchsorted_i64(const int64_t *sorted, int64_t n, const int64_t *query, int64_t m, int64_t *out) {
    ABACUS_ENTER();
    if (n < 0 || m < 0 || (m > 0 && (!query || !out)) || (n > 0 && !sorted)) return fail("abacus_searchsorted_i64: null argument");
    if (m == 0) return 0;
    Tmp t;
    int64_t *d_a, *d_q, *d_o;
    ABACUS_TRY(t.alloc(&d_a, n));
    ABACUS_TRY(t.alloc(&d_q, m));
    ABACUS_TRY(t.alloc(&d_o, m));
    if (n) HIP_TRY(hipMemcpyAsync(d_a, sorted, n * 8, hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipMemcpyAsync(d_q, query, m * 8, hipMemcpyHostToDevice, stream()));
    ABACUS_LAUNCH("staging_searchsorted", searchsorted_left, dim3(grid_for(m)), dim3(256), 0, d_a, n, d_q, m, d_o);
    HIP_TRY(hipMemcpyAsync(out, d_o, m * 8, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

}  // extern "C"

namespace abacus {
// calc_fenv_opt on device arrays (d_M masses, d_env the ranked quantity, d_edges the n_edges increasing bin edges, all float64 in
// HBM) -> d_out: what abacus_fenv_rank runs between its copies; prepare.hip's device-resident slab path calls it directly
int fenv_rank_device(const double *d_M, const double *d_env, int64_t n, const double *d_edges, int n_edges, double *d_out) {
    if (n <= 0) return 0;
    const int nb = n_edges - 1;
    Tmp t;
    unsigned long long *k0, *k1;
    unsigned int *v0, *v1, *v2;
    unsigned short *bin, *b0, *b1;
    int64_t *start, *end;
    ABACUS_TRY(t.alloc(&k0, n));
    ABACUS_TRY(t.alloc(&k1, n));
    ABACUS_TRY(t.alloc(&v0, n));
    ABACUS_TRY(t.alloc(&v1, n));
    ABACUS_TRY(t.alloc(&v2, n));
    ABACUS_TRY(t.alloc(&bin, n));
    ABACUS_TRY(t.alloc(&b0, n));
    ABACUS_TRY(t.alloc(&b1, n));
    ABACUS_TRY(t.alloc(&start, nb));
    ABACUS_TRY(t.alloc(&end, nb));
    HIP_TRY(hipMemsetAsync(start, 0, (size_t)nb * 8, stream()));
    HIP_TRY(hipMemsetAsync(end, 0, (size_t)nb * 8, stream()));
    const int g = grid_for(n);
    ABACUS_LAUNCH("staging_mass_bin", mass_bin, dim3(g), dim3(256), 0, d_M, n, d_edges, nb, bin);
    // (1) by Menv, (2) stably by bin: grouped by bin, ascending Menv inside
    ABACUS_LAUNCH("staging_keys", key_f64, dim3(g), dim3(256), 0, d_env, k0, n);
    ABACUS_LAUNCH("staging_iota", iota_u32, dim3(g), dim3(256), 0, v0, n);
    ABACUS_TRY(sort_pairs(t, k0, k1, v0, v1, n, 64));
    ABACUS_LAUNCH("staging_gather", gather_bin, dim3(g), dim3(256), 0, bin, v1, b0, n);
    ABACUS_TRY(sort_pairs(t, b0, b1, v1, v2, n, 16));
    ABACUS_LAUNCH("staging_bin_starts", bin_starts, dim3(g), dim3(256), 0, b1, n, start, end);
    ABACUS_LAUNCH("staging_rank", write_rank, dim3(g), dim3(256), 0, b1, v2, n, start, end, d_out);
    // the scratch blocks of `t` go back to the pool here; the stream is in order, so a later acquirer's kernels run behind these
    return 0;
}
}  // namespace abacus

extern "C" {

int abacus_fenv_rank(const double *Menv, const double *halosM, int64_t n, const double *mbins, int n_edges, double *out) {
    ABACUS_ENTER();
    if (n < 0 || (n > 0 && (!Menv || !halosM || !out)) || !mbins) return fail("abacus_fenv_rank: null argument");
    if (n_edges < 2 || n_edges > 60000) return fail("abacus_fenv_rank: %d bin edges", n_edges);
    if (n >= ((int64_t)1 << 31)) return fail("abacus_fenv_rank: more than 2^31 - 1 halos");
    for (int b = 0; b + 1 < n_edges; b++)
        if (!(mbins[b + 1] > mbins[b])) return fail("abacus_fenv_rank: the mass bin edges must increase");
    if (n == 0) return 0;
    Tmp t;
    double *d_M, *d_env, *d_edges, *d_out;
    ABACUS_TRY(t.alloc(&d_M, n));
    ABACUS_TRY(t.alloc(&d_env, n));
    ABACUS_TRY(t.alloc(&d_edges, n_edges));
    ABACUS_TRY(t.alloc(&d_out, n));
    HIP_TRY(hipMemcpyAsync(d_M, halosM, n * 8, hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipMemcpyAsync(d_env, Menv, n * 8, hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipMemcpyAsync(d_edges, mbins, (size_t)n_edges * 8, hipMemcpyHostToDevice, stream()));
    ABACUS_TRY(fenv_rank_device(d_M, d_env, n, d_edges, n_edges, d_out));
    HIP_TRY(hipMemcpyAsync(out, d_out, n * 8, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

}  // extern "C"
