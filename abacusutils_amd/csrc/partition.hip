// partition_parallel (abacusnbody/analysis/tsc.py:259-384, sort=False) on the device: stable counting sort of the
// particles into `npartition` stripes along one coordinate.  Our own deposit does not need stripes (tsc.hip cuts the
// mesh into LDS tiles instead); this entry point exists because partition_parallel is public API of the reference
// (`__all__`, tsc.py:7) with a tested contract (tests/test_tsc.py:162-208): same keys, same `starts`, stable order.
//
// key = min(int32(pos[coord] * dtype(npartition/box)), npartition-1)   (tsc.py:329,335)
// Stable order comes from an LSD radix sort of (key, original index) pairs (hipCUB, stable by construction); a
// gather then writes the particles.  HBM-bound: ~3 passes over 8-B pairs + one 12-B gather.
#include <hipcub/hipcub.hpp>

#include <vector>

#include "../../include/abacus_hip.h"
#include "common.hpp"

using namespace abacus;

namespace {

template <typename PT>
__global__ void part_keys(const PT *__restrict__ pos, int64_t n, int coord, PT inv_pwidth, int npartition,
                          int *__restrict__ keys, unsigned int *__restrict__ idx, unsigned long long *__restrict__ hist) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int k = (int)(pos[3 * i + coord] * inv_pwidth);
        k = max(min(k, npartition - 1), 0);
        keys[i] = k;
        idx[i] = (unsigned int)i;
        atomicAdd(&hist[k], 1ull);
    }
}

template <typename PT>
__global__ void part_gather(const PT *__restrict__ pos, const PT *__restrict__ w, const unsigned int *__restrict__ idx,
                            int64_t n, PT *__restrict__ psort, PT *__restrict__ wsort) {
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = idx[s];
        psort[3 * s] = pos[3 * i];
        psort[3 * s + 1] = pos[3 * i + 1];
        psort[3 * s + 2] = pos[3 * i + 2];
        if (w) wsort[s] = w[i];
    }
}

template <typename PT>
int partition_impl(const void *pos_, int64_t n, const void *w_, int npartition, double box, int coord, void *psort_,
                   int64_t *starts, void *wsort_) {
    ABACUS_ENTER();
    if (npartition < 1) return fail("abacus_partition: npartition < 1");
    if (coord < 0 || coord > 2) return fail("abacus_partition: coord out of range");
    if (n >= (int64_t)1 << 32) return fail("abacus_partition: more than 2^32 particles");
    DevBuf dpos, dw, dps, dws, keys, keys2, idx, idx2, hist, tmp;
    int rc = 0;
    std::vector<unsigned long long> h((size_t)npartition);
    do {
#define TRYB(x) if ((rc = (x)) != 0) break
        const size_t n1 = (size_t)std::max<int64_t>(n, 1);
        TRYB(dpos.reserve(3 * n1 * sizeof(PT)));
        TRYB(dps.reserve(3 * n1 * sizeof(PT)));
        if (w_) {
            TRYB(dw.reserve(n1 * sizeof(PT)));
            TRYB(dws.reserve(n1 * sizeof(PT)));
        }
        TRYB(keys.reserve(n1 * 4));
        TRYB(keys2.reserve(n1 * 4));
        TRYB(idx.reserve(n1 * 4));
        TRYB(idx2.reserve(n1 * 4));
        TRYB(hist.reserve((size_t)npartition * 8));
        (void)hipMemcpyAsync(dpos.p, pos_, 3 * n * sizeof(PT), hipMemcpyHostToDevice, stream());
        if (w_) (void)hipMemcpyAsync(dw.p, w_, n * sizeof(PT), hipMemcpyHostToDevice, stream());
        (void)hipMemsetAsync(hist.p, 0, (size_t)npartition * 8, stream());
        const PT inv_pwidth = (PT)(npartition / box);
        const int nblk = (int)std::min<int64_t>(std::max<int64_t>(ceil_div(n, 256), 1), 4096);
        if (n > 0) {
            hipLaunchKernelGGL(part_keys<PT>, dim3(nblk), dim3(256), 0, stream(), dpos.as<PT>(), n, coord, inv_pwidth,
                               npartition, keys.as<int>(), idx.as<unsigned int>(), hist.as<unsigned long long>());
            int end_bit = 1;
            while ((1ll << end_bit) < npartition) end_bit++;
            size_t tmp_bytes = 0;
            (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys.as<int>(), keys2.as<int>(),
                                               idx.as<unsigned int>(), idx2.as<unsigned int>(), (int)n, 0, end_bit,
                                               stream());
            TRYB(tmp.reserve(tmp_bytes));
            if (hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, keys.as<int>(), keys2.as<int>(),
                                                   idx.as<unsigned int>(), idx2.as<unsigned int>(), (int)n, 0, end_bit,
                                                   stream()) != hipSuccess) {
                rc = fail("abacus_partition: radix sort failed");
                break;
            }
            hipLaunchKernelGGL(part_gather<PT>, dim3(nblk), dim3(256), 0, stream(), dpos.as<PT>(),
                               w_ ? dw.as<PT>() : (const PT *)nullptr, idx2.as<unsigned int>(), n, dps.as<PT>(),
                               w_ ? dws.as<PT>() : (PT *)nullptr);
            (void)hipMemcpyAsync(psort_, dps.p, 3 * n * sizeof(PT), hipMemcpyDeviceToHost, stream());
            if (w_) (void)hipMemcpyAsync(wsort_, dws.p, n * sizeof(PT), hipMemcpyDeviceToHost, stream());
        }
        (void)hipMemcpyAsync(h.data(), hist.p, (size_t)npartition * 8, hipMemcpyDeviceToHost, stream());
        if (hipStreamSynchronize(stream()) != hipSuccess || hipGetLastError() != hipSuccess) {
            rc = fail("abacus_partition: device error");
            break;
        }
        int64_t run = 0;
        for (int k = 0; k < npartition; k++) {
            starts[k] = run;
            run += (int64_t)h[k];
        }
        starts[npartition] = n;
#undef TRYB
    } while (0);
    for (DevBuf *b : {&dpos, &dw, &dps, &dws, &keys, &keys2, &idx, &idx2, &hist, &tmp}) (void)b->release();
    return rc;
}

// owner of a particle in the x-slab decomposition of the mesh: wrapped x in [r L/W, (r+1) L/W), float32 like the deposit.
// fold: the box is cut into 2 W slabs and rank r owns slabs r and r + W (planes x and x + n/2 on one rank: the folded slabs
// of the P(k) estimator) - the caller passes inv_width = 2 W / L and slabs = 2 W
__global__ void route_keys(const float *__restrict__ pos, int64_t n, float box, float inv_width, int slabs, int world,
                           int *__restrict__ keys, unsigned int *__restrict__ idx, unsigned long long *__restrict__ hist) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = pos[3 * i];
        const float xw = x - floorf(x / box) * box;
        int k = (int)(xw * inv_width);
        k = max(min(k, slabs - 1), 0);
        if (k >= world) k -= world;
        keys[i] = k;
        idx[i] = (unsigned int)i;
        atomicAdd(&hist[k], 1ull);
    }
}

}  // namespace

// Particle routing of the slab P(k) on the device (no reference counterpart: the reference's mesh is single-process):
// stable bucket sort of (pos, w) by owning rank; counts[world] on the host.  The variable-size blocks then travel with
// abacus_comm_all_to_all_v.
extern "C" int abacus_slab_route_dev(const float *pos, int64_t n, const float *w, double Lbox, int world, int fold,
                                     float *pos_out, float *w_out, int64_t *counts) {
    ABACUS_ENTER();
    if (world < 1 || !counts || (n > 0 && (!pos || !pos_out)) || (w && !w_out)) return fail("abacus_slab_route_dev: bad argument");
    if (n >= (int64_t)1 << 31) return fail("abacus_slab_route_dev: more than 2^31 particles per rank");
    static DevBuf keys, keys2, idx, idx2, hist, tmp;   // library scratch, reused across calls (API mutex held)
    const size_t n1 = (size_t)std::max<int64_t>(n, 1);
    ABACUS_TRY(keys.reserve(n1 * 4));
    ABACUS_TRY(keys2.reserve(n1 * 4));
    ABACUS_TRY(idx.reserve(n1 * 4));
    ABACUS_TRY(idx2.reserve(n1 * 4));
    ABACUS_TRY(hist.reserve((size_t)world * 8));
    HIP_TRY(hipMemsetAsync(hist.p, 0, (size_t)world * 8, stream()));
    std::vector<unsigned long long> h((size_t)world, 0ull);
    if (n > 0) {
        const int nblk = (int)std::min<int64_t>(ceil_div(n, 256), 4096);
        const float box = (float)Lbox;
        ABACUS_LAUNCH("route_keys", route_keys, dim3(nblk), dim3(256), 0, pos, n, box, (float)(fold ? 2 * world : world) / box,
                      fold ? 2 * world : world, world, keys.as<int>(),
                      idx.as<unsigned int>(), hist.as<unsigned long long>());
        int end_bit = 1;
        while ((1ll << end_bit) < world) end_bit++;
        size_t tmp_bytes = 0;
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys.as<int>(), keys2.as<int>(), idx.as<unsigned int>(),
                                                 idx2.as<unsigned int>(), (int)n, 0, end_bit, stream());
        ABACUS_TRY(tmp.reserve(tmp_bytes));
        if (hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, keys.as<int>(), keys2.as<int>(), idx.as<unsigned int>(),
                                               idx2.as<unsigned int>(), (int)n, 0, end_bit, stream()) != hipSuccess)
            return fail("abacus_slab_route_dev: radix sort failed");
        ABACUS_LAUNCH("route_gather", part_gather<float>, dim3(nblk), dim3(256), 0, pos, w, idx2.as<unsigned int>(), n, pos_out,
                      w_out);
    }
    HIP_TRY(hipMemcpyAsync(h.data(), hist.p, (size_t)world * 8, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    for (int k = 0; k < world; k++) counts[k] = (int64_t)h[k];
    return 0;
}

extern "C" int abacus_partition(const void *pos, int64_t n, const void *weights, int dtype, int npartition,
                                double box, int coord, void *psort, int64_t *starts, void *wsort) {
    if ((n > 0 && (!pos || !psort)) || !starts) return fail("abacus_partition: null argument");
    if (dtype == ABACUS_F32) return partition_impl<float>(pos, n, weights, npartition, box, coord, psort, starts, wsort);
    if (dtype == ABACUS_F64) return partition_impl<double>(pos, n, weights, npartition, box, coord, psort, starts, wsort);
    return fail("abacus_partition: unknown dtype code");
}
