// Periodic-box pair counting on MI355X (gfx950): the kernel behind compute_xirppi / compute_wp / compute_multipole.
// In the reference this arithmetic lives in the third-party Corrfunc library (Corrfunc.theory.DDrppi / DDsmu / DD,
// called at abacusnbody/analysis/tpcf_corrfunc.py:144-156,164-179,240-273,328-362 and
// scripts/emulator/generate_cfs/generate_cf.py:65-74); it is not vendored and no reference test covers it, so the
// conventions below restate Corrfunc's published behaviour and parity is pinned only against the brute-force
// counter in oracle/abacus_oracle.c (same float32 expressions, hence identical integer counts).
//
//   float32 coordinates; periodic minimum image per pair; squared separations compared with squared float32 edges;
//   r-bin b holds edges[b] <= r < edges[b+1]; DDrppi: pi-bin = int(|dz| / (pimax/npibins)), |dz| < pimax;
//   DDsmu: mu = |dz|/s, mu-bin = int(mu * nmubins/mu_max), mu < mu_max; autocorrelation counts ordered pairs (i != j).
//
// Algorithm: both point sets are counting-sorted into a cell grid with cell size >= the largest separation
// (SoA x|y|z per set, cell_start offsets).  One workgroup per non-empty cell of set 1 walks the 27 neighbour cells
// of set 2, staging 256 neighbours at a time in LDS; every thread owns one point of set 1 and scans the staged
// chunk (LDS broadcast reads), binning into a per-workgroup LDS histogram that is flushed with 64-bit atomics.
// Work is O(N * nbar * 27 * cell^3); the loop is VALU/LDS bound (not HBM): ~20 VALU ops per candidate pair.
#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "../../include/abacus_hip.h"
#include "common.hpp"

using namespace abacus;

namespace abacus {
int exclusive_scan_u32(unsigned int *counters, int64_t n, int64_t *out, DevBuf &scratch, int zero_counters);
}

namespace {

constexpr int PB = 256;          // threads per workgroup
constexpr int MAX_HIST = 8192;   // LDS histogram entries (bins * sub-bins)

struct CellGrid {
    int ncx, ncy, ncz;
    float box, inv_box;
};

__device__ __forceinline__ int cell_coord(float v, int nc, float inv_box) {
    // fractional position in the box, periodic: works for [0,L), [-L/2,L/2) or anything else
    float f = v * inv_box;
    f -= floorf(f);
    int c = (int)(f * (float)nc);
    return c >= nc ? nc - 1 : c;
}

// Coordinate frame of a catalogue: the smallest and the largest coordinate per dimension as order-preserving integer keys
// (so that atomicMin / atomicMax work on floats).  Corrfunc accepts any coordinate range; galaxies from the HOD live in
// [-L/2, L/2).  When every coordinate of both sets lies in ONE interval [a, a + L) the periodic image of a pair of cells
// is known per cell pair (pair_count3); `outside` only says that the frame is not [0, L).
struct Frame {
    unsigned int mn[3], mx[3];
};
__device__ __forceinline__ unsigned int fkey(float v) {
    const unsigned int u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float funkey(unsigned int k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// COUNT: per-cell counters by global atomics (counting sort, small inputs); else the cell ids become the keys of a radix sort
// and idx the values
template <bool COUNT>
__global__ void cell_count(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z,
                           int64_t n, CellGrid g, unsigned int *__restrict__ counts, unsigned int *__restrict__ cellid,
                           unsigned int *__restrict__ idx, float4 *__restrict__ packed, int *__restrict__ outside,
                           Frame *__restrict__ frame) {
    bool out = false;
    unsigned int mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v[3] = {x[i], y[i], z[i]};
        const int c = (cell_coord(v[0], g.ncx, g.inv_box) * g.ncy + cell_coord(v[1], g.ncy, g.inv_box)) * g.ncz +
                      cell_coord(v[2], g.ncz, g.inv_box);
        cellid[i] = (unsigned int)c;
        if (COUNT) atomicAdd(&counts[c], 1u);
        else idx[i] = (unsigned int)i, packed[i] = make_float4(v[0], v[1], v[2], 0.f);   // one 16-B piece per point for the gather
        out = out || !(v[0] >= 0.f && v[0] < g.box && v[1] >= 0.f && v[1] < g.box && v[2] >= 0.f && v[2] < g.box);
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const unsigned int k = fkey(v[d]);
            mn[d] = min(mn[d], k), mx[d] = max(mx[d], k);
        }
    }
    if (out) *outside = 1;   // some coordinate is not in [0, L)
    // frame: wave reduction, then ONE pair of global atomics per workgroup and dimension (one per wave was 10^5 atomics
    // on six addresses: 1 ms of this kernel's 1.1)
    __shared__ unsigned int s_mn[3], s_mx[3];
    if (threadIdx.x < 3) s_mn[threadIdx.x] = 0xffffffffu, s_mx[threadIdx.x] = 0u;
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 3; d++) {
        unsigned int a = mn[d], b = mx[d];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a = min(a, (unsigned int)__shfl_xor((int)a, off, 64)), b = max(b, (unsigned int)__shfl_xor((int)b, off, 64));
        if ((threadIdx.x & 63) == 0 && a <= b) atomicMin(&s_mn[d], a), atomicMax(&s_mx[d], b);
    }
    __syncthreads();
    if (threadIdx.x < 3 && s_mn[threadIdx.x] <= s_mx[threadIdx.x])
        atomicMin(&frame->mn[threadIdx.x], s_mn[threadIdx.x]), atomicMax(&frame->mx[threadIdx.x], s_mx[threadIdx.x]);
}

__global__ void cell_fill(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z,
                          int64_t n, const unsigned int *__restrict__ cellid, const int64_t *__restrict__ start,
                          unsigned int *__restrict__ cursor, float *__restrict__ sx, float *__restrict__ sy,
                          float *__restrict__ sz) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned int c = cellid[i];
        const int64_t s = start[c] + atomicAdd(&cursor[c], 1u);
        sx[s] = x[i];
        sy[s] = y[i];
        sz[s] = z[i];
    }
}

// behind the radix sort of (cell id, point index): the points in cell order (ties in input order: the sort is stable, so
// the sorted arrays do not depend on the run) ...
__global__ void cell_gather(const float4 *__restrict__ packed, int64_t n, const unsigned int *__restrict__ idx,
                            float *__restrict__ sx, float *__restrict__ sy, float *__restrict__ sz) {
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x) {
        const float4 p = packed[idx[s]];   // one memory sector per point instead of three (x, y, z live in separate arrays)
        sx[s] = p.x, sy[s] = p.y, sz[s] = p.z;
    }
}
// ... and the first point of every cell: start[c] = number of keys below c (c = 0 .. ncell)
__global__ void cell_starts(const unsigned int *__restrict__ keys, int64_t n, int64_t ncell, int64_t *__restrict__ start) {
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c <= ncell; c += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)keys[mid] < c) lo = mid + 1;
            else hi = mid;
        }
        start[c] = lo;
    }
}

struct PairArgs {
    int mode, autocorr, in_box;   // in_box: every coordinate of both sets lies in [0, L)
    CellGrid g;
    int nbins, nsub;
    float half, pimax, dpi, mu_max, inv_dmu;
    const float *edges2;   // (nbins+1) squared edges
    // bin look-up of pair_count3: the float32 bit pattern of r^2, shifted by lut_sh, minus lut_off indexes lut_ncell cells, none
    // of which holds more than one inner edge (0 cells: the kernel walks the edges instead)
    int lut_sh, lut_off, lut_ncell;
    int no_blocks;         // pair_count3: every cell a job of its own (comparator of the two-cell blocks)
    const float *x1, *y1, *z1, *x2, *y2, *z2;
    const int64_t *start1, *start2;
    unsigned long long *npairs;
};

__device__ __forceinline__ float min_image(float d, float half, float box) {
    if (d > half) return d - box;
    if (d < -half) return d + box;
    return d;
}

// One workgroup per (cell of set 1, slice of its points).
__global__ __launch_bounds__(PB) void pair_count(PairArgs a, const int *__restrict__ work_cell,
                                                 const int *__restrict__ work_off) {
    __shared__ float jx[PB], jy[PB], jz[PB];
    __shared__ unsigned int hist[MAX_HIST];
    __shared__ float e2[64];
    const int tid = threadIdx.x;
    const int nh = a.nbins * a.nsub;
    for (int q = tid; q < nh; q += PB) hist[q] = 0u;
    if (tid <= a.nbins) e2[tid] = a.edges2[tid];
    __syncthreads();
    const float lo2 = e2[0], hi2 = e2[a.nbins];
    const int c1 = work_cell[blockIdx.x];
    const int64_t i0 = a.start1[c1] + work_off[blockIdx.x];
    const int64_t iend = a.start1[c1 + 1];
    const int64_t i = i0 + tid;
    const bool active = i < iend;
    float xi = 0, yi = 0, zi = 0;
    if (active) xi = a.x1[i], yi = a.y1[i], zi = a.z1[i];
    const int cz = c1 % a.g.ncz, cy = (c1 / a.g.ncz) % a.g.ncy, cx = c1 / (a.g.ncz * a.g.ncy);
    // neighbour ranges: all cells when a dimension has fewer than 3 cells (then the 27-stencil would repeat cells)
    const int rx = a.g.ncx >= 3 ? 1 : 0, ry = a.g.ncy >= 3 ? 1 : 0, rz = a.g.ncz >= 3 ? 1 : 0;
    for (int ox = -rx; ox <= rx; ox++)
        for (int oy = -ry; oy <= ry; oy++)
            for (int oz = -rz; oz <= rz; oz++) {
                int nx = cx + ox, ny = cy + oy, nz = cz + oz;
                nx = nx < 0 ? nx + a.g.ncx : (nx >= a.g.ncx ? nx - a.g.ncx : nx);
                ny = ny < 0 ? ny + a.g.ncy : (ny >= a.g.ncy ? ny - a.g.ncy : ny);
                nz = nz < 0 ? nz + a.g.ncz : (nz >= a.g.ncz ? nz - a.g.ncz : nz);
                const int c2 = (nx * a.g.ncy + ny) * a.g.ncz + nz;
                const int64_t j0 = a.start2[c2], j1 = a.start2[c2 + 1];
                for (int64_t jb = j0; jb < j1; jb += PB) {
                    const int m = (int)min((int64_t)PB, j1 - jb);
                    __syncthreads();
                    if (tid < m) {
                        jx[tid] = a.x2[jb + tid];
                        jy[tid] = a.y2[jb + tid];
                        jz[tid] = a.z2[jb + tid];
                    }
                    __syncthreads();
                    if (!active) continue;
                    for (int q = 0; q < m; q++) {
                        if (a.autocorr && jb + q == i) continue;   // same point (the two sorted sets are one array)
                        const float dx = min_image(xi - jx[q], a.half, a.g.box);
                        const float dy = min_image(yi - jy[q], a.half, a.g.box);
                        const float dz = min_image(zi - jz[q], a.half, a.g.box);
                        float r2;
                        int sub = 0;
                        if (a.mode == 1) {
                            const float adz = fabsf(dz);
                            if (adz >= a.pimax) continue;
                            r2 = dx * dx + dy * dy;
                            sub = (int)(adz / a.dpi);
                            if (sub >= a.nsub) continue;
                        } else {
                            r2 = dx * dx + dy * dy + dz * dz;
                        }
                        if (r2 < lo2 || r2 >= hi2) continue;
                        int b = 0;
                        while (r2 >= e2[b + 1]) b++;
                        if (a.mode == 2) {
                            const float s = sqrtf(r2);
                            const float mu = s > 0.f ? fabsf(dz) / s : 0.f;
                            if (mu >= a.mu_max) continue;
                            sub = (int)(mu * a.inv_dmu);
                            if (sub >= a.nsub) continue;
                        }
                        atomicAdd(&hist[b * a.nsub + sub], 1u);
                    }
                }
            }
    __syncthreads();
    for (int q = tid; q < nh; q += PB)
        if (hist[q]) atomicAdd(&a.npairs[q], (unsigned long long)hist[q]);
}

// Second-generation kernel: the (i, j) pairs of (a slice of <= 256 points of cell c1) x (a chunk of <= 256 points of a
// neighbour cell c2) are FLATTENED over the workgroup: thread t takes pairs p = t, t + 256, ... with i = p / m,
// j = p % m.  Every lane has work whatever the cell populations are (35 points per cell at 10^7 points in the
// 2 Gpc/h box left 86 % of the one-thread-per-i-point kernel's lanes idle), and both point sets are read from LDS
// (consecutive lanes: consecutive j, the same i for runs of m lanes: broadcast).  When a dimension has at least 5 cells
// the periodic image of a neighbour cell is known per cell: d = (xi - xj) + shift with shift in {0, +-L} is the very
// expression the per-pair minimum image evaluates, without its compares.  Accepted pairs (15 % of the candidates)
// search their bin from the top - most of the volume is in the outer bins - and bump the LDS histogram.
// AGG = false (dense catalogues): one staging round per neighbour cell chunk, the periodic shift and the self-pair
// test are uniform per round.  AGG = true (a few points per cell): the points of all 27 neighbour cells are appended
// to one 256-slot staging buffer with their shifts and indices, one round per slice instead of 27.
template <int MODE, bool AGG>
__global__ __launch_bounds__(PB) void pair_count2(PairArgs a, const int *__restrict__ outside, int ncell) {
    __shared__ float ix[PB], iy[PB], iz[PB];
    __shared__ float jx[PB], jy[PB], jz[PB];
    __shared__ float jsx[AGG ? PB : 1], jsy[AGG ? PB : 1], jsz[AGG ? PB : 1];   // per-point periodic shifts
    __shared__ int jg[AGG ? PB : 1];                                               // index in the sorted array
    extern __shared__ unsigned int hist[];   // nbins * nsub counters: small histograms leave room for 8 workgroups per CU
    __shared__ float e2[64];
    __shared__ int64_t nb_j0[27], nb_j1[27];   // ranges and shifts of the neighbour cells, fetched in ONE round trip
    __shared__ float nb_sh[27][3];
    const int tid = threadIdx.x;
    const int nh = a.nbins * a.nsub;
    // persistent workgroups over the cells of set 1 (no host-built work list, no host synchronisation before the
    // launch): the LDS histogram lives across cells and is flushed once - per-cell flushes of a sparse catalogue are
    // millions of global atomics on a dozen addresses.  A cell with more than 256 points is walked in slices.
    for (int q = tid; q < nh; q += PB) hist[q] = 0u;
    if (tid <= a.nbins) e2[tid] = a.edges2[tid];
    const bool in_box = *outside == 0;
    for (int c1 = blockIdx.x; c1 < ncell; c1 += gridDim.x) {
    const int64_t cbeg = a.start1[c1], cend = a.start1[c1 + 1];
    if (cbeg == cend) continue;
    __syncthreads();   // the previous cell's reads of the neighbour tables are done
    const int cz = c1 % a.g.ncz, cy = (c1 / a.g.ncz) % a.g.ncy, cx = c1 / (a.g.ncz * a.g.ncy);
    const int rx = a.g.ncx >= 3 ? 1 : 0, ry = a.g.ncy >= 3 ? 1 : 0, rz = a.g.ncz >= 3 ? 1 : 0;
    const int wx = 2 * rx + 1, wy = 2 * ry + 1, wz = 2 * rz + 1, nnb = wx * wy * wz;
    if (tid < nnb) {
        int nx = cx + tid / (wy * wz) - rx, ny = cy + (tid / wz) % wy - ry, nz = cz + tid % wz - rz;
        float shx = 0.f, shy = 0.f, shz = 0.f;   // xi - xj is about -L when c2 wrapped below 0: add L
        if (nx < 0) nx += a.g.ncx, shx = a.g.box;
        else if (nx >= a.g.ncx) nx -= a.g.ncx, shx = -a.g.box;
        if (ny < 0) ny += a.g.ncy, shy = a.g.box;
        else if (ny >= a.g.ncy) ny -= a.g.ncy, shy = -a.g.box;
        if (nz < 0) nz += a.g.ncz, shz = a.g.box;
        else if (nz >= a.g.ncz) nz -= a.g.ncz, shz = -a.g.box;
        const int c2 = (nx * a.g.ncy + ny) * a.g.ncz + nz;
        nb_j0[tid] = a.start2[c2], nb_j1[tid] = a.start2[c2 + 1];
        nb_sh[tid][0] = shx, nb_sh[tid][1] = shy, nb_sh[tid][2] = shz;
    }
    __syncthreads();
    const float lo2 = e2[0], hi2 = e2[a.nbins];
    // per-cell periodic shifts need |d| of an unwrapped neighbour pair (< 2 cells) to stay below half the box
    const bool fast = in_box && a.g.ncx >= 5 && a.g.ncy >= 5 && a.g.ncz >= 5;

    // all pairs (slice of c1) x (the `fill` staged neighbour points); !AGG: uniform shift (ux, uy, uz), the staged chunk
    // starts at sorted index jbase (self pairs only when `self_chunk`)
    auto process = [&](int64_t i0, int ni, int fill, float ux, float uy, float uz, int64_t jbase, bool self_chunk) {
        const int total = ni * fill;
        const float inv_m = 1.0f / (float)fill;
        for (int p = tid; p < total; p += PB) {
            const int i = (int)(((float)p + 0.5f) * inv_m);   // exact: p < 2^16, fill <= 2^8
            const int j = p - i * fill;
            if (AGG) {
                if (a.autocorr && (int64_t)jg[j] == i0 + i) continue;   // the same point
            } else {
                if (self_chunk && jbase + j == i0 + i) continue;
            }
            float dx = ix[i] - jx[j], dy = iy[i] - jy[j], dz = iz[i] - jz[j];
            if (fast) {
                if (AGG) dx += jsx[j], dy += jsy[j], dz += jsz[j];
                else dx += ux, dy += uy, dz += uz;
            } else {
                dx = min_image(dx, a.half, a.g.box);
                dy = min_image(dy, a.half, a.g.box);
                dz = min_image(dz, a.half, a.g.box);
            }
            float r2;
            int sub = 0;
            if (MODE == 1) {
                const float adz = fabsf(dz);
                if (adz >= a.pimax) continue;
                r2 = dx * dx + dy * dy;
                if (r2 < lo2 || r2 >= hi2) continue;
                sub = (int)(adz / a.dpi);
                if (sub >= a.nsub) continue;
            } else {
                r2 = dx * dx + dy * dy + dz * dz;
                if (r2 < lo2 || r2 >= hi2) continue;
            }
            int b = a.nbins - 1;
            while (r2 < e2[b]) b--;
            if (MODE == 2) {
                const float sr = sqrtf(r2);
                const float mu = sr > 0.f ? fabsf(dz) / sr : 0.f;
                if (mu >= a.mu_max) continue;
                sub = (int)(mu * a.inv_dmu);
                if (sub >= a.nsub) continue;
            }
            atomicAdd(&hist[b * a.nsub + sub], 1u);
        }
    };

    for (int64_t i0 = cbeg; i0 < cend; i0 += PB) {
        const int ni = (int)min((int64_t)PB, cend - i0);
        __syncthreads();   // the previous slice's reads of ix/iy/iz (and of the staging buffer) are done
        if (tid < ni) ix[tid] = a.x1[i0 + tid], iy[tid] = a.y1[i0 + tid], iz[tid] = a.z1[i0 + tid];
        int fill = 0;
        for (int nb = 0; nb < nnb; nb++) {
            int64_t jb = nb_j0[nb];
            const int64_t j1 = nb_j1[nb];
            const float shx = nb_sh[nb][0], shy = nb_sh[nb][1], shz = nb_sh[nb][2];
            while (jb < j1) {
                const int take = (int)min((int64_t)(PB - fill), j1 - jb);
                if (!AGG) __syncthreads();   // the previous round's reads of the staging buffer are done
                if (tid < take) {
                    const int q = fill + tid;
                    jx[q] = a.x2[jb + tid], jy[q] = a.y2[jb + tid], jz[q] = a.z2[jb + tid];
                    if (AGG) {
                        jsx[q] = shx, jsy[q] = shy, jsz[q] = shz;
                        jg[q] = (int)(jb + tid);
                    }
                }
                if (AGG) {
                    fill += take;
                    if (fill == PB) {
                        __syncthreads();
                        process(i0, ni, fill, 0.f, 0.f, 0.f, 0, false);
                        __syncthreads();
                        fill = 0;
                    }
                } else {
                    __syncthreads();
                    process(i0, ni, take, shx, shy, shz, jb, a.autocorr && jb < i0 + ni && jb + take > i0);
                }
                jb += take;
            }
        }
        if (AGG && fill) {
            __syncthreads();
            process(i0, ni, fill, 0.f, 0.f, 0.f, 0, false);
        }
    }
    }   // cells
    __syncthreads();
    for (int q = tid; q < nh; q += PB)
        if (hist[q]) atomicAdd(&a.npairs[q], (unsigned long long)hist[q]);
}

// Third-generation kernel: ONE WAVE per cell of set 1, cells of size reach / R (R = 1 or 2), and for an autocorrelation the
// HALF stencil - every unordered pair is evaluated once and the histogram is doubled at the end (exact: dx -> -dx,
// the minimum image and the per-cell shifts negate exactly, so r^2, |dz| and mu of (i, j) and (j, i) are bit-equal).
// Against the workgroup-per-cell kernel with cells of the full reach: R = 2 evaluates 5^3 / 8 instead of 27 cell volumes
// per point (x 1/1.7), the half stencil halves that again, and a wave needs no workgroup barrier - the staging buffers, the
// segment table and the i-slice are wave-private LDS, the waves of a workgroup only share the histogram.
//   * Neighbour cells along z are consecutive in the sorted arrays: the stencil is walked as (2R+1)^2 ROWS, each one
//     contiguous range of points (two when the row wraps around the box in z).
//   * Half stencil: rows (ox, oy) lexicographically after (0, 0) with all 2R+1 cells; of row (0, 0) the cells behind the
//     own one, and the own cell with the test j > i (sorted indices).
//   * Periodic images per segment: coordinates of both sets lie in one interval [a, a + L) (Frame), so a point's
//     coordinate is L frac + const, raised by L when its fraction lies below frac(a); the true separation of a pair from
//     cells (c1, c1 + o) is (xi - xj) - L (w + s(c1) - s(c2)) with w the index wrap and s(c) = [c below the cut cell] - the
//     same single float addition of 0 or -+L the per-pair minimum image performs.  Segments touching the cut cell or its
//     two neighbours (mixed cells, float rounding at the cut) take the per-pair minimum image instead.
constexpr int P3_WAVES = 4, P3_JCAP = 320, P3_ICAP = 64, P3_SLOTS = 64;

//   * Bin of a pair (LUT): r^2 is a positive float, so its bit pattern is monotone in it; the pattern's top bits index a
//     table of cells (2^m per octave, m chosen by the host so that no cell holds two edges) whose entry is the bin of the
//     cell's lower end and the next edge: bin = entry.bin + (r^2 >= entry.edge) - the walk's answer (largest b with
//     edges2[b] <= r^2) from ONE LDS read instead of a data-dependent loop of them.  The inner loop is straight-line,
//     predicated code (counting the pairs of the last bin - three quarters of those in range for logarithmic bins - by a
//     wave-wide ballot + population count instead of LDS atomics measured slower: 6.6 against 5.8 ms).
template <int MODE, bool LUT>
__global__ __launch_bounds__(P3_WAVES * 64) void pair_count3(PairArgs a, const Frame *__restrict__ frame, int ncell, int R,
                                                              unsigned long long *__restrict__ evaluated) {
    __shared__ int64_t seg_j0[P3_WAVES][P3_SLOTS];
    __shared__ int seg_pre[P3_WAVES][P3_SLOTS + 1], seg_code[P3_WAVES][P3_SLOTS];   // exclusive prefix of the segment lengths
    __shared__ float e2[64];
    __shared__ int s_cut[3], s_general[3], s_ok;
    extern __shared__ __align__(8) unsigned int hist[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // w: a scalar, like everything per cell
    const int nh = a.nbins * a.nsub;
    uint2 *lut = reinterpret_cast<uint2 *>(hist + ((nh + 1) & ~1));
    for (int q = tid; q < nh; q += P3_WAVES * 64) hist[q] = 0u;
    if (tid <= a.nbins) e2[tid] = a.edges2[tid];
    if (tid == 0) s_ok = 1;
    __syncthreads();
    if (LUT)
        for (int c = tid; c < a.lut_ncell; c += P3_WAVES * 64) {
            const float low = __uint_as_float((unsigned int)(c + a.lut_off) << a.lut_sh);
            int cnt = 0;
            for (int b = 0; b < a.nbins; b++) cnt += e2[b] <= low ? 1 : 0;
            const int b0 = max(cnt - 1, 0);
            lut[c] = make_uint2((unsigned int)b0, __float_as_uint(b0 + 1 < a.nbins ? e2[b0 + 1] : __builtin_huge_valf()));
        }
    if (tid < 3) {
        const float lo = funkey(frame->mn[tid]), hi = funkey(frame->mx[tid]);
        const int nc = tid == 0 ? a.g.ncx : (tid == 1 ? a.g.ncy : a.g.ncz);
        // one period holds everything?  (margin: the cut cells absorb rounding at the ends)
        if (!((double)hi - (double)lo < (double)a.g.box)) s_ok = 0;
        const bool std_frame = lo >= 0.f && hi < a.g.box;       // [0, L): no raised coordinates, the cut is the index wrap
        s_general[tid] = std_frame ? 0 : 1;
        s_cut[tid] = std_frame ? 0 : cell_coord(lo, nc, a.g.inv_box);
    }
    __syncthreads();
    const float lo2 = e2[0], hi2 = e2[a.nbins];
    const bool frame_ok = s_ok != 0;
    const int ncx = a.g.ncx, ncy = a.g.ncy, ncz = a.g.ncz, W = 2 * R + 1;
    const int nrow = a.autocorr ? (W * W - 1) / 2 + 1 : W * W;     // half stencil: rows after (0, 0), then row (0, 0)
    const int nslot = 2 * nrow + (a.autocorr ? 1 : 0);
    // s(c) - [c below the cut] - and the mixed set {cut-1, cut, cut+1} of a dimension
    auto below = [&](int d, int c) { return s_general[d] && c < s_cut[d] ? 1 : 0; };
    auto mixed1 = [&](int d, int c, int nc) {
        if (!s_general[d]) return false;
        int t = c - s_cut[d];
        if (t < 0) t += nc;
        return t == 0 || t == 1 || t == nc - 1;
    };
    unsigned long long n_eval = 0;   // candidate pairs this wave evaluated (uniform per wave; lane 0 reports)
    const int lsh = a.lut_sh, loff = a.lut_off;
    // the global loads of a cell's table (its own range, the ranges of its segments): issued ONE CELL AHEAD, so that
    // their round trip runs under the pair loop of the cell before
    struct Table {
        int64_t cbeg, cend, j0;
        int len, code;
    };
    // Lane constants of the table: the stencil row / part of slot `lane`.  For a cell at least R cells away from every face of
    // the grid in a standard frame no row wraps and nothing is "mixed": the slot's cell range is the own cell index plus a
    // constant and its code is a constant - two adds and two loads instead of the ~80 instructions of the general case
    // (which the cells of the outer shell and every cell of a shifted frame still take).  PMC had the kernel at 93 % VALU
    // utilisation with two thirds of its vector instructions outside the pair loop.
    int s_ox = 0, s_oy = 0, s_zlo = -R, s_zhi = R, s_filt = 0;
    if (a.autocorr && lane == 2 * nrow) {
        s_zlo = s_zhi = 0, s_filt = 1;
    } else {
        const int row = lane >> 1;
        if (a.autocorr) {
            if (row == nrow - 1) s_zlo = 1;
            else {
                const int t2 = row + (R * W + R) + 1;
                s_ox = t2 / W - R, s_oy = t2 % W - R;
            }
        } else {
            s_ox = row / W - R, s_oy = row % W - R;
        }
    }
    const bool s_valid_in = lane < nslot && !(lane & 1) && s_zlo <= s_zhi;      // interior cells: part 0 only
    const int s_da = (s_ox * ncy + s_oy) * ncz + s_zlo, s_db = (s_ox * ncy + s_oy) * ncz + s_zhi + 1;
    const int s_code_in = 2 | (2 << 3) | (2 << 6) | (s_filt ? 1024 : 0);
    const bool std_frame_all = frame_ok && !s_general[0] && !s_general[1] && !s_general[2];
    // BLOCKS: two z-adjacent interior cells (cz even) are one job.  Their points are consecutive in the sorted arrays, their
    // stencils differ by one cell in z: one table, one staging of the union (rows one cell taller; row (0, 0): the cells behind
    // the block; the own range = both cells with the test j > i, which also counts every (A, B) pair once) for 8.5 slice
    // points instead of 4.3 - the per-cell overheads halve; what a slice point sees of the extra cell is out of reach and falls
    // to the range test (+ 20 % candidates).
    const bool s_own = a.autocorr && lane == 2 * nrow, s_row00 = a.autocorr && !s_own && (lane >> 1) == nrow - 1;
    const int s_zlo2 = s_own ? 0 : (s_row00 ? 2 : -R), s_zhi2 = s_own ? 1 : R + 1;
    const int s_da2 = (s_ox * ncy + s_oy) * ncz + s_zlo2, s_db2 = (s_ox * ncy + s_oy) * ncz + s_zhi2 + 1;
    const bool s_valid_in2 = lane < nslot && !(lane & 1) && s_zlo2 <= s_zhi2;
    auto fetch = [&](int c1, int cx, int cy, int cz, int nb) {
        Table t;
        t.cbeg = a.start1[c1], t.cend = a.start1[c1 + nb], t.j0 = 0, t.len = 0, t.code = 0;
        if (nb == 2) {        // a block: interior by construction
            if (s_valid_in2) {
                t.j0 = a.start2[c1 + s_da2];
                t.len = (int)(a.start2[c1 + s_db2] - t.j0);
                t.code = s_code_in;
            }
            return t;
        }
        const bool inner = std_frame_all && cx >= R && cx < ncx - R && cy >= R && cy < ncy - R && cz >= R && cz < ncz - R;
        if (inner) {
            if (s_valid_in) {
                t.j0 = a.start2[c1 + s_da];
                t.len = (int)(a.start2[c1 + s_db] - t.j0);
                t.code = s_code_in;
            }
            return t;
        }
        if (lane < nslot) {
            int ox = 0, oy = 0, zlo = -R, zhi = R, filt = 0;
            bool valid = true;
            const int part = lane & 1;
            if (a.autocorr && lane == 2 * nrow) {          // the own cell: pairs j > i
                zlo = zhi = 0, filt = 1;
            } else {
                const int row = lane >> 1;
                if (a.autocorr) {
                    if (row == nrow - 1) zlo = 1;          // row (0, 0): the cells behind the own one
                    else {
                        const int t2 = row + (R * W + R) + 1;
                        ox = t2 / W - R, oy = t2 % W - R;
                    }
                } else {
                    ox = row / W - R, oy = row % W - R;
                }
            }
            int nx = cx + ox, ny = cy + oy, wx = 0, wy = 0;
            if (nx < 0) nx += ncx, wx = -1;
            else if (nx >= ncx) nx -= ncx, wx = 1;
            if (ny < 0) ny += ncy, wy = -1;
            else if (ny >= ncy) ny -= ncy, wy = 1;
            // z cells [cz + zlo, cz + zhi]: part 0 = the piece inside [0, ncz), part 1 = the piece that wraps
            int za = cz + zlo, zb = cz + zhi, wz = 0;
            if (filt) {
                if (part) valid = false;     // (lane 2 nrow is even: never taken, kept for clarity)
            } else if (zlo > zhi) {
                valid = false;
            } else if (!part) {
                za = max(za, 0), zb = min(zb, ncz - 1);
            } else if (za < 0) {
                zb = min(zb, -1) + ncz, za = za + ncz, wz = -1;
            } else if (zb >= ncz) {
                za = max(za, ncz) - ncz, zb = zb - ncz, wz = 1;
            } else {
                valid = false;
            }
            if (valid && za <= zb) {
                const int cA = (nx * ncy + ny) * ncz + za, cB = (nx * ncy + ny) * ncz + zb;
                t.j0 = a.start2[cA];
                t.len = (int)(a.start2[cB + 1] - t.j0);
                const int kx = wx + below(0, cx) - below(0, nx), ky = wy + below(1, cy) - below(1, ny),
                          kz = wz + below(2, cz) - below(2, za);
                bool mixed = !frame_ok || mixed1(0, cx, ncx) || mixed1(0, nx, ncx) || mixed1(1, cy, ncy) || mixed1(1, ny, ncy) ||
                             mixed1(2, cz, ncz);
                for (int zc = za; zc <= zb; zc++) mixed = mixed || mixed1(2, zc, ncz);
                t.code = (kx + 2) | ((ky + 2) << 3) | ((kz + 2) << 6) | (mixed ? 512 : 0) | (filt ? 1024 : 0);
            }
        }
        return t;
    };
    // Jobs of this wave: units (cx, cy, pair of z cells) in steps of the grid's wave count; a unit is ONE job when it is a block
    // (both cells interior, standard frame), else its one or two cells are jobs of their own.  Unit coordinates advance by
    // the decomposed stride (scalar adds and carries, no integer division per job).
    const int nbz = (ncz + 1) >> 1, nunit = ncx * ncy * nbz;
    const int ustride = gridDim.x * P3_WAVES;
    int un = blockIdx.x * P3_WAVES + w;
    const int dub = ustride % nbz, duy = (ustride / nbz) % ncy, dux = ustride / (nbz * ncy);
    int ub = un % nbz, uy = (un / nbz) % ncy, ux = un / (nbz * ncy);
    const bool blocks_on = std_frame_all && !a.no_blocks;
    struct Job {
        int c1, cx, cy, cz, nb;
    };
    Job second;                 // the second cell of a unit that is no block, handed out next
    bool have_second = false;
    auto next_job = [&](Job &j) -> bool {
        if (have_second) {
            have_second = false;
            j = second;
            return true;
        }
        if (un >= nunit) return false;
        const int cz = 2 * ub, nb = min(2, ncz - cz), c1 = (ux * ncy + uy) * ncz + cz;
        const bool blk = nb == 2 && blocks_on && ux >= R && ux < ncx - R && uy >= R && uy < ncy - R && cz >= R && cz + 1 < ncz - R;
        j = Job{c1, ux, uy, cz, blk ? 2 : 1};
        if (nb == 2 && !blk) second = Job{c1 + 1, ux, uy, cz + 1, 1}, have_second = true;
        un += ustride;
        ub += dub;
        int carry = ub >= nbz ? 1 : 0;
        ub -= carry * nbz;
        uy += duy + carry;
        carry = uy >= ncy ? 1 : 0;
        uy -= carry * ncy;
        ux += dux + carry;
        return true;
    };
    Job job;
    bool more = next_job(job);
    Table nxt;
    if (more) nxt = fetch(job.c1, job.cx, job.cy, job.cz, job.nb);
    while (more) {
        const Table cur = nxt;
        more = next_job(job);
        if (more) nxt = fetch(job.c1, job.cx, job.cy, job.cz, job.nb);
        const int64_t cbeg = cur.cbeg, cend = cur.cend;
        if (cbeg == cend) continue;
        wave_sync();   // the previous cell's reads of the segment table are done
        const int64_t j0 = cur.j0;
        const int len = cur.len, code = cur.code;
        // exclusive prefix of the segment lengths over the wave: the neighbour points of the cell form ONE virtual list, which
        // is staged P3_JCAP (320) at a time by all lanes at once (a segment at a time was one memory round trip per segment: 27 to 51
        // dependent round trips per cell, 40 us - the kernel was latency-bound at 18 % of its instruction rate)
        int incl = len;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d, 64);
            if (lane >= d) incl += v;
        }
        const int M = __shfl(incl, 63, 64);
        // the table holds the non-empty segments only, in order (half of the slots - the pieces that wrap in z - are empty for
        // all but the outer cells): 14 of them for the half stencil, so the slot search of a staged point takes 4 steps, not 5
        const bool nz = len > 0;
        const unsigned long long nzmask = __ballot(nz);
        const int nseg = __popcll(nzmask);
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(nzmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)nzmask, 0u));
        const int slot = nz ? rank : nseg + (lane - rank);     // empty ones behind, as sentinels holding the total
        seg_pre[w][slot] = nz ? incl - len : M;
        if (nz) seg_j0[w][slot] = j0, seg_code[w][slot] = code;
        if (lane == 63) seg_pre[w][64] = M;
        const int shi = nseg <= 16 ? 15 : (nseg <= 32 ? 31 : 63);
        wave_sync();
        for (int64_t i0 = cbeg; i0 < cend; i0 += P3_ICAP) {
            const int ni = (int)min((int64_t)P3_ICAP, cend - i0);
            // lane l holds point l of the cell's slice; the pair loop broadcasts point i to the wave with v_readlane (a scalar
            // register: no LDS read, nothing to wait for - PMC had the waves of the LDS-staged form parked in s_waitcnt for
            // 54 % of their cycles at 4 waves per SIMD).  A lane OWNS neighbour points - in the registers it fetched them
            // into - and walks the few points of the slice: no index arithmetic per pair.
            float vix = 0.f, viy = 0.f, viz = 0.f;
            if (lane < ni) vix = a.x1[i0 + lane], viy = a.y1[i0 + lane], viz = a.z1[i0 + lane];
            const int i0i = (int)i0;
            const int niu = __builtin_amdgcn_readfirstlane(ni);
            for (int base = 0; base < M; base += P3_JCAP) {
                const int cnt = min(P3_JCAP, M - base);
                // every lane fetches its points of this round first (all loads in flight together)
                float tx[P3_JCAP / 64], ty[P3_JCAP / 64], tz[P3_JCAP / 64];
                int tg[P3_JCAP / 64], tc[P3_JCAP / 64];
#pragma unroll
                for (int u = 0; u < P3_JCAP / 64; u++) {
                    const int q = u * 64 + lane;
                    tg[u] = -1;
                    tc[u] = 0;
                    tx[u] = ty[u] = tz[u] = 0.f;
                    if (q < cnt) {
                        const int v = base + q;
                        int lo = 0, hi = shi;            // largest slot with seg_pre <= v
                        while (lo < hi) {
                            const int mid = (lo + hi + 1) >> 1;
                            if (seg_pre[w][mid] <= v) lo = mid;
                            else hi = mid - 1;
                        }
                        const int64_t jj = seg_j0[w][lo] + (v - seg_pre[w][lo]);
                        tx[u] = a.x2[jj], ty[u] = a.y2[jj], tz[u] = a.z2[jj];
                        tg[u] = (int)jj, tc[u] = seg_code[w][lo];
                    }
                }
                n_eval += (unsigned long long)ni * (unsigned long long)cnt;
#pragma unroll
                for (int u = 0; u < P3_JCAP / 64; u++) {
                    if (u * 64 >= cnt) break;   // uniform
                    const bool have = tg[u] >= 0;
                    const float xj = tx[u], yj = ty[u], zj = tz[u];
                    const int flags = tc[u];
                    const float sx = (float)(2 - (flags & 7)) * a.g.box;          // -k L, k in {-2 .. 2}
                    const float sy = (float)(2 - ((flags >> 3) & 7)) * a.g.box;
                    const float sz = (float)(2 - ((flags >> 6) & 7)) * a.g.box;
                    // own cell: pairs with i < j - i0 only (every unordered pair once, i ascending); no point: no pairs
                    const int ilim = !have ? 0 : ((flags & 1024) ? min(tg[u] - i0i, ni) : ni);
                    const bool mixed = (flags & 512) != 0;
                    if constexpr (LUT) {
                        // waves none of whose neighbour points needs the per-pair minimum image (all but those near the
                        // frame's cut) run the loop without that code
                        // ... and waves all of whose points come from segments without a periodic shift (every cell away from
                        // the faces of the grid) run it without the shift additions
                        auto run = [&](auto any_mixed, auto no_shift) {
                            constexpr bool MIX = decltype(any_mixed)::value, PLAIN = decltype(no_shift)::value;
                            auto eval = [&](const int i, float &r2o, int &subo, bool &oko) {
                                const int ii = i & 63;
                                const float xi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vix), ii));
                                const float yi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(viy), ii));
                                const float zi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(viz), ii));
                                float dx = xi - xj, dy = yi - yj, dz = zi - zj;
                                if (MIX) {   // branch-free: the minimum image's single addition of 0 or -+L, per lane
                                    const float mx = dx > a.half ? -a.g.box : (dx < -a.half ? a.g.box : 0.f);
                                    const float my = dy > a.half ? -a.g.box : (dy < -a.half ? a.g.box : 0.f);
                                    const float mz = dz > a.half ? -a.g.box : (dz < -a.half ? a.g.box : 0.f);
                                    dx += mixed ? mx : sx, dy += mixed ? my : sy, dz += mixed ? mz : sz;
                                } else if (!PLAIN) {
                                    dx += sx, dy += sy, dz += sz;
                                }
                                const float adz = fabsf(dz);
                                const float r2 = MODE == 1 ? dx * dx + dy * dy : dx * dx + dy * dy + dz * dz;
                                bool ok = i < ilim && r2 >= lo2 && r2 < hi2;
                                int sub = 0;
                                if (MODE == 1) {
                                    sub = (int)(adz / a.dpi);
                                    ok = ok && adz < a.pimax && sub < a.nsub;
                                }
                                if (MODE == 2) {
                                    const float sr = sqrtf(r2);
                                    const float mu = sr > 0.f ? adz / sr : 0.f;
                                    sub = (int)(mu * a.inv_dmu);
                                    ok = ok && mu < a.mu_max && sub < a.nsub;
                                }
                                r2o = r2, subo = sub, oko = ok;
                            };
                            auto look = [&](const float r2, const bool ok) { return lut[ok ? (int)(__float_as_uint(r2) >> lsh) - loff : 0]; };
                            auto count = [&](const float r2, const int sub, const bool ok, const uint2 cell) {
                                const int b = (int)cell.x + (r2 >= __uint_as_float(cell.y) ? 1 : 0);
                                if (ok) atomicAdd(&hist[MODE == 0 ? b : b * a.nsub + sub], 1u);
                            };
                            // one point of the slice per trip (two or four per trip - their table reads in flight together - pad
                            // the slice of 4.4 points on average: 5.8 / 6.6 ms against 5.5; a software-pipelined read: 5.5)
                            for (int i = 0; i < niu; i++) {
                                float r2;
                                int sub;
                                bool ok;
                                eval(i, r2, sub, ok);
                                count(r2, sub, ok, look(r2, ok));
                            }
                        };
                        if (__any(mixed)) run(std::true_type(), std::false_type());
                        else if (__any(have && (flags & 0x1FF) != (2 | (2 << 3) | (2 << 6)))) run(std::false_type(), std::false_type());
                        else run(std::false_type(), std::true_type());
                        continue;
                    }
                    for (int i = 0; i < ni; i++) {   // uniform trip count: readlane needs a wave-uniform index
                        const float xi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vix), i));
                        const float yi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(viy), i));
                        const float zi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(viz), i));
                        if (i >= ilim) continue;
                        float dx = xi - xj, dy = yi - yj, dz = zi - zj;
                        if (mixed) {
                            dx = min_image(dx, a.half, a.g.box);
                            dy = min_image(dy, a.half, a.g.box);
                            dz = min_image(dz, a.half, a.g.box);
                        } else {   // 0 or -+L (or -+2L): the minimum image's own single addition
                            dx += sx, dy += sy, dz += sz;
                        }
                        float r2;
                        int sub = 0;
                        if (MODE == 1) {
                            const float adz = fabsf(dz);
                            if (adz >= a.pimax) continue;
                            r2 = dx * dx + dy * dy;
                            if (r2 < lo2 || r2 >= hi2) continue;
                            sub = (int)(adz / a.dpi);
                            if (sub >= a.nsub) continue;
                        } else {
                            r2 = dx * dx + dy * dy + dz * dz;
                            if (r2 < lo2 || r2 >= hi2) continue;
                        }
                        int b = a.nbins - 1;
                        while (r2 < e2[b]) b--;
                        if (MODE == 2) {
                            const float sr = sqrtf(r2);
                            const float mu = sr > 0.f ? fabsf(dz) / sr : 0.f;
                            if (mu >= a.mu_max) continue;
                            sub = (int)(mu * a.inv_dmu);
                            if (sub >= a.nsub) continue;
                        }
                        atomicAdd(&hist[b * a.nsub + sub], 1u);
                    }
                }
            }
        }
    }
    if (lane == 0 && n_eval) atomicAdd(evaluated, n_eval);
    __syncthreads();
    const unsigned long long mult = a.autocorr ? 2ull : 1ull;   // the half stencil evaluated every unordered pair once
    for (int q = tid; q < nh; q += P3_WAVES * 64)
        if (hist[q]) atomicAdd(&a.npairs[q], mult * (unsigned long long)hist[q]);
}

struct SortedSet {
    DevBuf raw, sorted, counts, cellid, start, keys2, idx, idx2, tmp, packed;
    float *sx, *sy, *sz;
    int64_t n;
};

// float64 device columns (the HOD catalogue) -> float32, as the reference casts before it calls Corrfunc
// (analysis/tpcf_corrfunc.py:134-139: `.astype(np.float32)`, round to nearest)
__global__ void cast_f64_f32(const double *__restrict__ src, float *__restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (float)src[i];
}

// where = 0: host float32 arrays; 1: device float32; 2: device float64
int sort_into_cells(const void *hx, const void *hy, const void *hz, int where, int64_t n, const CellGrid &g, SortedSet &s,
                    DevBuf &scratch, int *d_outside, Frame *d_frame) {
    const int64_t ncell = (int64_t)g.ncx * g.ncy * g.ncz;
    const size_t n1 = (size_t)std::max<int64_t>(n, 1);
    s.n = n;
    ABACUS_TRY(s.sorted.reserve(3 * n1 * 4));
    ABACUS_TRY(s.counts.reserve((size_t)(ncell + 1) * 4));
    ABACUS_TRY(s.cellid.reserve(n1 * 4));
    ABACUS_TRY(s.start.reserve((size_t)(ncell + 1) * 8));
    const float *rx, *ry, *rz;
    s.sx = s.sorted.as<float>();
    s.sy = s.sx + n1;
    s.sz = s.sy + n1;
    const int nblk = (int)std::min<int64_t>(std::max<int64_t>(ceil_div(n, 256), 1), 2048);
    if (where == 1) {   // caller's device arrays, read in place
        rx = (const float *)hx, ry = (const float *)hy, rz = (const float *)hz;
    } else {
        ABACUS_TRY(s.raw.reserve(3 * n1 * 4));
        float *bx = s.raw.as<float>(), *by = bx + n1, *bz = by + n1;
        const void *src[3] = {hx, hy, hz};
        float *dst[3] = {bx, by, bz};
        for (int d = 0; d < 3 && n > 0; d++) {
            if (where == 0) HIP_TRY(hipMemcpyAsync(dst[d], src[d], n * 4, hipMemcpyHostToDevice, stream()));
            else ABACUS_LAUNCH("pair_cast", cast_f64_f32, dim3(nblk), dim3(256), 0, (const double *)src[d], dst[d], n);
        }
        rx = bx, ry = by, rz = bz;
    }
    // large inputs: stable radix sort of (cell id, index) + gather + a search per cell (10^7 points: 0.6 ms against 2.2 ms of
    // the counting sort below, whose two passes are a global atomic per point each)
    if (n >= 200000 && n < ((int64_t)1 << 31) && !option("pairs_countsort")) {
        ABACUS_TRY(s.keys2.reserve(n1 * 4));
        ABACUS_TRY(s.idx.reserve(n1 * 4));
        ABACUS_TRY(s.idx2.reserve(n1 * 4));
        ABACUS_TRY(s.packed.reserve(n1 * 16));
        ABACUS_LAUNCH("pair_cell_count", cell_count<false>, dim3(nblk), dim3(256), 0, rx, ry, rz, n, g, (unsigned int *)nullptr,
                      s.cellid.as<unsigned int>(), s.idx.as<unsigned int>(), s.packed.as<float4>(), d_outside, d_frame);
        int end_bit = 1;
        while (((int64_t)1 << end_bit) < ncell) end_bit++;
        size_t tmp_bytes = 0;
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, s.cellid.as<unsigned int>(), s.keys2.as<unsigned int>(),
                                                 s.idx.as<unsigned int>(), s.idx2.as<unsigned int>(), (int)n, 0, end_bit, stream());
        ABACUS_TRY(s.tmp.reserve(tmp_bytes));
        prof_begin("pair_cell_sort");         // library kernels (rocPRIM radix sort of the cell ids): listed with the hand-written ones
        const hipError_t sorted = hipcub::DeviceRadixSort::SortPairs(s.tmp.p, tmp_bytes, s.cellid.as<unsigned int>(), s.keys2.as<unsigned int>(),
                                                                     s.idx.as<unsigned int>(), s.idx2.as<unsigned int>(), (int)n, 0, end_bit, stream());
        prof_end("pair_cell_sort");
        if (sorted != hipSuccess) return fail("abacus_paircount: radix sort failed");
        ABACUS_LAUNCH("pair_cell_fill", cell_gather, dim3(nblk), dim3(256), 0, s.packed.as<float4>(), n, s.idx2.as<unsigned int>(), s.sx,
                      s.sy, s.sz);
        const int cblk = (int)std::min<int64_t>(ceil_div(ncell + 1, 256), 8192);
        ABACUS_LAUNCH("pair_cell_starts", cell_starts, dim3(cblk), dim3(256), 0, s.keys2.as<unsigned int>(), n, ncell, s.start.as<int64_t>());
        return 0;
    }
    HIP_TRY(hipMemsetAsync(s.counts.p, 0, (size_t)(ncell + 1) * 4, stream()));
    if (n > 0)
        ABACUS_LAUNCH("pair_cell_count", cell_count<true>, dim3(nblk), dim3(256), 0, rx, ry, rz, n, g,
                      s.counts.as<unsigned int>(), s.cellid.as<unsigned int>(), (unsigned int *)nullptr, (float4 *)nullptr, d_outside,
                      d_frame);
    ABACUS_TRY(exclusive_scan_u32(s.counts.as<unsigned int>(), ncell, s.start.as<int64_t>(), scratch, 1));
    if (n > 0)
        ABACUS_LAUNCH("pair_cell_fill", cell_fill, dim3(nblk), dim3(256), 0, rx, ry, rz, n,
                      s.cellid.as<unsigned int>(), s.start.as<int64_t>(), s.counts.as<unsigned int>(), s.sx, s.sy, s.sz);
    return 0;
}

}  // namespace

static unsigned long long g_last_evaluated = 0;   // candidate pairs of the last call (wave-per-cell kernel only)
static int g_last_cells[3] = {0, 0, 0};

static int paircount_impl(int mode, const void *x1, const void *y1, const void *z1, int64_t n1, const void *x2, const void *y2,
                          const void *z2, int64_t n2, int where, float boxsize, const float *bins, int nbins, float pimax,
                          int npibins, float mu_max, int nmubins, uint64_t *npairs) {
    ABACUS_ENTER();
    if (mode < 0 || mode > 2) return fail("abacus_paircount: unknown mode %d", mode);
    if (!x1 || !y1 || !z1 || !bins || !npairs || nbins < 1) return fail("abacus_paircount: null/empty argument");
    if (!(boxsize > 0)) return fail("abacus_paircount: boxsize must be positive");
    const int autocorr = x2 == nullptr;
    const int nsub = mode == 0 ? 1 : (mode == 1 ? npibins : nmubins);
    if (nsub < 1) return fail("abacus_paircount: need at least one pi / mu bin");
    if (nsub > MAX_HIST) return fail("abacus_paircount: %d pi / mu bins exceed the LDS histogram", nsub);
    // One launch bins into at most 63 separation bins (the kernels' edge tables) and MAX_HIST counters in LDS.  More bins -
    // Corrfunc takes any number - are counted in runs of consecutive separation bins, each a call of its own with its own,
    // smaller reach; a pair ON an edge shared by two runs belongs to the upper bin in both ([lo, hi) bins)
    const int run = std::min(63, MAX_HIST / nsub);
    if (nbins > run) {
        for (int a = 0; a < nbins; a += run) {
            const int n = std::min(run, nbins - a);
            ABACUS_TRY(paircount_impl(mode, x1, y1, z1, n1, x2, y2, z2, n2, where, boxsize, bins + a, n, pimax, npibins, mu_max,
                                      nmubins, npairs + (size_t)a * nsub));
        }
        return 0;
    }
    if (n1 >= ((int64_t)1 << 31) || n2 >= ((int64_t)1 << 31)) return fail("abacus_paircount: too many points");
    const float rmax = bins[nbins];
    for (int b = 0; b < nbins; b++)
        if (!(bins[b + 1] > bins[b])) return fail("abacus_paircount: bin edges must increase");
    const float reach_xy = rmax, reach_z = mode == 1 ? pimax : rmax;
    if (reach_xy > 0.5f * boxsize || reach_z > 0.5f * boxsize)
        return fail("abacus_paircount: maximum separation exceeds half the box (minimum image not unique)");
    const size_t ntot = (size_t)nbins * nsub;
    memset(npairs, 0, ntot * sizeof(uint64_t));
    if (n1 == 0 || (!autocorr && n2 == 0)) return 0;

    CellGrid g;
    g.box = boxsize;
    g.inv_box = 1.0f / boxsize;
    // cells of size >= reach / R: R = 2 (125-cell stencil of cells an eighth the volume: 1.7x fewer candidate pairs) when
    // the catalogue is dense enough to keep a wave busy with a small cell, else R = 1
    const int gen = option("pairs_gen") ? option("pairs_gen") : 3;   // comparators: 1, 2 = the older kernels
    auto ncells = [&](float reach, int R, int cap) {
        int nc = (int)floorf(boxsize / reach * (float)R * 0.9999f);   // cell size strictly >= reach / R
        nc = std::min(nc, cap);
        return nc < 3 ? 1 : nc;
    };
    const bool v1 = gen == 1;   // first-generation kernel (comparator of the tests)
    int R = 1;
    if (!v1) {
        const double nmax = (double)std::max(n1, autocorr ? n1 : n2);
        const double per_cell = nmax * ((double)reach_xy / boxsize) * ((double)reach_xy / boxsize) * ((double)reach_z / boxsize);
        if (gen >= 3 && per_cell > 12.0 && ncells(reach_xy, 2, 192) >= 5 && ncells(reach_z, 2, 192) >= 5) R = 2;
    }
    g.ncx = g.ncy = ncells(reach_xy, R, R == 2 ? 192 : 128);
    g.ncz = ncells(reach_z, R, R == 2 ? 192 : 128);
    // the cap may leave cells larger than reach / R: still correct (a cell >= reach / R is all the stencil needs)
    const int64_t ncell = (int64_t)g.ncx * g.ncy * g.ncz;
    const bool use3 = !v1 && gen >= 3 && g.ncx >= 2 * R + 1 && g.ncy >= 2 * R + 1 && g.ncz >= 2 * R + 1;

    static SortedSet S1, S2;
    static DevBuf scratch, d_edges, d_npairs, d_work, d_flag;
    ABACUS_TRY(d_flag.reserve(64));
    HIP_TRY(hipMemsetAsync(d_flag.p, 0, 64, stream()));
    Frame *d_frame = reinterpret_cast<Frame *>(d_flag.as<int>() + 4);
    unsigned long long *d_eval = reinterpret_cast<unsigned long long *>(d_flag.as<int>() + 12);   // byte 48
    {
        Frame f0;
        for (int d = 0; d < 3; d++) f0.mn[d] = 0xffffffffu, f0.mx[d] = 0u;
        HIP_TRY(hipMemcpyAsync(d_frame, &f0, sizeof f0, hipMemcpyHostToDevice, stream()));
        HIP_TRY(hipStreamSynchronize(stream()));   // f0 is a stack object
    }
    ABACUS_TRY(sort_into_cells(x1, y1, z1, where, n1, g, S1, scratch, d_flag.as<int>(), d_frame));
    if (!autocorr) ABACUS_TRY(sort_into_cells(x2, y2, z2, where, n2, g, S2, scratch, d_flag.as<int>(), d_frame));
    SortedSet &T = autocorr ? S1 : S2;

    int nwork = 0, *d_wc = nullptr, *d_wo = nullptr;
    int h_outside = 0;
    if (v1) {   // first-generation kernel: host-built work list, one workgroup per 256 points of a non-empty cell
        std::vector<int64_t> start1((size_t)ncell + 1);
        HIP_TRY(hipMemcpyAsync(start1.data(), S1.start.p, (size_t)(ncell + 1) * 8, hipMemcpyDeviceToHost, stream()));
        HIP_TRY(hipStreamSynchronize(stream()));
        std::vector<int> wc, wo;
        for (int64_t c = 0; c < ncell; c++)
            for (int64_t o = 0; o < start1[c + 1] - start1[c]; o += PB) wc.push_back((int)c), wo.push_back((int)o);
        nwork = (int)wc.size();
        ABACUS_TRY(d_work.reserve((size_t)std::max(nwork, 1) * 8));
        d_wc = d_work.as<int>(), d_wo = d_wc + std::max(nwork, 1);
        HIP_TRY(hipMemcpyAsync(d_wc, wc.data(), (size_t)nwork * 4, hipMemcpyHostToDevice, stream()));
        HIP_TRY(hipMemcpyAsync(d_wo, wo.data(), (size_t)nwork * 4, hipMemcpyHostToDevice, stream()));
        HIP_TRY(hipStreamSynchronize(stream()));
    }

    std::vector<float> e2(nbins + 1);
    for (int b = 0; b <= nbins; b++) e2[b] = bins[b] * bins[b];
    ABACUS_TRY(d_edges.reserve((nbins + 1) * 4));
    ABACUS_TRY(d_npairs.reserve(ntot * 8));
    HIP_TRY(hipMemcpyAsync(d_edges.p, e2.data(), (nbins + 1) * 4, hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipMemsetAsync(d_npairs.p, 0, ntot * 8, stream()));

    PairArgs a;
    a.mode = mode;
    a.autocorr = autocorr;
    a.in_box = !h_outside;   // (first-generation kernel does not use it)
    a.g = g;
    a.nbins = nbins;
    a.nsub = nsub;
    a.half = boxsize * 0.5f;
    a.pimax = pimax;
    a.dpi = npibins > 0 ? pimax / (float)npibins : 1.0f;
    a.mu_max = mu_max;
    a.inv_dmu = nmubins > 0 ? (float)nmubins / mu_max : 1.0f;
    a.edges2 = d_edges.as<float>();
    // cells of the bin table: the coarsest 2^m per octave (m = 2 .. 8) that keep the inner edges apart, at most 8192 cells
    a.lut_sh = a.lut_off = a.lut_ncell = 0;
    a.no_blocks = option("pairs_noblocks");
    if (!option("pairs_nolut") && nbins >= 1 && e2[0] >= 0.f) {
        auto fbits = [](float v) {
            unsigned int u;
            memcpy(&u, &v, 4);
            return u;
        };
        for (int m = 2; m <= 8 && !a.lut_ncell; m++) {
            const int sh = 23 - m;
            const unsigned int c0 = fbits(e2[0]) >> sh, c1 = fbits(e2[nbins]) >> sh;
            if (c1 - c0 + 1 > 8192u) break;
            bool ok = true;
            for (int b = 1; b + 1 < nbins && ok; b++) ok = (fbits(e2[b]) >> sh) != (fbits(e2[b + 1]) >> sh);
            for (int b = 0; b < nbins && ok; b++) ok = e2[b] < e2[b + 1];
            if (ok) a.lut_sh = sh, a.lut_off = (int)c0, a.lut_ncell = (int)(c1 - c0 + 1);
        }
    }
    a.x1 = S1.sx, a.y1 = S1.sy, a.z1 = S1.sz;
    a.x2 = T.sx, a.y2 = T.sy, a.z2 = T.sz;
    a.start1 = S1.start.as<int64_t>();
    a.start2 = T.start.as<int64_t>();
    a.npairs = d_npairs.as<unsigned long long>();
    if (v1) {
        if (nwork > 0) ABACUS_LAUNCH("pair_count", pair_count, dim3(nwork), dim3(PB), 0, a, d_wc, d_wo);
    } else if (use3) {
        int dev = 0, ncu = 256;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        const bool lut = a.lut_ncell > 0;
        const size_t hist_bytes = ((ntot + 1) & ~(size_t)1) * sizeof(unsigned int) + (size_t)a.lut_ncell * sizeof(uint2);
        int per_cu = 4;   // persistent waves: as many workgroups as are resident at once
#define P3_FN(M) (lut ? (const void *)pair_count3<M, true> : (const void *)pair_count3<M, false>)
        const void *fn = mode == 0 ? P3_FN(0) : (mode == 1 ? P3_FN(1) : P3_FN(2));
        if (hist_bytes > 48 * 1024) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hist_bytes));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, P3_WAVES * 64, hist_bytes));
        const dim3 grid((unsigned int)std::min<int64_t>(ceil_div(ncell, P3_WAVES), (int64_t)ncu * std::max(per_cu, 1)));
#define P3_RUN(M)                                                                                                                  \
    do {                                                                                                                           \
        if (lut) ABACUS_LAUNCH("pair_count", (pair_count3<M, true>), grid, dim3(P3_WAVES * 64), hist_bytes, a, d_frame, (int)ncell, R, d_eval); \
        else ABACUS_LAUNCH("pair_count", (pair_count3<M, false>), grid, dim3(P3_WAVES * 64), hist_bytes, a, d_frame, (int)ncell, R, d_eval); \
    } while (0)
        if (mode == 0) P3_RUN(0);
        else if (mode == 1) P3_RUN(1);
        else P3_RUN(2);
#undef P3_RUN
#undef P3_FN
    } else {
        int dev = 0, ncu = 256;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        const dim3 grid((unsigned int)std::min<int64_t>(ncell, (int64_t)ncu * 8));
        const size_t hist_bytes = ntot * sizeof(unsigned int);
        const int *flag = d_flag.as<int>();
        // a few points per cell: stage all neighbour cells together (one round per slice instead of 27)
        const bool agg = (double)T.n / (double)ncell < 12.0;
#define LAUNCH_PC(M)                                                                                         \
    do {                                                                                                     \
        if (agg) ABACUS_LAUNCH("pair_count", (pair_count2<M, true>), grid, dim3(PB), hist_bytes, a, flag, (int)ncell);  \
        else ABACUS_LAUNCH("pair_count", (pair_count2<M, false>), grid, dim3(PB), hist_bytes, a, flag, (int)ncell); \
    } while (0)
        if (mode == 0) LAUNCH_PC(0);
        else if (mode == 1) LAUNCH_PC(1);
        else LAUNCH_PC(2);
#undef LAUNCH_PC
    }
    HIP_TRY(hipMemcpyAsync(npairs, d_npairs.p, ntot * 8, hipMemcpyDeviceToHost, stream()));
    g_last_evaluated = 0;
    HIP_TRY(hipMemcpyAsync(&g_last_evaluated, d_eval, 8, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    g_last_cells[0] = g.ncx, g_last_cells[1] = g.ncz, g_last_cells[2] = use3 ? R : 0;
    return 0;
}

extern "C" int abacus_paircount(int mode, const float *x1, const float *y1, const float *z1, int64_t n1,
                                const float *x2, const float *y2, const float *z2, int64_t n2, float boxsize,
                                const float *bins, int nbins, float pimax, int npibins, float mu_max, int nmubins,
                                uint64_t *npairs) {
    return paircount_impl(mode, x1, y1, z1, n1, x2, y2, z2, n2, 0, boxsize, bins, nbins, pimax, npibins, mu_max, nmubins, npairs);
}

extern "C" int abacus_paircount_dev(int mode, const void *x1, const void *y1, const void *z1, int64_t n1, const void *x2,
                                    const void *y2, const void *z2, int64_t n2, int pos_dtype, float boxsize, const float *bins,
                                    int nbins, float pimax, int npibins, float mu_max, int nmubins, uint64_t *npairs) {
    if (pos_dtype != ABACUS_F32 && pos_dtype != ABACUS_F64) return fail("abacus_paircount_dev: pos_dtype must be ABACUS_F32 or ABACUS_F64");
    return paircount_impl(mode, x1, y1, z1, n1, x2, y2, z2, n2, pos_dtype == ABACUS_F32 ? 1 : 2, boxsize, bins, nbins, pimax,
                          npibins, mu_max, nmubins, npairs);
}

extern "C" int abacus_paircount_stats(uint64_t *candidates, int *ncell_xy, int *ncell_z, int *stencil_R) {
    ABACUS_ENTER();
    if (candidates) *candidates = g_last_evaluated;
    if (ncell_xy) *ncell_xy = g_last_cells[0];
    if (ncell_z) *ncell_z = g_last_cells[1];
    if (stencil_R) *stencil_R = g_last_cells[2];
    return 0;
}
