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
#include <vector>

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

__global__ void cell_count(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z,
                           int64_t n, CellGrid g, unsigned int *__restrict__ counts, unsigned int *__restrict__ cellid,
                           int *__restrict__ outside) {
    bool out = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (cell_coord(x[i], g.ncx, g.inv_box) * g.ncy + cell_coord(y[i], g.ncy, g.inv_box)) * g.ncz +
                      cell_coord(z[i], g.ncz, g.inv_box);
        cellid[i] = (unsigned int)c;
        atomicAdd(&counts[c], 1u);
        out = out || !(x[i] >= 0.f && x[i] < g.box && y[i] >= 0.f && y[i] < g.box && z[i] >= 0.f && z[i] < g.box);
    }
    if (out) *outside = 1;   // some coordinate is not in [0, L): the per-cell periodic image is not known
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

struct PairArgs {
    int mode, autocorr, in_box;   // in_box: every coordinate of both sets lies in [0, L)
    CellGrid g;
    int nbins, nsub;
    float half, pimax, dpi, mu_max, inv_dmu;
    const float *edges2;   // (nbins+1) squared edges
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
template <int MODE>
__global__ __launch_bounds__(PB) void pair_count2(PairArgs a, const int *__restrict__ work_cell,
                                                  const int *__restrict__ work_off) {
    __shared__ float ix[PB], iy[PB], iz[PB];
    __shared__ float jx[PB], jy[PB], jz[PB];
    __shared__ unsigned int hist[MAX_HIST];
    __shared__ float e2[64];
    const int tid = threadIdx.x;
    const int nh = a.nbins * a.nsub;
    for (int q = tid; q < nh; q += PB) hist[q] = 0u;
    if (tid <= a.nbins) e2[tid] = a.edges2[tid];
    const int c1 = work_cell[blockIdx.x];
    const int64_t i0 = a.start1[c1] + work_off[blockIdx.x];
    const int ni = (int)min((int64_t)PB, a.start1[c1 + 1] - i0);
    if (tid < ni) ix[tid] = a.x1[i0 + tid], iy[tid] = a.y1[i0 + tid], iz[tid] = a.z1[i0 + tid];
    __syncthreads();
    const float lo2 = e2[0], hi2 = e2[a.nbins];
    const int cz = c1 % a.g.ncz, cy = (c1 / a.g.ncz) % a.g.ncy, cx = c1 / (a.g.ncz * a.g.ncy);
    const int rx = a.g.ncx >= 3 ? 1 : 0, ry = a.g.ncy >= 3 ? 1 : 0, rz = a.g.ncz >= 3 ? 1 : 0;
    // per-cell periodic shifts need |d| of an unwrapped neighbour pair (< 2 cells) to stay below half the box
    const bool fast = a.in_box && a.g.ncx >= 5 && a.g.ncy >= 5 && a.g.ncz >= 5;
    for (int ox = -rx; ox <= rx; ox++)
        for (int oy = -ry; oy <= ry; oy++)
            for (int oz = -rz; oz <= rz; oz++) {
                int nx = cx + ox, ny = cy + oy, nz = cz + oz;
                float shx = 0.f, shy = 0.f, shz = 0.f;   // xi - xj is about +L when c2 wrapped below 0: subtract L
                if (nx < 0) nx += a.g.ncx, shx = a.g.box;
                else if (nx >= a.g.ncx) nx -= a.g.ncx, shx = -a.g.box;
                if (ny < 0) ny += a.g.ncy, shy = a.g.box;
                else if (ny >= a.g.ncy) ny -= a.g.ncy, shy = -a.g.box;
                if (nz < 0) nz += a.g.ncz, shz = a.g.box;
                else if (nz >= a.g.ncz) nz -= a.g.ncz, shz = -a.g.box;
                // c1 at the low edge (cx = 0) with ox = -1: c2 = ncx - 1, xj ~ L, xi ~ 0: xi - xj ~ -L -> add L
                const int c2 = (nx * a.g.ncy + ny) * a.g.ncz + nz;
                const int64_t j0 = a.start2[c2], j1 = a.start2[c2 + 1];
                for (int64_t jb = j0; jb < j1; jb += PB) {
                    const int m = (int)min((int64_t)PB, j1 - jb);
                    __syncthreads();
                    if (tid < m) jx[tid] = a.x2[jb + tid], jy[tid] = a.y2[jb + tid], jz[tid] = a.z2[jb + tid];
                    __syncthreads();
                    const int total = ni * m;
                    const float inv_m = 1.0f / (float)m;
                    const bool self_chunk = a.autocorr && jb < i0 + ni && jb + m > i0;   // the chunks overlap in the array
                    for (int p = tid; p < total; p += PB) {
                        const int i = (int)(((float)p + 0.5f) * inv_m);   // exact: p < 2^16, m <= 2^8
                        const int j = p - i * m;
                        if (self_chunk && jb + j == i0 + i) continue;      // the same point
                        float dx = ix[i] - jx[j], dy = iy[i] - jy[j], dz = iz[i] - jz[j];
                        if (fast) {
                            dx += shx, dy += shy, dz += shz;
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
                }
            }
    __syncthreads();
    for (int q = tid; q < nh; q += PB)
        if (hist[q]) atomicAdd(&a.npairs[q], (unsigned long long)hist[q]);
}

struct SortedSet {
    DevBuf raw, sorted, counts, cellid, start;
    float *sx, *sy, *sz;
    int64_t n;
};

int sort_into_cells(const float *hx, const float *hy, const float *hz, int64_t n, const CellGrid &g, SortedSet &s,
                    DevBuf &scratch, int *d_outside) {
    const int64_t ncell = (int64_t)g.ncx * g.ncy * g.ncz;
    const size_t n1 = (size_t)std::max<int64_t>(n, 1);
    s.n = n;
    ABACUS_TRY(s.raw.reserve(3 * n1 * 4));
    ABACUS_TRY(s.sorted.reserve(3 * n1 * 4));
    ABACUS_TRY(s.counts.reserve((size_t)(ncell + 1) * 4));
    ABACUS_TRY(s.cellid.reserve(n1 * 4));
    ABACUS_TRY(s.start.reserve((size_t)(ncell + 1) * 8));
    float *rx = s.raw.as<float>(), *ry = rx + n1, *rz = ry + n1;
    s.sx = s.sorted.as<float>();
    s.sy = s.sx + n1;
    s.sz = s.sy + n1;
    HIP_TRY(hipMemcpyAsync(rx, hx, n * 4, hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipMemcpyAsync(ry, hy, n * 4, hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipMemcpyAsync(rz, hz, n * 4, hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipMemsetAsync(s.counts.p, 0, (size_t)(ncell + 1) * 4, stream()));
    const int nblk = (int)std::min<int64_t>(std::max<int64_t>(ceil_div(n, 256), 1), 4096);
    if (n > 0)
        ABACUS_LAUNCH("pair_cell_count", cell_count, dim3(nblk), dim3(256), 0, rx, ry, rz, n, g,
                      s.counts.as<unsigned int>(), s.cellid.as<unsigned int>(), d_outside);
    ABACUS_TRY(exclusive_scan_u32(s.counts.as<unsigned int>(), ncell, s.start.as<int64_t>(), scratch, 1));
    if (n > 0)
        ABACUS_LAUNCH("pair_cell_fill", cell_fill, dim3(nblk), dim3(256), 0, rx, ry, rz, n,
                      s.cellid.as<unsigned int>(), s.start.as<int64_t>(), s.counts.as<unsigned int>(), s.sx, s.sy, s.sz);
    return 0;
}

}  // namespace

extern "C" int abacus_paircount(int mode, const float *x1, const float *y1, const float *z1, int64_t n1,
                                const float *x2, const float *y2, const float *z2, int64_t n2, float boxsize,
                                const float *bins, int nbins, float pimax, int npibins, float mu_max, int nmubins,
                                uint64_t *npairs) {
    ABACUS_TRY(ensure_init());
    if (mode < 0 || mode > 2) return fail("abacus_paircount: unknown mode %d", mode);
    if (!x1 || !y1 || !z1 || !bins || !npairs || nbins < 1) return fail("abacus_paircount: null/empty argument");
    if (nbins > 63) return fail("abacus_paircount: more than 63 separation bins");
    if (!(boxsize > 0)) return fail("abacus_paircount: boxsize must be positive");
    const int autocorr = x2 == nullptr;
    const int nsub = mode == 0 ? 1 : (mode == 1 ? npibins : nmubins);
    if (nsub < 1) return fail("abacus_paircount: need at least one pi / mu bin");
    if ((int64_t)nbins * nsub > MAX_HIST) return fail("abacus_paircount: %d x %d bins exceed the LDS histogram", nbins, nsub);
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
    auto ncells = [&](float reach) {
        int nc = (int)floorf(boxsize / reach * 0.9999f);   // cell size strictly >= reach
        nc = std::min(nc, 128);
        return nc < 3 ? 1 : nc;
    };
    g.ncx = g.ncy = ncells(reach_xy);
    g.ncz = ncells(reach_z);
    const int64_t ncell = (int64_t)g.ncx * g.ncy * g.ncz;

    static SortedSet S1, S2;
    static DevBuf scratch, d_edges, d_npairs, d_work, d_flag;
    ABACUS_TRY(d_flag.reserve(64));
    HIP_TRY(hipMemsetAsync(d_flag.p, 0, sizeof(int), stream()));
    ABACUS_TRY(sort_into_cells(x1, y1, z1, n1, g, S1, scratch, d_flag.as<int>()));
    if (!autocorr) ABACUS_TRY(sort_into_cells(x2, y2, z2, n2, g, S2, scratch, d_flag.as<int>()));
    SortedSet &T = autocorr ? S1 : S2;

    // work list: one workgroup per 256 points of every non-empty cell of set 1 (host side: ncell <= 2M)
    std::vector<int64_t> start1((size_t)ncell + 1);
    int h_outside = 0;
    HIP_TRY(hipMemcpyAsync(start1.data(), S1.start.p, (size_t)(ncell + 1) * 8, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipMemcpyAsync(&h_outside, d_flag.p, sizeof(int), hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    std::vector<int> work;
    for (int64_t c = 0; c < ncell; c++)
        for (int64_t o = 0; o < start1[c + 1] - start1[c]; o += PB) {
            work.push_back((int)c);
            work.push_back((int)o);
        }
    const int nwork = (int)(work.size() / 2);
    std::vector<int> wc(nwork), wo(nwork);
    for (int q = 0; q < nwork; q++) wc[q] = work[2 * q], wo[q] = work[2 * q + 1];
    ABACUS_TRY(d_work.reserve((size_t)std::max(nwork, 1) * 8));
    int *d_wc = d_work.as<int>(), *d_wo = d_wc + std::max(nwork, 1);
    HIP_TRY(hipMemcpyAsync(d_wc, wc.data(), (size_t)nwork * 4, hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipMemcpyAsync(d_wo, wo.data(), (size_t)nwork * 4, hipMemcpyHostToDevice, stream()));

    std::vector<float> e2(nbins + 1);
    for (int b = 0; b <= nbins; b++) e2[b] = bins[b] * bins[b];
    ABACUS_TRY(d_edges.reserve((nbins + 1) * 4));
    ABACUS_TRY(d_npairs.reserve(ntot * 8));
    HIP_TRY(hipMemcpyAsync(d_edges.p, e2.data(), (nbins + 1) * 4, hipMemcpyHostToDevice, stream()));
    HIP_TRY(hipMemsetAsync(d_npairs.p, 0, ntot * 8, stream()));

    PairArgs a;
    a.mode = mode;
    a.autocorr = autocorr;
    a.in_box = !h_outside;
    a.g = g;
    a.nbins = nbins;
    a.nsub = nsub;
    a.half = boxsize * 0.5f;
    a.pimax = pimax;
    a.dpi = npibins > 0 ? pimax / (float)npibins : 1.0f;
    a.mu_max = mu_max;
    a.inv_dmu = nmubins > 0 ? (float)nmubins / mu_max : 1.0f;
    a.edges2 = d_edges.as<float>();
    a.x1 = S1.sx, a.y1 = S1.sy, a.z1 = S1.sz;
    a.x2 = T.sx, a.y2 = T.sy, a.z2 = T.sz;
    a.start1 = S1.start.as<int64_t>();
    a.start2 = T.start.as<int64_t>();
    a.npairs = d_npairs.as<unsigned long long>();
    if (nwork > 0) {
        if (getenv("ABACUS_PAIRS_V1")) ABACUS_LAUNCH("pair_count", pair_count, dim3(nwork), dim3(PB), 0, a, d_wc, d_wo);
        else if (mode == 0) ABACUS_LAUNCH("pair_count", (pair_count2<0>), dim3(nwork), dim3(PB), 0, a, d_wc, d_wo);
        else if (mode == 1) ABACUS_LAUNCH("pair_count", (pair_count2<1>), dim3(nwork), dim3(PB), 0, a, d_wc, d_wo);
        else ABACUS_LAUNCH("pair_count", (pair_count2<2>), dim3(nwork), dim3(PB), 0, a, d_wc, d_wo);
    }
    HIP_TRY(hipMemcpyAsync(npairs, d_npairs.p, ntot * 8, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}
