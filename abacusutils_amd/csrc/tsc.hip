// TSC / CIC particle-to-mesh deposit on MI355X (gfx950): replaces tsc_parallel and its helpers
// (abacusnbody/analysis/tsc.py:10-206 _wrap_inplace :219-226, partition_parallel :259-384, _tsc_scatter :394-507)
// and cic_serial (analysis/cic.py:13-125).
//
// The reference avoids write races with x-stripes processed even-then-odd by CPU threads.  Here the mesh is cut
// into LDS-sized tiles (16 x 16 x 32 cells = 64 KiB of float64 accumulators, two workgroups per CU):
//   (lists for >= 2e6 particles are built by the two-level LDS multisplit ms_coarse / ms_fine instead of tsc_bin)
//   tsc_bin<COUNT>   one pass over the particles: wrap (in place, like the reference), find the tiles the 3x3x3
//                    cloud touches (1..8, 1.3 on average) and count them per tile (integer atomics, L2);
//   scan.hip         exclusive scan of the tile counts;
//   tsc_bin<FILL>    second pass: append (x, y, z, w) to every touched tile's list (16-B stores);
//   tsc_tile_deposit one workgroup per tile: zero the tile in LDS, accumulate the list with LDS float atomics
//                    (contributions falling outside the tile are dropped - the neighbour tile has its own copy of
//                    the particle), then write the tile to HBM with plain coalesced 128-B row stores.  No global
//                    atomics, no separate zeroing pass, optional fused normalisation delta = rho*norm - 1
//                    (normalize_field, analysis/power_spectrum.py:860-901).
// HBM traffic: 2 x 12 B/particle reads + ~1.3 x 16 B/particle list write+read + 4 B/cell mesh write.
//
// Arithmetic follows _tsc_scatter exactly (math in the position dtype, round-half-even, int16 cell index,
// weights 0.75-d^2 and 0.5(0.5+-d)^2, product order wx*wy*wz*W; built with -ffp-contract=off); only the order in
// which contributions are summed into a cell differs (unordered LDS atomics vs the reference's particle order),
// a float32 rounding-level effect covered by the reference's own tolerance (tests/test_tsc.py:136).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "../../include/abacus_hip.h"
#include "common.hpp"

using namespace abacus;

namespace abacus {
int exclusive_scan_u32(unsigned int *counters, int64_t n, int64_t *out, DevBuf &scratch, int zero_counters);
}

namespace {

constexpr int TSC_BLOCK = 256;

struct TileGeom {
    int gx, gy, gz;       // mesh
    int tx, ty, tz;       // tile shape (cells)
    int ntx, nty, ntz;    // tiles per dimension
    int64_t zstride;      // elements per z-row in memory (gz, or a padded pitch for the in-place R2C layout)
    int gxg, xoff;        // x-slab meshes: global x size (scale + periodic wrap) and global index of local plane 0;
                          // gxg == gx, xoff == 0 for a full mesh
    int xoff2, xwin;      // folded slabs (two windows of xwin = gx / 2 planes each, back to back in memory): global index
                          // of the second window's first plane; xoff2 < 0: one window of gx planes
    int shx, shy, shz;    // log2 of the tile shape where it is a power of two (the usual 16 x 16 x 32), else -1:
                          // cell -> tile by a shift instead of a runtime integer division (nine per particle and pass)
    int f4x, f4y, f4z;    // four consecutive (periodically wrapped) cells always lie in at most two tiles of this
                          // dimension: tiles of >= 4 cells, the last one included (x: full meshes only, not x-slabs)
};

__device__ __forceinline__ int tile_of_cell(int c, int t, int sh) { return sh >= 0 ? c >> sh : c / t; }

__device__ __forceinline__ int wrapcell(int c, int g) {
    // _rightwrap (tsc.py:387-391) + NumPy negative indexing; general modulo for far-out-of-box positions
    if (c >= g) {
        c -= g;
        if (c >= g) c %= g;
    } else if (c < 0) {
        c += g;
        if (c < 0) c = ((c % g) + g) % g;
    }
    return c;
}

// global x cell -> local plane index of an x-slab mesh (-1: not held by this slab).  Two windows: a plane both hold (their
// ghost regions may overlap when the slabs are thin) belongs to the first, so every contribution is written exactly once
__device__ __forceinline__ int xloc(int i, const TileGeom &g) {
    const int c = wrapcell(i, g.gxg);
    int l = c - g.xoff;
    if (l < 0) l += g.gxg;
    if (g.xoff2 < 0) return l < g.gx ? l : -1;
    if (l < g.xwin) return l;
    l = c - g.xoff2;
    if (l < 0) l += g.gxg;
    return l < g.xwin ? g.xwin + l : -1;
}

template <typename PT>
struct Cloud {
    int i[3];      // nearest cell per dimension (unwrapped, int16 range like the reference)
    PT w[3][3];    // [dim][-1,0,+1]
};

template <typename PT>
__device__ __forceinline__ PT rnd(PT x);
template <>
__device__ __forceinline__ float rnd<float>(float x) { return rintf(x); }
template <>
__device__ __forceinline__ double rnd<double>(double x) { return rint(x); }

// TSC weights, _tsc_scatter (tsc.py:419-451)
template <typename PT>
__device__ __forceinline__ void tsc_cloud(PT x, PT y, PT z, PT offset, PT ihx, PT ihy, PT ihz, Cloud<PT> &c) {
    const PT HALF = (PT)0.5, P75 = (PT)0.75;
    PT p[3] = {(x + offset) * ihx, (y + offset) * ihy, (z + offset) * ihz};
#pragma unroll
    for (int a = 0; a < 3; a++) {
        int i = (int)(short)rnd<PT>(p[a]);
        PT d = (PT)i - p[a];
        PT tm = HALF + d, tp = HALF - d;
        c.i[a] = i;
        c.w[a][1] = P75 - d * d;
        c.w[a][0] = HALF * (tm * tm);
        c.w[a][2] = HALF * (tp * tp);
    }
}

// CIC weights, cic_serial (cic.py:29-71): float64 math whatever the position dtype, positions NOT wrapped
template <typename PT>
__device__ __forceinline__ void cic_cloud(PT x, PT y, PT z, double box, int gx, int gy, int gz, Cloud<double> &c) {
    double p[3] = {((double)x / box) * gx, ((double)y / box) * gy, ((double)z / box) * gz};
#pragma unroll
    for (int a = 0; a < 3; a++) {
        int i = (int)rint(p[a]);
        double d = i - p[a];
        c.i[a] = i;
        c.w[a][1] = 1.0 - fabs(d);
        if (d > 0.0) {
            c.w[a][0] = d;
            c.w[a][2] = 0.0;
        } else {
            c.w[a][2] = -d;
            c.w[a][0] = 0.0;
        }
    }
}

// x: like tiles_1d below, on local plane indices; planes outside the slab are skipped
__device__ __forceinline__ int tiles_1d_x(int i, const TileGeom &g, int out[3]) {
    int n = 0;
#pragma unroll
    for (int a = -1; a <= 1; a++) {
        const int l = xloc(i + a, g);
        if (l < 0) continue;
        const int t = tile_of_cell(l, g.tx, g.shx);
        bool dup = false;
        for (int q = 0; q < n; q++) dup = dup || out[q] == t;
        if (!dup) out[n++] = t;
    }
    return n;
}

// distinct tiles touched along one dimension by cells i-1, i, i+1 (2 at most unless the mesh is tiny)
__device__ __forceinline__ int tiles_1d(int i, int g, int t, int sh, int out[3]) {
    int a = tile_of_cell(wrapcell(i - 1, g), t, sh), b = tile_of_cell(wrapcell(i, g), t, sh),
        c = tile_of_cell(wrapcell(i + 1, g), t, sh);
    int n = 1;
    out[0] = a;
    if (b != a) out[n++] = b;
    if (c != a && c != b) out[n++] = c;
    return n;
}

// the same with the cell range i-1 .. i+1+ext: ext = 1 covers the clouds of BOTH deposits of an interlaced pair (offset 0
// and half a cell: the nearest cell moves by 0 or +1), so one set of lists serves both
__device__ __forceinline__ int tiles_1d_x_ext(int i, const TileGeom &g, int ext, int out[4]) {
    if (g.f4x) {   // the end cells decide
        const int a = tile_of_cell(wrapcell(i - 1, g.gx), g.tx, g.shx), b = tile_of_cell(wrapcell(i + 1 + ext, g.gx), g.tx, g.shx);
        out[0] = a, out[1] = b;
        return a == b ? 1 : 2;
    }
    int n = 0;
    for (int a = -1; a <= 1 + ext; a++) {
        const int l = xloc(i + a, g);
        if (l < 0) continue;
        const int t = tile_of_cell(l, g.tx, g.shx);
        bool dup = false;
        for (int q = 0; q < n; q++) dup = dup || out[q] == t;
        if (!dup) out[n++] = t;
    }
    return n;
}
__device__ __forceinline__ int tiles_1d_ext(int i, int g, int t, int sh, int ext, int f4, int out[4]) {
    if (f4) {
        const int a = tile_of_cell(wrapcell(i - 1, g), t, sh), b = tile_of_cell(wrapcell(i + 1 + ext, g), t, sh);
        out[0] = a, out[1] = b;
        return a == b ? 1 : 2;
    }
    int n = 0;
    for (int a = -1; a <= 1 + ext; a++) {
        const int tt = tile_of_cell(wrapcell(i + a, g), t, sh);
        bool dup = false;
        for (int q = 0; q < n; q++) dup = dup || out[q] == tt;
        if (!dup) out[n++] = tt;
    }
    return n;
}

// FAST paths (template flag of the list-build and tile kernels): a FULL mesh (no x-slab offset) whose tile shape is a
// power of two and satisfies the four-cells-two-tiles rule in every dimension - the 16 x 16 x 32 tiles of every
// calc_power mesh that is a multiple of 32.  Cell -> tile is a shift, the periodic wrap one compare-and-add each way
// (positions were wrapped into [0, L): the nearest cell lies in [0, g], its neighbours in [-1, g + 2]), the tile pair of a
// dimension comes from its two end cells.  The generic code - runtime divisions, x-slab offsets, modulo wraps for
// far-out-of-box positions, duplicate scans - compiled to 6 300 instructions with 136 integer-division sequences and 460
// branches in ms_coarse alone; garbage (non-finite) positions are clamped into the mesh here instead of taking a modulo.
__device__ __forceinline__ int fast_wrap(int c, int g) {
    c = c < 0 ? c + g : (c >= g ? c - g : c);
    return min(max(c, 0), g - 1);
}
// SLAB (FAST == 2): an x-slab mesh (one or two windows of planes, TileGeom::xoff / xoff2) - y and z as above, along x every
// cell of the cloud is mapped to its local plane (xloc; cells the slab does not hold drop out) and consecutive cells in the
// same tile collapse: the list build of a rank of the multi-GPU estimator, which the generic code served at half the rate
template <typename PT, bool SLAB, typename F>
__device__ __forceinline__ void for_each_tile_fast(PT x, PT y, PT z, const TileGeom &g, PT offset, PT ihx, PT ihy, PT ihz, int ext,
                                                   F f) {
    Cloud<PT> c;
    tsc_cloud<PT>(x, y, z, offset, ihx, ihy, ihz, c);
    if constexpr (SLAB) {
        const int ya = fast_wrap(c.i[1] - 1, g.gy) >> g.shy, yb = fast_wrap(c.i[1] + 1 + ext, g.gy) >> g.shy;
        const int za = fast_wrap(c.i[2] - 1, g.gz) >> g.shz, zb = fast_wrap(c.i[2] + 1 + ext, g.gz) >> g.shz;
        const bool dy = ya != yb, dz = za != zb;
        int last = -1;
#pragma unroll
        for (int a = -1; a <= 2; a++) {
            if (a == 2 && !ext) break;
            const int l = xloc(c.i[0] + a, g);
            if (l < 0) continue;
            const int t = l >> g.shx;
            if (t == last) continue;
            last = t;
            const int ra = (t * g.nty + ya) * g.ntz, rb = (t * g.nty + yb) * g.ntz;
            f((unsigned int)(ra + za));
            if (dz) f((unsigned int)(ra + zb));
            if (dy) {
                f((unsigned int)(rb + za));
                if (dz) f((unsigned int)(rb + zb));
            }
        }
        return;
    }
    const int xa = fast_wrap(c.i[0] - 1, g.gx) >> g.shx, xb = fast_wrap(c.i[0] + 1 + ext, g.gx) >> g.shx;
    const int ya = fast_wrap(c.i[1] - 1, g.gy) >> g.shy, yb = fast_wrap(c.i[1] + 1 + ext, g.gy) >> g.shy;
    const int za = fast_wrap(c.i[2] - 1, g.gz) >> g.shz, zb = fast_wrap(c.i[2] + 1 + ext, g.gz) >> g.shz;
    const int ra = (xa * g.nty + ya) * g.ntz, rb = (xa * g.nty + yb) * g.ntz, rc = (xb * g.nty + ya) * g.ntz,
              rd = (xb * g.nty + yb) * g.ntz;
    const bool dx = xa != xb, dy = ya != yb, dz = za != zb;
    f((unsigned int)(ra + za));
    if (dz) f((unsigned int)(ra + zb));
    if (dy) {
        f((unsigned int)(rb + za));
        if (dz) f((unsigned int)(rb + zb));
    }
    if (dx) {
        f((unsigned int)(rc + za));
        if (dz) f((unsigned int)(rc + zb));
        if (dy) {
            f((unsigned int)(rd + za));
            if (dz) f((unsigned int)(rd + zb));
        }
    }
}

template <typename PT>
struct Entry {
    PT x, y, z, w;
};

// _wrap_inplace (tsc.py:219-226): one +-box shift, compared/shifted in float64 (Numba promotion), stored back as PT
template <typename PT>
__device__ __forceinline__ PT wrap1(PT v, double box, bool &changed) {
    if ((double)v >= box) {
        changed = true;
        return (PT)((double)v - box);
    }
    if (v < 0) {
        changed = true;
        return (PT)((double)v + box);
    }
    return v;
}

// pass over the particles: FILL=false counts list lengths (and wraps in place), FILL=true appends entries
template <typename PT, bool FILL, bool CIC>
__global__ __launch_bounds__(TSC_BLOCK) void tsc_bin(PT *__restrict__ pos, int64_t n, const PT *__restrict__ weights,
                                                     TileGeom g, double box, double offset_, int wrap,
                                                     unsigned int *__restrict__ tile_count,
                                                     const int64_t *__restrict__ tile_start,
                                                     Entry<PT> *__restrict__ entries, int *__restrict__ wrapped_flag) {
    const PT ihx = (PT)(g.gxg / box), ihy = (PT)(g.gy / box), ihz = (PT)(g.gz / box);
    const PT offset = (PT)offset_;
    bool any_changed = false;
    for (int64_t p = (int64_t)blockIdx.x * TSC_BLOCK + threadIdx.x; p < n; p += (int64_t)gridDim.x * TSC_BLOCK) {
        PT x = pos[3 * p], y = pos[3 * p + 1], z = pos[3 * p + 2];
        if (!FILL && wrap) {
            bool ch = false;
            x = wrap1(x, box, ch);
            y = wrap1(y, box, ch);
            z = wrap1(z, box, ch);
            if (ch) {
                pos[3 * p] = x;
                pos[3 * p + 1] = y;
                pos[3 * p + 2] = z;
                any_changed = true;
            }
        }
        int ci[3];
        if (CIC) {
            Cloud<double> c;
            cic_cloud<PT>(x + offset, y + offset, z + offset, box, g.gxg, g.gy, g.gz, c);
            ci[0] = c.i[0], ci[1] = c.i[1], ci[2] = c.i[2];
        } else {
            Cloud<PT> c;
            tsc_cloud<PT>(x, y, z, offset, ihx, ihy, ihz, c);
            ci[0] = c.i[0], ci[1] = c.i[1], ci[2] = c.i[2];
        }
        int ax[3], ay[3], az[3];
        const int nx = tiles_1d_x(ci[0], g, ax), ny = tiles_1d(ci[1], g.gy, g.ty, g.shy, ay),
                  nz = tiles_1d(ci[2], g.gz, g.tz, g.shz, az);
        const PT w = (FILL && weights) ? weights[p] : (PT)1;
        for (int a = 0; a < nx; a++)
            for (int b = 0; b < ny; b++)
                for (int c = 0; c < nz; c++) {
                    const int tile = (ax[a] * g.nty + ay[b]) * g.ntz + az[c];
                    const unsigned int slot = atomicAdd(&tile_count[tile], 1u);
                    if (FILL) entries[tile_start[tile] + slot] = Entry<PT>{x, y, z, w};
                }
    }
    if (!FILL && any_changed) *wrapped_flag = 1;
}

// ---- two-level multisplit (large particle counts) -------------------------------------------------------------
// tsc_bin above costs one scattered global atomic per list entry (~2.4e10/s on MI355X whatever the address pattern).
// For large inputs the lists are built by an MSD multisplit instead: a coarse pass over <= 1024 groups of tiles and a
// fine pass inside each group, both with workgroup-private LDS histograms, so global atomics are issued once per
// (workgroup, bucket) instead of once per entry.  Order inside a list is arbitrary (the float64 tile accumulation
// makes the mesh independent of it at the 1e-16 level).
constexpr int MS_BLOCK = 512;
constexpr int MS_CHUNK = 32768;    // particles per workgroup in the coarse passes
constexpr int MS_FCHUNK = 32768;   // list entries per workgroup in the fine passes
constexpr int MS_BINS = 1024;
// Scatter passes sort a SUB-CHUNK of entries by bucket in LDS before they write it: PMC showed every scattered 16-B entry
// and every 4-B key leaving the L2 as its own 32-B sector (8.2 GB written for 2.7 GB of entries + keys in the coarse pass
// at 1e8 particles, 72 % of the wave cycles stalled on the store issue).  Sorted, the entries of a bucket are written by
// adjacent lanes of one store instruction and share sectors: coarse 3.16 -> 2.7 ms, fine 1.77 -> 1.65 ms at 1024^3 (2.9 -> 2.66
// and 2.45 -> 2.2 at 2048^3).  The split of the tiles between the passes does not matter (1024 x 128 ... 128 x 1024 coarse
// buckets x tiles per bucket at 1024^3: 4.3 - 4.6 ms for the two scatters together).
template <typename PT>
struct MsSub {                     // sub-chunk sizes: the kernel's static LDS stays under 64 KB (two workgroups per CU)
    static constexpr int E = sizeof(PT) == 4 ? 1920 : 1024;   // entries (float: 30 KB of entries + 11 KB of keys and bucket ids)
    static constexpr int P = E * 5 / 8;                       // particles of the coarse pass (1.35 entries per particle on average)
};

// exclusive scan of cnt[0..MS_BINS) into start[] by the MS_BLOCK threads of the workgroup (two bins per thread)
__device__ __forceinline__ void ms_block_scan(const unsigned int *cnt, unsigned int *start, unsigned int *wave_tot) {
    static_assert(MS_BINS == 2 * MS_BLOCK, "two bins per thread");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned int a = cnt[2 * tid], b = cnt[2 * tid + 1];
    unsigned int incl = a + b;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    unsigned int before = 0;
#pragma unroll
    for (int w = 0; w < MS_BLOCK / 64; w++)
        if (w < wv) before += wave_tot[w];
    const unsigned int excl = before + incl - (a + b);
    start[2 * tid] = excl;
    start[2 * tid + 1] = excl + a;
}

template <typename PT, bool CIC, typename F>
__device__ __forceinline__ void for_each_tile(PT x, PT y, PT z, const TileGeom &g, double box, PT offset, PT ihx, PT ihy,
                                              PT ihz, int ext, F f) {
    int ci[3];
    if (CIC) {
        Cloud<double> c;
        cic_cloud<PT>(x + offset, y + offset, z + offset, box, g.gxg, g.gy, g.gz, c);
        ci[0] = c.i[0], ci[1] = c.i[1], ci[2] = c.i[2];
    } else {
        Cloud<PT> c;
        tsc_cloud<PT>(x, y, z, offset, ihx, ihy, ihz, c);
        ci[0] = c.i[0], ci[1] = c.i[1], ci[2] = c.i[2];
    }
    int ax[4], ay[4], az[4];
    int nx, ny, nz;
    if (ext) {
        nx = tiles_1d_x_ext(ci[0], g, ext, ax), ny = tiles_1d_ext(ci[1], g.gy, g.ty, g.shy, ext, g.f4y, ay),
        nz = tiles_1d_ext(ci[2], g.gz, g.tz, g.shz, ext, g.f4z, az);
    } else {
        nx = tiles_1d_x(ci[0], g, ax), ny = tiles_1d(ci[1], g.gy, g.ty, g.shy, ay), nz = tiles_1d(ci[2], g.gz, g.tz, g.shz, az);
    }
    for (int a = 0; a < nx; a++)
        for (int b = 0; b < ny; b++)
            for (int c = 0; c < nz; c++) f((unsigned int)((ax[a] * g.nty + ay[b]) * g.ntz + az[c]));
}

// coarse pass over the particles: SCATTER=false counts entries per coarse bucket (and wraps in place),
// SCATTER=true writes (entry, tile id) grouped by coarse bucket
template <typename PT, bool CIC, bool SCATTER, int FAST>
__global__ __launch_bounds__(MS_BLOCK) void ms_coarse(PT *__restrict__ pos, int64_t n, const PT *__restrict__ weights,
                                                      TileGeom g, double box, double offset_, int wrap, int cshift,
                                                      int ncoarse, unsigned int *__restrict__ gcount,
                                                      const int64_t *__restrict__ gstart,
                                                      Entry<PT> *__restrict__ stage_entry,
                                                      unsigned int *__restrict__ stage_key, int *__restrict__ wrapped_flag,
                                                      int ext) {
    __shared__ unsigned int hist[MS_BINS];
    __shared__ int64_t base[MS_BINS];
    const int tid = threadIdx.x;
    for (int b = tid; b < MS_BINS; b += MS_BLOCK) hist[b] = 0u;
    __syncthreads();
    const PT ihx = (PT)(g.gxg / box), ihy = (PT)(g.gy / box), ihz = (PT)(g.gz / box);
    const PT offset = (PT)offset_;
    const int64_t p0 = (int64_t)blockIdx.x * MS_CHUNK, p1 = min(p0 + MS_CHUNK, n);
    bool any_changed = false;
    for (int64_t p = p0 + tid; p < p1; p += MS_BLOCK) {
        PT x = pos[3 * p], y = pos[3 * p + 1], z = pos[3 * p + 2];
        if (!SCATTER && wrap) {
            bool ch = false;
            x = wrap1(x, box, ch);
            y = wrap1(y, box, ch);
            z = wrap1(z, box, ch);
            if (ext) {   // lists shared with the second deposit of an interlaced pair: the reference wraps again before it
                x = wrap1(x, box, ch);
                y = wrap1(y, box, ch);
                z = wrap1(z, box, ch);
            }
            if (ch) {
                pos[3 * p] = x;
                pos[3 * p + 1] = y;
                pos[3 * p + 2] = z;
                any_changed = true;
            }
        }
        auto count = [&](unsigned int tile) { atomicAdd(&hist[tile >> cshift], 1u); };
        if constexpr (FAST != 0) for_each_tile_fast<PT, FAST == 2>(x, y, z, g, offset, ihx, ihy, ihz, ext, count);
        else for_each_tile<PT, CIC>(x, y, z, g, box, offset, ihx, ihy, ihz, ext, count);
    }
    if (!SCATTER && any_changed) *wrapped_flag = 1;
    __syncthreads();
    if constexpr (!SCATTER) {
        for (int b = tid; b < ncoarse; b += MS_BLOCK)
            if (hist[b]) atomicAdd(&gcount[b], hist[b]);
        return;
    } else {
    constexpr int MS_SUBE = MsSub<PT>::E, MS_SUBP = MsSub<PT>::P;
    // reserve this workgroup's slice of every bucket; hist[] then counts what the workgroup has already written to it
    for (int b = tid; b < MS_BINS; b += MS_BLOCK) {
        const unsigned int c = b < ncoarse ? hist[b] : 0u;
        base[b] = c ? gstart[b] + (int64_t)atomicAdd(&gcount[b], c) : 0;
        hist[b] = 0u;
    }
    // sub-chunks: count per bucket, scan, place into LDS sorted by bucket, write out with adjacent lanes on adjacent entries
    __shared__ unsigned int lstart[MS_BINS], lcur[MS_BINS], wave_tot[MS_BLOCK / 64];
    __shared__ Entry<PT> buf[MS_SUBE];
    __shared__ unsigned int bkey[MS_SUBE];
    __shared__ unsigned short bid[MS_SUBE];
    constexpr int PPT = (MS_SUBP + MS_BLOCK - 1) / MS_BLOCK;   // particles per thread and sub-chunk
    for (int64_t s0 = p0; s0 < p1; s0 += MS_SUBP) {
        const int64_t s1 = min(s0 + MS_SUBP, p1);
        for (int b = tid; b < MS_BINS; b += MS_BLOCK) lcur[b] = 0u;
        __syncthreads();
        PT px[PPT], py[PPT], pz[PPT], pw[PPT];
#pragma unroll
        for (int q = 0; q < PPT; q++) {
            const int64_t p = s0 + tid + (int64_t)q * MS_BLOCK;
            if (p < s1) {
                px[q] = pos[3 * p], py[q] = pos[3 * p + 1], pz[q] = pos[3 * p + 2];
                pw[q] = weights ? weights[p] : (PT)1;
                auto count = [&](unsigned int tile) { atomicAdd(&lcur[tile >> cshift], 1u); };
                if constexpr (FAST != 0) for_each_tile_fast<PT, FAST == 2>(px[q], py[q], pz[q], g, offset, ihx, ihy, ihz, ext, count);
                else for_each_tile<PT, CIC>(px[q], py[q], pz[q], g, box, offset, ihx, ihy, ihz, ext, count);
            }
        }
        __syncthreads();
        ms_block_scan(lcur, lstart, wave_tot);
        __syncthreads();
        const unsigned int total = lstart[MS_BINS - 1] + lcur[MS_BINS - 1];
        __syncthreads();
        for (int b = tid; b < MS_BINS; b += MS_BLOCK) lcur[b] = 0u;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PPT; q++) {
            const int64_t p = s0 + tid + (int64_t)q * MS_BLOCK;
            if (p < s1) {
                const Entry<PT> en{px[q], py[q], pz[q], pw[q]};
                auto place = [&](unsigned int tile) {
                    const unsigned int b = tile >> cshift, k = atomicAdd(&lcur[b], 1u), slot = lstart[b] + k;
                    if (slot < (unsigned int)MS_SUBE) {
                        buf[slot] = en, bkey[slot] = tile, bid[slot] = (unsigned short)b;
                    } else {   // more entries than the buffer holds (clouds on tile corners): straight to its place
                        const int64_t dst = base[b] + hist[b] + k;
                        stage_entry[dst] = en;
                        stage_key[dst] = tile;
                    }
                };
                if constexpr (FAST != 0) for_each_tile_fast<PT, FAST == 2>(px[q], py[q], pz[q], g, offset, ihx, ihy, ihz, ext, place);
                else for_each_tile<PT, CIC>(px[q], py[q], pz[q], g, box, offset, ihx, ihy, ihz, ext, place);
            }
        }
        __syncthreads();
        const unsigned int nbuf = min(total, (unsigned int)MS_SUBE);
        for (unsigned int i = tid; i < nbuf; i += MS_BLOCK) {
            const unsigned int b = bid[i];
            const int64_t dst = base[b] + hist[b] + (i - lstart[b]);
            stage_entry[dst] = buf[i];
            stage_key[dst] = bkey[i];
        }
        __syncthreads();
        for (int b = tid; b < MS_BINS; b += MS_BLOCK) hist[b] += lcur[b];
        __syncthreads();
    }
    }
}

// fine pass inside coarse bucket blockIdx.y, chunk blockIdx.x of its entries
template <typename PT, bool SCATTER>
__global__ __launch_bounds__(MS_BLOCK) void ms_fine(const int64_t *__restrict__ gstart, int cshift, int ntiles,
                                                    const Entry<PT> *__restrict__ stage_entry,
                                                    const unsigned int *__restrict__ stage_key,
                                                    unsigned int *__restrict__ tile_count,
                                                    const int64_t *__restrict__ tile_start,
                                                    Entry<PT> *__restrict__ entries) {
    __shared__ unsigned int hist[MS_BINS];
    __shared__ int64_t base[MS_BINS];
    const int B = blockIdx.y, tid = threadIdx.x;
    const int64_t e0 = gstart[B] + (int64_t)blockIdx.x * MS_FCHUNK, e1 = min(gstart[B + 1], e0 + MS_FCHUNK);
    if (e0 >= e1) return;
    const unsigned int tile0 = (unsigned int)B << cshift;
    const int nfine = min(1 << cshift, ntiles - (int)tile0);
    for (int f = tid; f < nfine; f += MS_BLOCK) hist[f] = 0u;
    __syncthreads();
    for (int64_t e = e0 + tid; e < e1; e += MS_BLOCK) atomicAdd(&hist[stage_key[e] - tile0], 1u);
    __syncthreads();
    if (!SCATTER) {
        for (int f = tid; f < nfine; f += MS_BLOCK)
            if (hist[f]) atomicAdd(&tile_count[tile0 + f], hist[f]);
        return;
    }
    if constexpr (SCATTER) {
        constexpr int MS_SUBE = MsSub<PT>::E;
        for (int f = tid; f < MS_BINS; f += MS_BLOCK) {
            const unsigned int c = f < nfine ? hist[f] : 0u;
            base[f] = c ? tile_start[tile0 + f] + (int64_t)atomicAdd(&tile_count[tile0 + f], c) : 0;
            hist[f] = 0u;   // from here on: entries the workgroup has already written to the tile's list
        }
        // sub-chunks sorted by tile in LDS (see MsSub): adjacent lanes write adjacent entries of a list
        __shared__ unsigned int lstart[MS_BINS], lcur[MS_BINS], wave_tot[MS_BLOCK / 64];
        __shared__ Entry<PT> buf[MS_SUBE];
        __shared__ unsigned short bid[MS_SUBE];
        constexpr int EPT = (MS_SUBE + MS_BLOCK - 1) / MS_BLOCK;
        for (int64_t s0 = e0; s0 < e1; s0 += MS_SUBE) {
            const int64_t s1 = min(s0 + MS_SUBE, e1);
            for (int f = tid; f < MS_BINS; f += MS_BLOCK) lcur[f] = 0u;
            __syncthreads();
            unsigned int fk[EPT];
#pragma unroll
            for (int q = 0; q < EPT; q++) {
                const int64_t e = s0 + tid + (int64_t)q * MS_BLOCK;
                fk[q] = 0u;
                if (e < s1) {
                    fk[q] = stage_key[e] - tile0;
                    atomicAdd(&lcur[fk[q]], 1u);
                }
            }
            __syncthreads();
            ms_block_scan(lcur, lstart, wave_tot);
            __syncthreads();
            for (int f = tid; f < MS_BINS; f += MS_BLOCK) lcur[f] = 0u;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < EPT; q++) {
                const int64_t e = s0 + tid + (int64_t)q * MS_BLOCK;
                if (e < s1) {
                    const unsigned int slot = lstart[fk[q]] + atomicAdd(&lcur[fk[q]], 1u);   // < MS_SUBE: one entry per input
                    buf[slot] = stage_entry[e];
                    bid[slot] = (unsigned short)fk[q];
                }
            }
            __syncthreads();
            for (int i = tid; i < (int)(s1 - s0); i += MS_BLOCK) {
                const unsigned int f = bid[i];
                entries[base[f] + hist[f] + (i - lstart[f])] = buf[i];
            }
            __syncthreads();
            for (int f = tid; f < MS_BINS; f += MS_BLOCK) hist[f] += lcur[f];
            __syncthreads();
        }
    }
}

// One addend of an LDS tile cell.  ACC = double / float: a floating-point LDS atomic.  ACC = unsigned long long: FIXED POINT -
// the addend v >= 0 (unweighted TSC / CIC: every cloud weight is non-negative) becomes round(v * 2^S) by one fused
// multiply-add onto 2^52 (the integer then sits in the low mantissa bits: no float -> int64 conversion sequence) and goes
// into the cell as an integer add.  Measured on MI355X (scripts/ubench/ldsatomic.hip, profiles/r04/ubench_ldsatomic.txt):
// ds_add_u64 on random cells of a tile takes 12.2 cycles per wave instruction, ds_add_f64 21.5 - and integer sums do not
// depend on the order of the adds at all, so the mesh is bit-reproducible run to run by construction.  S is picked by the
// host so that n addends of at most 1 cannot overflow 63 bits (S = 36 at 1e8 particles: a resolution of 1.5e-11 per addend).
template <typename ACC, typename V>
__device__ __forceinline__ void acc_add(ACC *cell, V v, double fxscale) {
    if constexpr (std::is_same<ACC, unsigned long long>::value) {
        const double m = __builtin_fma((double)v, fxscale, 0x1p52);
        atomicAdd(cell, (unsigned long long)__double_as_longlong(m) & 0x000fffffffffffffull);
    } else {
        atomicAdd(cell, (ACC)v);
    }
}
template <typename ACC>
__device__ __forceinline__ double acc_value(ACC a, double fxinv) {
    if constexpr (std::is_same<ACC, unsigned long long>::value) return (double)a * fxinv;
    else return (double)a;
}

// contribution of one list entry to the LDS tile with origin (ox, oy, oz) and extent (dx, dy, dz)
template <typename PT, int TYS, int TZS, bool CIC, typename ACC = double, int FAST = 0>
__device__ __forceinline__ void tile_accumulate(ACC *tile, const Entry<PT> &en, const TileGeom &g, int ox, int oy, int oz,
                                                int dx, int dy, int dz, double box, PT offset, PT ihx, PT ihy, PT ihz,
                                                double fxscale = 0.0) {
    int lx[3], ly[3], lz[3];
    if (CIC) {
        Cloud<double> c;
        cic_cloud<PT>(en.x + offset, en.y + offset, en.z + offset, box, g.gxg, g.gy, g.gz, c);
        const double W = (double)en.w;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            lx[a] = xloc(c.i[0] + a - 1, g) - ox;
            ly[a] = wrapcell(c.i[1] + a - 1, g.gy) - oy;
            lz[a] = wrapcell(c.i[2] + a - 1, g.gz) - oz;
        }
#pragma unroll
        for (int a = 0; a < 3; a++) {
            if ((unsigned)lx[a] >= (unsigned)dx) continue;
#pragma unroll
            for (int b = 0; b < 3; b++) {
                if ((unsigned)ly[b] >= (unsigned)dy) continue;
#pragma unroll
                for (int cc = 0; cc < 3; cc++) {
                    if ((unsigned)lz[cc] >= (unsigned)dz) continue;
                    const double v = c.w[0][a] * c.w[1][b] * c.w[2][cc] * W;
                    if (v != 0.0) acc_add<ACC>(&tile[(lx[a] * TYS + ly[b]) * TZS + lz[cc]], v, fxscale);
                }
            }
        }
    } else {
        Cloud<PT> c;
        tsc_cloud<PT>(en.x, en.y, en.z, offset, ihx, ihy, ihz, c);
#pragma unroll
        for (int a = 0; a < 3; a++) {
            if constexpr (FAST != 0) {   // wrapped positions: one compare-and-add each way (see for_each_tile_fast); x of a slab by its plane map
                lx[a] = (FAST == 2 ? xloc(c.i[0] + a - 1, g) : fast_wrap(c.i[0] + a - 1, g.gx)) - ox;
                ly[a] = fast_wrap(c.i[1] + a - 1, g.gy) - oy;
                lz[a] = fast_wrap(c.i[2] + a - 1, g.gz) - oz;
            } else {
                lx[a] = xloc(c.i[0] + a - 1, g) - ox;
                ly[a] = wrapcell(c.i[1] + a - 1, g.gy) - oy;
                lz[a] = wrapcell(c.i[2] + a - 1, g.gz) - oz;
            }
        }
#pragma unroll
        for (int a = 0; a < 3; a++) {
            if ((unsigned)lx[a] >= (unsigned)dx) continue;
#pragma unroll
            for (int b = 0; b < 3; b++) {
                if ((unsigned)ly[b] >= (unsigned)dy) continue;
#pragma unroll
                for (int cc = 0; cc < 3; cc++) {
                    if ((unsigned)lz[cc] >= (unsigned)dz) continue;
                    const PT v = c.w[0][a] * c.w[1][b] * c.w[2][cc] * en.w;  // wx*wy*wz*W (tsc.py:471-507)
                    acc_add<ACC>(&tile[(lx[a] * TYS + ly[b]) * TZS + lz[cc]], v, fxscale);
                }
            }
        }
    }
}

// One workgroup per tile.  LDS tile: tx*ty*tz cells of GT, strides (TYS*TZS, TZS, 1) fixed at compile time.
template <typename PT, typename GT, int TXS, int TYS, int TZS, bool CIC>
__global__ __launch_bounds__(TSC_BLOCK) void tsc_tile_deposit(const Entry<PT> *__restrict__ entries,
                                                              const int64_t *__restrict__ tile_start, TileGeom g,
                                                              double box, double offset_, GT *__restrict__ grid,
                                                              int zero_grid, GT norm, GT sub, int dbg) {
    // The tile is accumulated in float64 whatever the mesh dtype: the order of the LDS atomics then only matters at
    // the 1e-16 level, so the float32 mesh is reproducible run to run (and each cell is rounded once, not per add).
    __shared__ double tile[TXS * TYS * TZS];
    const int tid = threadIdx.x;
    const int tzi = blockIdx.x % g.ntz, tyi = (blockIdx.x / g.ntz) % g.nty, txi = blockIdx.x / (g.ntz * g.nty);
    const int ox = txi * g.tx, oy = tyi * g.ty, oz = tzi * g.tz;
    const int dx = min(g.tx, g.gx - ox), dy = min(g.ty, g.gy - oy), dz = min(g.tz, g.gz - oz);
    // the list bounds and the first entry of every thread are requested before the tile is zeroed, so their HBM
    // latency overlaps the LDS stores
    const int64_t e0 = tile_start[blockIdx.x], e1 = tile_start[blockIdx.x + 1];
    Entry<PT> first = Entry<PT>{(PT)0, (PT)0, (PT)0, (PT)0};
    if (e0 + tid < e1) first = entries[e0 + tid];
    {   // zero the tile with 16-B LDS stores
        double2 *t2 = reinterpret_cast<double2 *>(tile);
        for (int q = tid; q < TXS * TYS * TZS / 2; q += TSC_BLOCK) t2[q] = make_double2(0.0, 0.0);
    }
    __syncthreads();
    const PT ihx = (PT)(g.gxg / box), ihy = (PT)(g.gy / box), ihz = (PT)(g.gz / box);
    const PT offset = (PT)offset_;
    for (int64_t e = e0 + tid; e < e1 && !(dbg & 1); e += TSC_BLOCK) {
        const Entry<PT> en = e == e0 + tid ? first : entries[e];
        tile_accumulate<PT, TYS, TZS, CIC>(tile, en, g, ox, oy, oz, dx, dy, dz, box, offset, ihx, ihy, ihz);
    }
    __syncthreads();
    // flush: consecutive threads walk z, so a wave writes whole rows
    if (dbg & 2) return;
    if (dx == TXS && dy == TYS && dz == TZS && sizeof(GT) == 4) {
        // full tile of a float32 mesh: two cells per thread and store (8-B aligned for any even zstride), index math
        // by shifts; a wave covers four 128-B rows per store instruction
        static_assert((TZS & (TZS - 1)) == 0 && (TYS & (TYS - 1)) == 0, "tile dims must be powers of two");
        constexpr int ZP = TZS / 2;   // pairs per row
        for (int q = tid; q < TXS * TYS * ZP; q += TSC_BLOCK) {
            const int zp = q & (ZP - 1), y = (q / ZP) & (TYS - 1), x = q / (ZP * TYS);
            const double2 acc = *reinterpret_cast<const double2 *>(&tile[(x * TYS + y) * TZS + 2 * zp]);
            float2 *dst = reinterpret_cast<float2 *>(grid + ((int64_t)(ox + x) * g.gy + (oy + y)) * g.zstride + oz) + zp;
            double a0 = acc.x, a1 = acc.y;
            if (!zero_grid) {
                const float2 old = *dst;
                a0 += (double)old.x;
                a1 += (double)old.y;
            }
            float2 v = make_float2((float)a0, (float)a1);
            if (norm != (GT)0) {
                v.x = v.x * (float)norm - (float)sub;
                v.y = v.y * (float)norm - (float)sub;
            }
            *dst = v;
        }
    } else {
        const int cells = dx * dy * dz;
        for (int q = tid; q < cells; q += TSC_BLOCK) {
            const int z = q % dz, y = (q / dz) % dy, x = q / (dz * dy);
            double acc = tile[(x * TYS + y) * TZS + z];
            GT *dst = grid + ((int64_t)(ox + x) * g.gy + (oy + y)) * g.zstride + (oz + z);
            if (!zero_grid) acc += (double)*dst;
            GT v = (GT)acc;
            if (norm != (GT)0) v = v * norm - sub;
            *dst = v;
        }
    }
}

// Persistent variant for the hot case (float32 positions and mesh).  With one short-lived workgroup per tile the
// kernel is latency-bound: bounds -> entries -> LDS zero -> atomics -> flush is a chain of dependent round trips and
// only two 64-KiB tiles fit a CU.  Here a workgroup walks tiles t, t+G, ...: the list bounds of tile t+2G and the first
// entries of tile t+G are requested (untracked asynchronous loads, see fft.hip) before tile t is flushed, the flush
// re-zeroes the LDS tile as it reads it, and the stores of a flush stay in flight while the next tile is accumulated.
typedef float tsc_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void tsc_gload16_async(tsc_v4f &dst, const void *p) {
    tsc_v4f t;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(p) : "memory");
    dst = t;
}
__device__ __forceinline__ void tsc_touch(tsc_v4f &a) {
    tsc_v4f t = a;
    asm volatile("" : "+v"(t));
    a = t;
}
template <int K>
__device__ __forceinline__ void tsc_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory");
}

constexpr int TP_RANGE = 2048;   // consecutive tiles a persistent workgroup takes at a time (their list bounds sit in LDS)

template <int TXS, int TYS, int TZS, bool CIC, int NT, typename ACC, int FAST>
__global__ __launch_bounds__(NT) void tsc_tile_deposit_p(const Entry<float> *__restrict__ entries, int64_t nentries,
                                                         const int64_t *__restrict__ tile_start, int ntiles, int range_len,
                                                         TileGeom g, double box, double offset_,
                                                         float *__restrict__ grid, int zero_grid, float norm, float sub,
                                                         int dbg, double fxscale) {
    constexpr int NPRE = 2;                                // prefetched entries per thread
    constexpr int FL = TXS * TYS * (TZS / 4) / NT;         // flush stores (16 B) per thread of a full tile
    static_assert((TXS * TYS * (TZS / 4)) % NT == 0 && 2 * FL + NPRE <= 60, "whole flush stores per thread");
    __shared__ __align__(16) ACC tile[TXS * TYS * TZS];
    __shared__ unsigned int bnd[TP_RANGE + 1];             // list bounds of the range, relative to its first entry
    const int tid = threadIdx.x;
    {
        float4 *t4 = reinterpret_cast<float4 *>(tile);     // all-zero bits are 0.0 in either type
        for (int q = tid; q < (int)(sizeof(tile) / 16); q += NT) t4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float ihx = (float)(g.gxg / box), ihy = (float)(g.gy / box), ihz = (float)(g.gz / box);
    const float offset = (float)offset_;
    const double fxinv = fxscale != 0.0 ? 1.0 / fxscale : 0.0;   // a power of two: exact
    const int64_t last = nentries > 0 ? nentries - 1 : 0;
    const int nranges = (ntiles + range_len - 1) / range_len;   // range_len <= TP_RANGE
    for (int r = blockIdx.x; r < nranges; r += gridDim.x) {
        const int t0 = r * range_len, nt = min(range_len, ntiles - t0);
        const int64_t ebase = tile_start[t0];
        __syncthreads();                                   // previous range: all bnd reads done
        for (int q = tid; q <= nt; q += NT) bnd[q] = (unsigned int)(tile_start[t0 + q] - ebase);
        __syncthreads();
        tsc_v4f X[NPRE], Y[NPRE];
        auto issue = [&](tsc_v4f(&set)[NPRE], int tt) {    // first NPRE*NT entries of tile tt of this range
            const int64_t e0 = ebase + bnd[tt];
#pragma unroll
            for (int q = 0; q < NPRE; q++) tsc_gload16_async(set[q], entries + min(e0 + q * NT + tid, last));
        };
        // one tile: accumulate (from `cur`), request tile tt+2 into `cur`, flush; returns with `nxt` (tile tt+1) landed
        bool prev_plain = false;                           // the previous tile issued exactly FL stores and no load
        // tile coordinates of the tile being processed: divided out once per range, then carried (three runtime integer
        // divisions per tile in every lane were a tenth of the per-tile instruction count)
        int tzi = t0 % g.ntz, tyi = (t0 / g.ntz) % g.nty, txi = t0 / (g.ntz * g.nty);
        auto process = [&](tsc_v4f(&cur)[NPRE], int tt) {
            const int ox = txi * g.tx, oy = tyi * g.ty, oz = tzi * g.tz;
            if (++tzi == g.ntz) {       // coordinates of the NEXT tile (processed strictly in order)
                tzi = 0;
                if (++tyi == g.nty) tyi = 0, txi++;
            }
            const int dx = min(g.tx, g.gx - ox), dy = min(g.ty, g.gy - oy), dz = min(g.tz, g.gz - oz);
            const int64_t e0 = ebase + bnd[tt], e1 = ebase + bnd[tt + 1];
#pragma unroll
            for (int q = 0; q < NPRE; q++) tsc_touch(cur[q]);
#pragma unroll
            for (int q = 0; q < NPRE; q++) {
                if (e0 + q * NT + tid < e1 && !(dbg & 1)) {
                    const Entry<float> en{cur[q].x, cur[q].y, cur[q].z, cur[q].w};
                    tile_accumulate<float, TYS, TZS, CIC, ACC, FAST>(tile, en, g, ox, oy, oz, dx, dy, dz, box, offset, ihx, ihy, ihz, fxscale);
                }
            }
            bool extra = false;                            // tracked loads below: fall back to a full wait
            for (int64_t e = e0 + NPRE * NT + tid; e < e1; e += NT) {
                const Entry<float> en = entries[e];        // a copy: a reference would be re-read around every LDS atomic
                tile_accumulate<float, TYS, TZS, CIC, ACC, FAST>(tile, en, g, ox, oy, oz, dx, dy, dz, box, offset, ihx, ihy, ihz, fxscale);
            }
            extra = e1 - e0 > NPRE * NT;
            __syncthreads();
            const bool more = tt + 2 < nt;
            if (more) issue(cur, tt + 2);                  // BEFORE this tile's stores
            const bool full = dx == TXS && dy == TYS && dz == TZS;
            if (full) {
                // four cells (one 16-B store) per thread and step; a thread keeps its (y, z) and walks x, so the LDS
                // and mesh addresses advance by constants - the flush is instruction-bound at two waves per SIMD
                static_assert((TZS & (TZS - 1)) == 0 && (TYS & (TYS - 1)) == 0, "tile dims must be powers of two");
                constexpr int ZQ = TZS / 4;                       // quads per row
                constexpr int ROWS = NT / ZQ;                     // (x, y) rows covered per step
                static_assert(NT % ZQ == 0 && ROWS % TYS == 0 && (TXS * TYS) % ROWS == 0, "flush mapping");
                constexpr int XSTEP = ROWS / TYS, STEPS = TXS / XSTEP;
                const int zq = tid & (ZQ - 1), yy = (tid / ZQ) & (TYS - 1), x0 = tid / (ZQ * TYS);
                ACC *cell = &tile[(x0 * TYS + yy) * TZS + 4 * zq];
                float *dst = grid + ((int64_t)(ox + x0) * g.gy + (oy + yy)) * g.zstride + oz + 4 * zq;
                const int64_t dstep = (int64_t)XSTEP * g.gy * g.zstride;
#pragma unroll
                for (int st = 0; st < STEPS; st++, cell += XSTEP * TYS * TZS, dst += dstep) {
                    float v[4];
                    if constexpr (std::is_same<ACC, unsigned long long>::value) {
                        ulonglong2 *c2 = reinterpret_cast<ulonglong2 *>(cell);
                        const ulonglong2 a01 = c2[0], a23 = c2[1];
                        c2[0] = make_ulonglong2(0ull, 0ull);  // the tile is zero again for the next one
                        c2[1] = make_ulonglong2(0ull, 0ull);
                        double a[4] = {(double)a01.x * fxinv, (double)a01.y * fxinv, (double)a23.x * fxinv, (double)a23.y * fxinv};
                        if (!zero_grid) {
                            const float4 old = *reinterpret_cast<const float4 *>(dst);
                            a[0] += (double)old.x, a[1] += (double)old.y, a[2] += (double)old.z, a[3] += (double)old.w;
                        }
#pragma unroll
                        for (int c = 0; c < 4; c++) v[c] = (float)a[c];
                    } else if constexpr (sizeof(ACC) == 8) {
                        double2 *c2 = reinterpret_cast<double2 *>(cell);
                        const double2 a01 = c2[0], a23 = c2[1];
                        c2[0] = make_double2(0.0, 0.0);       // the tile is zero again for the next one
                        c2[1] = make_double2(0.0, 0.0);
                        double a[4] = {a01.x, a01.y, a23.x, a23.y};
                        if (!zero_grid) {
                            const float4 old = *reinterpret_cast<const float4 *>(dst);
                            a[0] += (double)old.x, a[1] += (double)old.y, a[2] += (double)old.z, a[3] += (double)old.w;
                        }
#pragma unroll
                        for (int c = 0; c < 4; c++) v[c] = (float)a[c];
                    } else {
                        float4 *c4 = reinterpret_cast<float4 *>(cell);
                        const float4 r = *c4;
                        *c4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        v[0] = r.x, v[1] = r.y, v[2] = r.z, v[3] = r.w;
                        if (!zero_grid) {
                            const float4 old = *reinterpret_cast<const float4 *>(dst);
                            v[0] += old.x, v[1] += old.y, v[2] += old.z, v[3] += old.w;
                        }
                    }
                    if (norm != 0.f) {
#pragma unroll
                        for (int c = 0; c < 4; c++) v[c] = v[c] * norm - sub;
                    }
                    if (!(dbg & 2)) *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                }
            } else {
                for (int q = tid; q < TXS * TYS * TZS; q += NT) {
                    const int z = q % TZS, y = (q / TZS) % TYS, x = q / (TZS * TYS);
                    double acc = acc_value<ACC>(tile[q], fxinv);
                    tile[q] = (ACC)0;
                    if (x < dx && y < dy && z < dz) {
                        float *dst = grid + ((int64_t)(ox + x) * g.gy + (oy + y)) * g.zstride + (oz + z);
                        if (!zero_grid) acc += (double)*dst;
                        float v = (float)acc;
                        if (norm != 0.f) v = v * norm - sub;
                        *dst = v;
                    }
                }
            }
            __syncthreads();
            // vector-memory operations issued since the request for tile tt+1: [FL stores of tile tt-1] [NPRE loads of
            // tile tt+2] [FL stores of this tile] - when exactly so, vmcnt(2 FL + NPRE) means tile tt+1 has landed and
            // everything younger may stay in flight; otherwise wait for everything
            const bool plain = full && zero_grid && !extra && !(dbg & 2);   // (no stores issued: the count below would not hold)
            if (plain && prev_plain && more) tsc_wait_vmcnt<2 * FL + NPRE>();
            else tsc_wait_vmcnt<0>();
            prev_plain = plain;
        };
        issue(X, 0);
        if (nt > 1) issue(Y, 1);
        tsc_wait_vmcnt<0>();
        for (int tt = 0; tt < nt; tt += 2) {
            process(X, tt);
            if (tt + 1 < nt) process(Y, tt + 1);
        }
    }
}

// did any deposit since the last reset move a position into the box?  (the host entry points copy the positions back to the
// caller - the reference wraps its argument in place, tsc.py:171-173 - only then: 1.2 GB over PCIe at 1e8 particles)
static int g_wrapped_seen = 0;

#include "tsc_lines.hpp"
#include "tsc_lines3.hpp"

// ---- host-side driver ------------------------------------------------------------------------------------
// line lists (tsc_lines.hpp, tsc_lines3.hpp): work buffers and the staged block records kept for the second deposit of an
// interlaced pair (the fine level is rebuilt per origin from the same records)
struct LinesWork {
    DevBuf M, tot, tables, staged, C, tile_start, tile_cnt, entries, flag;
};
LinesWork g_lw;

// block shape: tiles per block and blocks within the limits of a configuration (0: 512 tiles x 256 blocks, both scatter
// kernels at two workgroups per CU; 1: 1024 x 1024), smallest surface-to-volume ratio of a block
static bool lines_geometry(int gx, int gy, int gz, int64_t zstride, LGeom &g, int &cfg) {
    if (gx % LN_TX || gy % LN_TY || gz % LN_TZ || gx / LN_TX > 256 || gy / LN_TY > 256 || gz / LN_TZ > 256) return false;
    g.n[0] = gx, g.n[1] = gy, g.n[2] = gz;
    g.nt[0] = gx / LN_TX, g.nt[1] = gy / LN_TY, g.nt[2] = gz / LN_TZ;
    g.zstride = zstride;
    const int T[3] = {LN_TX, LN_TY, LN_TZ};
    for (cfg = option("tsc_lines_cfg1") ? 1 : 0; cfg < 2; cfg++) {
        const int limT = cfg ? 1024 : 512, limB = cfg ? 1024 : 256;
        double best = 1e30;
        bool found = false;
        for (int a = 0; a <= 5; a++)
            for (int b = 0; b <= 5; b++)
                for (int c = 0; c <= 5; c++) {
                    const int sb[3] = {a, b, c};
                    bool ok = true;
                    int64_t nb = 1;
                    double surf = 0;
                    for (int d = 0; d < 3; d++) {
                        if (g.nt[d] % (1 << sb[d])) ok = false;
                        nb *= g.nt[d] >> sb[d];
                        surf += 1.0 / ((double)T[d] * (1 << sb[d]));
                    }
                    if (!ok || (1 << (a + b + c)) > limT || nb > limB) continue;
                    // every block at least two tiles... (a cloud touches at most two blocks per dimension only if a block
                    // is at least five cells wide: always, a tile is 16)
                    const double cost = surf + 1e-6 * (double)nb;
                    if (cost < best) {
                        best = cost, found = true;
                        for (int d = 0; d < 3; d++) g.sb[d] = sb[d], g.nb[d] = g.nt[d] >> sb[d];
                        g.tpb = 1 << (a + b + c);
                        g.nbuckets = (int)nb;
                    }
                }
        if (found) return true;
    }
    return false;
}

static int lines_fxscale(int64_t n, double &fxscale) {
    int fxs = 40;
    while (fxs > 8 && (double)std::max<int64_t>(n, 1) * std::ldexp(1.0, fxs) >= 0x1p62) fxs--;
    fxscale = std::ldexp(1.0, fxs);
    return fxs;
}

// tile deposit from packed entries: 32-bit fixed-point tile sums (four workgroups per CU) or, `tsc_acc64`, 64-bit sums
static int lines_launch_deposit(const unsigned long long *entries, int64_t fs, int64_t nentries, int64_t n, const unsigned int *tile_start,
                                const unsigned int *tile_cnt, const LGeom &g, float *grid, int zero_grid, double norm, double sub) {
    const int ntiles = g.nbuckets * g.tpb;
    int dev = 0, ncu = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    const int range_len = (int)std::min<int64_t>(LD_RANGE, std::max<int64_t>(64, ceil_div(ntiles, (int64_t)ncu * 8)));
    const int nranges = (int)ceil_div(ntiles, range_len);
    const int grid_p = (int)std::min<int64_t>(nranges, (int64_t)ncu * 2);
    const int dbg = option("dbg_tsc") & 3;
    if (!option("tsc_acc64")) {   // 32-bit tile sums, four workgroups per CU
        const int range32 = (int)std::min<int64_t>(LD32_RANGE, std::max<int64_t>(64, ceil_div(ntiles, (int64_t)ncu * 16)));
        const int grid32 = (int)std::min<int64_t>(ceil_div(ntiles, range32), (int64_t)ncu * 4);
        ABACUS_LAUNCH("tsc_tile_deposit", (lines_deposit32<256>), dim3(grid32), dim3(256), 0, entries, fs, tile_start, tile_cnt, ntiles, range32, g,
                      grid, zero_grid, (float)norm, (float)sub, dbg);
        return 0;
    }
    const bool dense = nentries / std::max(ntiles, 1) > 400;
    double fxscale;
    lines_fxscale(n, fxscale);
    if (dense)
        ABACUS_LAUNCH("tsc_tile_deposit", (lines_deposit<512>), dim3(grid_p), dim3(512), 0, entries, fs, tile_start, tile_cnt, ntiles, range_len, g,
                      grid, zero_grid, (float)norm, (float)sub, fxscale, dbg);
    else
        ABACUS_LAUNCH("tsc_tile_deposit", (lines_deposit<256>), dim3(grid_p), dim3(256), 0, entries, fs, tile_start, tile_cnt, ntiles, range_len, g,
                      grid, zero_grid, (float)norm, (float)sub, fxscale, dbg);
    return 0;
}

// ---- third generation (tsc_lines3.hpp): block records; one build serves both deposits of an interlaced pair ----------
// diagnostic phase clocks of the split rounds (option tsc_lines_clk; abacus_tsc_lines_clocks reads and clears them)
DevBuf g_lines_clk;
static unsigned long long *lines_clk(int off) {
    if (!option("tsc_lines_clk")) return nullptr;
    if (!g_lines_clk.p) {
        if (g_lines_clk.reserve(32 * sizeof(unsigned long long)) != 0) return nullptr;
        (void)hipMemsetAsync(g_lines_clk.p, 0, 32 * sizeof(unsigned long long), stream());
    }
    return g_lines_clk.as<unsigned long long>() + off;
}

struct Lines3State {
    bool valid = false;
    const void *pos = nullptr;
    int64_t n = 0, zstride = 0, fs = 0;
    int gx = 0, gy = 0, gz = 0, cfg = 0, np = 0, ext = 0;
    double box = 0, offset = 0;
    LGeom g;
    size_t o_f = 0, o_pf = 0, o_p = 0;
};
Lines3State g_l3;

// Deferred mode (tsc_lines_defer(1), set by the P(k) pipeline around its deposits): a build whose mesh matches the last EXACT
// build sizes its buffers from that build's records / entries per particle and makes its tables on the device
// (lines3_tables) - no stream synchronise between the counting and the scattering pass.  What the build needed comes back
// asynchronously into page-locked memory; the pipeline reads it after its own final synchronise
// (tsc_lines_deferred_check) and, should the buffers have been too small, runs again with deferred mode off.
struct L3Caps {
    bool valid = false;
    int gx = 0, gy = 0, gz = 0, cfg = 0, ext = 0;
    double rec_pp = 0, ent_pp = 0;
};
L3Caps g_l3_caps;
bool g_l3_defer = false;
unsigned int *g_l3_pend = nullptr;   // page-locked: 32 slots of 8 words
int g_l3_npend = 0;

void lines_defer_set(int on) { g_l3_defer = on != 0; }
// after a stream synchronise: 1 if a deferred build of this pipeline did not fit its buffers (its mesh is garbage)
int lines_deferred_check() {
    int over = 0;
    for (int i = 0; i < g_l3_npend; i++) {
        g_wrapped_seen |= (int)g_l3_pend[8 * i + 4];
        over |= g_l3_pend[8 * i + 3] ? 1 : 0;
    }
    g_l3_npend = 0;
    if (over) g_l3_caps.valid = false, g_l3.valid = false;
    return over;
}

// count + coarse: the staged block records of `pos` at mesh offset `offset` (ext: blocks of the 4-cell union of the clouds at
// `offset` and `offset` + half a cell).  Returns 1 when 32-bit indices do not hold the lists (caller falls back).
static int lines3_build(float *pos, int64_t n, const LGeom &g, int cfg, double box, double offset, int wrap, int ext, int *wrapped_out,
                        const L3Win &wn = L3Win{0, 0, 0, 0, -1}) {
    g_l3.valid = false;
    const int nb = g.nbuckets;
    const int64_t CH = std::max<int64_t>(8192, ceil_div(n, 1024));
    const int nchunk = (int)ceil_div(n, CH);
    ABACUS_TRY(g_lw.M.reserve((size_t)nchunk * nb * sizeof(unsigned int)));
    // [records per block (nb)] [wrapped flag] [tile entries per block (nb)]: one read-back into page-locked memory
    const size_t nword = (size_t)2 * nb + 1;
    ABACUS_TRY(g_lw.tot.reserve(nword * sizeof(unsigned int)));
    unsigned int *M = g_lw.M.as<unsigned int>(), *tot = g_lw.tot.as<unsigned int>();
    int *flag = reinterpret_cast<int *>(tot + nb);
    unsigned int *ent = tot + nb + 1;
    HIP_TRY(hipMemsetAsync(flag, 0, (size_t)(nb + 1) * sizeof(unsigned int), stream()));
    const float offA = (float)offset;
    const size_t count_lds = (size_t)3 * (std::max({wn.on ? wn.n : g.n[0], g.n[1], g.n[2]}) + 5) * sizeof(unsigned int);
#define L3_COUNT(NBK, EXT_) ABACUS_LAUNCH("tsc_lines_count", (lines3_count<NBK, EXT_>), dim3(nchunk), dim3(512), count_lds, pos, n, g, box, offA, wrap, CH, M, ent, flag, wn)
    if (cfg == 0) {
        if (ext) L3_COUNT(256, true);
        else L3_COUNT(256, false);
    } else {
        if (ext) L3_COUNT(1024, true);
        else L3_COUNT(1024, false);
    }
#undef L3_COUNT
    ABACUS_LAUNCH("tsc_lines_colscan", lines_colscan, dim3(nb), dim3(1024), 0, M, nchunk, nb, tot);
    const int64_t PIECE = 64 * (int64_t)g.tpb;   // measured at BASELINE config 3 (r04): 32K 10.36 ms, 64K 10.40, 128K 10.52, 256K 10.58
#define L3_COARSE(NBK, LINE_, SBUF_, NT_, EXT_, GSTART, NEED)                                                                        \
    ABACUS_LAUNCH("tsc_lines_coarse", (lines3_coarse<NBK, LINE_, SBUF_, NT_, EXT_>), dim3(nchunk), dim3(NT_), 0, (const float *)pos, n, g, box, \
                  offA, CH, (const unsigned int *)M, GSTART, g_lw.staged.as<uint4>(), lines_clk(0), NEED, wn)
#define L3_COARSE_ALL(GSTART, NEED)                                  \
    do {                                                             \
        if (cfg == 0) {                                              \
            if (ext) L3_COARSE(256, 8, 2560, 512, true, GSTART, NEED);     \
            else L3_COARSE(256, 8, 2560, 512, false, GSTART, NEED);        \
        } else {                                                     \
            if (ext) L3_COARSE(1024, 4, 3328, 1024, true, GSTART, NEED);   \
            else L3_COARSE(1024, 4, 3328, 1024, false, GSTART, NEED);      \
        }                                                            \
    } while (0)
    auto remember = [&](int np_, int64_t fs_, size_t o_f_, size_t o_pf_, size_t o_p_) {
        g_l3.pos = pos, g_l3.n = n, g_l3.fs = fs_, g_l3.cfg = cfg, g_l3.np = np_, g_l3.ext = ext;
        g_l3.box = box, g_l3.offset = offset, g_l3.g = g;
        g_l3.gx = g.n[0], g_l3.gy = g.n[1], g_l3.gz = g.n[2], g_l3.zstride = g.zstride;
        g_l3.o_f = o_f_, g_l3.o_pf = o_pf_, g_l3.o_p = o_p_;
        g_l3.valid = true;
    };
    // ---- deferred: sizes from the last exact build of this mesh, tables on the device, no synchronise
    const L3Caps &cp = g_l3_caps;
    if (g_l3_defer && !wn.on && cp.valid && cp.gx == g.n[0] && cp.gy == g.n[1] && cp.gz == g.n[2] && cp.cfg == cfg && cp.ext == ext && g_l3_npend < 32 &&
        option("tsc_lines_sync") != 1 && nb <= 1024) {
        // (diagnostic: tsc_lines_sync = 2 sizes the record buffer for a twentieth of the particles - the overflow path of the tests)
        const double shrink = option("tsc_lines_sync") == 2 ? 0.05 : 1.03;
        const int64_t gs_cap = ((int64_t)((double)n * cp.rec_pp * shrink) + 16 * (int64_t)nb + 65536) & ~(int64_t)15;
        const int64_t fs_cap = ((int64_t)((double)n * cp.ent_pp * 1.03) + (15 * (int64_t)g.tpb + 16) * nb + 65536) & ~(int64_t)15;
        const int64_t np_cap64 = gs_cap / PIECE + nb + 1;
        if (gs_cap < 0xfff00000ll && fs_cap < 0xfff00000ll && np_cap64 < (1 << 24)) {
            const int np_cap = (int)np_cap64;
            const size_t o_g = 0, o_f = o_g + (size_t)(nb + 1) * 4, o_pf = o_f + (size_t)(nb + 1) * 4, o_p = (o_pf + (size_t)(nb + 1) * 4 + 15) & ~(size_t)15,
                         o_n = o_p + (size_t)np_cap * sizeof(LnPiece), tbytes = o_n + 32;
            ABACUS_TRY(g_lw.tables.reserve(tbytes));
            ABACUS_TRY(g_lw.staged.reserve((size_t)gs_cap * sizeof(uint4)));
            if (!g_l3_pend) HIP_TRY(hipHostMalloc((void **)&g_l3_pend, 32 * 8 * sizeof(unsigned int), hipHostMallocDefault));
            char *tb = g_lw.tables.as<char>();
            unsigned int *d_need = reinterpret_cast<unsigned int *>(tb + o_n);
            ABACUS_LAUNCH("tsc_lines_tables", lines3_tables, dim3(1), dim3(1024), 0, (const unsigned int *)tot, (const unsigned int *)ent, (const int *)flag, nb, g.tpb,
                          (unsigned int)PIECE, (unsigned long long)gs_cap, (unsigned long long)fs_cap, np_cap, reinterpret_cast<unsigned int *>(tb + o_g),
                          reinterpret_cast<unsigned int *>(tb + o_f), reinterpret_cast<int *>(tb + o_pf), reinterpret_cast<LnPiece *>(tb + o_p), d_need);
            HIP_TRY(hipMemcpyAsync(g_l3_pend + 8 * g_l3_npend, d_need, 5 * sizeof(unsigned int), hipMemcpyDeviceToHost, stream()));
            g_l3_npend++;
            L3_COARSE_ALL(reinterpret_cast<const unsigned int *>(tb + o_g), (const unsigned int *)d_need);
            if (wrapped_out) *wrapped_out = 0;   // (known only after the pipeline's synchronise: tsc_lines_deferred_check)
            remember(np_cap, fs_cap, o_f, o_pf, o_p);
            return 0;
        }
    }
    static unsigned int *h_tot = nullptr;
    static size_t h_tot_cap = 0;
    if (nword > h_tot_cap) {
        if (h_tot) HIP_TRY(hipHostFree(h_tot));
        h_tot = nullptr, h_tot_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&h_tot, nword * sizeof(unsigned int) * 2, hipHostMallocDefault));
        h_tot_cap = 2 * nword;
    }
    HIP_TRY(hipMemcpyAsync(h_tot, tot, nword * sizeof(unsigned int), hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    const int h_flag = (int)h_tot[nb];
    const unsigned int *h_ent = h_tot + nb + 1;
    if (wrapped_out) *wrapped_out = h_flag;
    g_wrapped_seen |= h_flag;
    std::vector<unsigned int> gstart((size_t)nb + 1), fstart((size_t)nb + 1);
    std::vector<int> piece_first((size_t)nb + 1);
    std::vector<LnPiece> pieces;
    int64_t gs = 0, fs = 0;
    for (int b = 0; b < nb; b++) {
        gstart[b] = (unsigned int)gs, fstart[b] = (unsigned int)fs;
        piece_first[b] = (int)pieces.size();
        for (int64_t e = 0; e < (int64_t)h_tot[b]; e += PIECE)
            pieces.push_back(LnPiece{b, (unsigned int)(gs + e), (unsigned int)(gs + std::min<int64_t>(e + PIECE, h_tot[b])), e == 0});
        gs += ((int64_t)h_tot[b] + 15) & ~(int64_t)15;
        fs += ((int64_t)h_ent[b] + 15 * (int64_t)g.tpb + 15) & ~(int64_t)15;   // every tile list starts on a line boundary
        if (gs >= 0xfff00000ll || fs >= 0xfff00000ll) return 1;
    }
    // what a deferred build of this mesh may assume (records and tile entries per particle, padding included)
    g_l3_caps.valid = n > 0 && !wn.on, g_l3_caps.gx = g.n[0], g_l3_caps.gy = g.n[1], g_l3_caps.gz = g.n[2], g_l3_caps.cfg = cfg, g_l3_caps.ext = ext;
    g_l3_caps.rec_pp = (double)gs / (double)std::max<int64_t>(n, 1), g_l3_caps.ent_pp = (double)fs / (double)std::max<int64_t>(n, 1);
    gstart[nb] = (unsigned int)gs, fstart[nb] = (unsigned int)fs;
    piece_first[nb] = (int)pieces.size();
    const int np = (int)pieces.size();
    const size_t o_g = 0, o_f = o_g + (size_t)(nb + 1) * 4, o_pf = o_f + (size_t)(nb + 1) * 4, o_p = (o_pf + (size_t)(nb + 1) * 4 + 15) & ~(size_t)15,
                 tbytes = o_p + std::max<size_t>(pieces.size(), 1) * sizeof(LnPiece);
    static char *h_blob = nullptr;
    static size_t h_blob_cap = 0;
    if (tbytes > h_blob_cap) {
        if (h_blob) HIP_TRY(hipHostFree(h_blob));
        h_blob = nullptr, h_blob_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&h_blob, tbytes * 2, hipHostMallocDefault));
        h_blob_cap = tbytes * 2;
    }
    memcpy(h_blob + o_g, gstart.data(), (size_t)(nb + 1) * 4);
    memcpy(h_blob + o_f, fstart.data(), (size_t)(nb + 1) * 4);
    memcpy(h_blob + o_pf, piece_first.data(), (size_t)(nb + 1) * 4);
    if (np) memcpy(h_blob + o_p, pieces.data(), pieces.size() * sizeof(LnPiece));
    ABACUS_TRY(g_lw.tables.reserve(tbytes));
    ABACUS_TRY(g_lw.staged.reserve((size_t)std::max<int64_t>(gs, 16) * sizeof(uint4)));
    HIP_TRY(hipMemcpyAsync(g_lw.tables.p, h_blob, tbytes, hipMemcpyHostToDevice, stream()));
    const unsigned int *d_gstart = reinterpret_cast<const unsigned int *>(g_lw.tables.as<char>() + o_g);
    L3_COARSE_ALL(d_gstart, (const unsigned int *)nullptr);
#undef L3_COARSE_ALL
#undef L3_COARSE
    remember(np, fs, o_f, o_pf, o_p);
    return 0;
}

// fcount + fscan + fine + deposit from the staged records; org 1: the mesh origin half a cell further (needs an `ext` build)
static int lines3_deposit(float *grid, int org, int zero_grid, double norm, double sub) {
    const LGeom &g = g_l3.g;
    const int nb = g.nbuckets, ntiles = nb * g.tpb, np = g_l3.np, cfg = g_l3.cfg;
    const int64_t fs = g_l3.fs;
    ABACUS_TRY(g_lw.C.reserve((size_t)std::max(np, 1) * g.tpb * sizeof(unsigned int)));
    ABACUS_TRY(g_lw.tile_start.reserve((size_t)ntiles * sizeof(unsigned int)));
    ABACUS_TRY(g_lw.tile_cnt.reserve((size_t)ntiles * sizeof(unsigned int)));
    ABACUS_TRY(g_lw.entries.reserve((size_t)std::max<int64_t>(fs, 16) * sizeof(unsigned long long)));
    const char *tb = g_lw.tables.as<char>();
    const unsigned int *d_fstart = reinterpret_cast<const unsigned int *>(tb + g_l3.o_f);
    const int *d_pfirst = reinterpret_cast<const int *>(tb + g_l3.o_pf);
    const LnPiece *d_pieces = reinterpret_cast<const LnPiece *>(tb + g_l3.o_p);
    const uint4 *staged = g_lw.staged.as<uint4>();
    unsigned int *C = g_lw.C.as<unsigned int>(), *tile_start = g_lw.tile_start.as<unsigned int>(), *tile_cnt = g_lw.tile_cnt.as<unsigned int>();
    unsigned long long *entries = g_lw.entries.as<unsigned long long>();
#define L3_FINE(NBF, SBUF_, NT_, ORG_)                                                                                                 \
    do {                                                                                                                               \
        if (np) ABACUS_LAUNCH("tsc_lines_fcount", (lines3_fcount<NBF, ORG_>), dim3(np), dim3(512), 0, staged, d_pieces, g, C);          \
        ABACUS_LAUNCH("tsc_lines_fscan", (lines_fscan<NBF>), dim3(nb), dim3(NBF), 0, C, d_pfirst, g, d_fstart, tile_start, tile_cnt);   \
        if (np) ABACUS_LAUNCH("tsc_lines_fine", (lines3_fine<NBF, 8, SBUF_, NT_, ORG_>), dim3(np), dim3(NT_), 0, staged, d_pieces, g,  \
                              (const unsigned int *)C, entries, lines_clk(16));                                                        \
    } while (0)
    if (cfg == 0) {
        if (org) L3_FINE(512, 2560, 512, 1);
        else L3_FINE(512, 2560, 512, 0);
    } else {
        if (org) L3_FINE(1024, 4096, 1024, 1);
        else L3_FINE(1024, 4096, 1024, 0);
    }
#undef L3_FINE
    return lines_launch_deposit(entries, fs, fs, g_l3.n, tile_start, tile_cnt, g, grid, zero_grid, norm, sub);
}

struct TscWork {
    DevBuf tile_count, tile_start, entries, flag, scan, gcount, gstart, stage_entry, stage_key;
};
TscWork g_work;

// Lists kept from a `list_mode = 1` build (extended clouds): a following `list_mode = 2` deposit of the same particles on
// the same mesh (the half-cell-shifted deposit of an interlaced pair) reuses them instead of sorting again.
struct ListCache {
    bool valid = false;
    const void *pos = nullptr;
    int64_t n = 0, zstride = 0, nentries = 0;
    int gx = 0, gy = 0, gz = 0, gxg = 0, xoff = 0, xoff2 = -1, cic = 0;
    double box = 0;
    void *entries = nullptr;
};
ListCache g_lists;

TileGeom make_geom(int gx, int gy, int gz, int64_t zstride, int TX, int TY, int TZ, int gxg, int xoff, int xoff2) {
    TileGeom g;
    g.gx = gx, g.gy = gy, g.gz = gz;
    g.gxg = gxg, g.xoff = xoff;
    g.xoff2 = xoff2, g.xwin = xoff2 < 0 ? gx : gx / 2;
    g.tx = std::min(TX, gx), g.ty = std::min(TY, gy), g.tz = std::min(TZ, gz);
    g.ntx = (gx + g.tx - 1) / g.tx, g.nty = (gy + g.ty - 1) / g.ty, g.ntz = (gz + g.tz - 1) / g.tz;
    auto lg = [](int v) {
        int sh = 0;
        while ((1 << sh) < v) sh++;
        return (1 << sh) == v ? sh : -1;
    };
    g.shx = lg(g.tx), g.shy = lg(g.ty), g.shz = lg(g.tz);
    auto four = [](int cells, int t) { return t >= 4 && (cells % t == 0 || cells % t >= 4) ? 1 : 0; };
    g.f4x = (gxg == gx && xoff == 0) ? four(gx, g.tx) : 0, g.f4y = four(gy, g.ty), g.f4z = four(gz, g.tz);
    g.zstride = zstride;
    return g;
}


template <typename PT, typename GT, bool CIC>
int deposit_dev(PT *pos, int64_t n, const PT *weights, GT *grid, int gx, int gy, int gz, int64_t zstride, double box,
                double offset, int wrap, int zero_grid, double norm, int *wrapped_out, int gxg = -1, int xoff = 0,
                double sub = 1.0, int list_mode = 0, int xoff2 = -1, int nx_alloc = 0) {
    if (gxg < 0) gxg = gx;
    if (xoff2 >= 0 && (gx % 2 || gx / 2 > gxg)) return fail("tsc: two windows of %d planes in a mesh of %d", gx, gxg);
    if (gx < 1 || gy < 1 || gz < 1) return fail("tsc: empty mesh");
    if (gxg > 32767 || gy > 32767 || gz > 32767) return fail("tsc: mesh dimension > 32767 (int16 cell index of the reference)");
    constexpr int TX = 16, TY = 16, TZ = 32;   // 8192 float64 cells = 64 KiB of LDS -> two workgroups per CU
    const TileGeom g = make_geom(gx, gy, gz, zstride, TX, TY, TZ, gxg, xoff, xoff2);
    const int64_t ntiles64 = (int64_t)g.ntx * g.nty * g.ntz;
    if (ntiles64 > 0x7fffffff) return fail("tsc: too many tiles");
    const int ntiles = (int)ntiles64;
    ABACUS_TRY(g_work.tile_count.reserve((size_t)(ntiles + 1) * sizeof(unsigned int)));
    ABACUS_TRY(g_work.tile_start.reserve((size_t)(ntiles + 1) * sizeof(int64_t)));
    ABACUS_TRY(g_work.flag.reserve(256));
    unsigned int *tile_count = g_work.tile_count.as<unsigned int>();
    int64_t *tile_start = g_work.tile_start.as<int64_t>();
    int *flag = g_work.flag.as<int>();
    HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int), stream()));
    Entry<PT> *entries = nullptr;
    int cshift = 0;
    while (((int64_t)ntiles + (1 << cshift) - 1) >> cshift > MS_BINS) cshift++;
    const bool multisplit = n >= 2000000 && cshift <= 10 && !option("tsc_atomic");
    int64_t nentries_total = 0;
    // list sharing (multisplit path only): mode 1 builds lists that also cover a deposit shifted by up to half a cell,
    // mode 2 reuses them when nothing about the particles or the mesh changed
    const bool share = multisplit && list_mode != 0 && !option("tsc_noshare");
    if (share && list_mode == 1 && offset != 0.0) return fail("tsc: shared lists are built at offset 0");
    if (share && list_mode == 2 && (offset < 0.0 || offset > 0.5 * box / gxg * 1.0000001))
        return fail("tsc: shared lists cover offsets up to half a cell");
    // line lists (tsc_lines3.hpp + tsc_lines.hpp): unweighted float32 TSC on a full periodic mesh of whole tiles
    int lines_wrapped = 0;   // its counting pass wrapped positions in place before it handed over to the first generation
    if constexpr (std::is_same<PT, float>::value && std::is_same<GT, float>::value && !CIC) {
        LGeom lg;
        int lcfg = 0;
        const double cell = box / gx;
        // (|offset| up to two cells of the FINEST dimension: the lists take nearest cells -2 .. n + 2)
        const double mincell = box / std::max(gx, std::max(gy, gz));
        if (multisplit && !weights && wrap && gxg == gx && xoff == 0 && xoff2 < 0 && ntiles >= 4096 && !option("tsc_oldlists") &&
            std::fabs(offset) <= 2.0 * mincell && lines_geometry(gx, gy, gz, zstride, lg, lcfg)) {
            g_lists.valid = false;
            // block records (tsc_lines3.hpp).  list_mode 1: built at offset 0 for both deposits of an interlaced pair;
            // list_mode 2: the half-cell-shifted deposit from the records of that build, if nothing changed since
            const bool want_share = list_mode != 0 && !option("tsc_noshare");
            const bool reuse3 = want_share && list_mode == 2 && g_l3.valid && g_l3.ext && g_l3.pos == (const void *)pos && g_l3.n == n &&
                                g_l3.zstride == zstride && g_l3.gx == gx && g_l3.gy == gy && g_l3.gz == gz && g_l3.box == box &&
                                g_l3.offset == 0.0 && std::fabs(offset - 0.5 * cell) <= 1e-7 * cell;
            if (reuse3) {
                if (wrapped_out) *wrapped_out = 0;
                return lines3_deposit(grid, 1, zero_grid, norm, sub);
            }
            const int rc = lines3_build(pos, n, lg, lcfg, box, offset, wrap, want_share && list_mode == 1 && offset == 0.0 ? 1 : 0, wrapped_out);
            if (rc == 0) return lines3_deposit(grid, 0, zero_grid, norm, sub);
            if (rc <= 0) return rc;   // 1: more than 2^32 entries - the first-generation lists below
            if (wrapped_out) lines_wrapped = *wrapped_out;
        }
    }
    // the same lists for the windows of a slab-decomposed mesh (one or two x-windows stored back to back in a buffer the caller
    // padded to `nx_alloc` planes, a whole number of tiles): tsc_lines3.hpp, L3Win
    if constexpr (std::is_same<PT, float>::value && std::is_same<GT, float>::value && !CIC) {
        const int win = xoff2 < 0 ? gx : gx / 2;
        const int dsep = xoff2 < 0 ? gxg : std::min((xoff2 - xoff + gxg) % gxg, (xoff - xoff2 + gxg) % gxg);
        LGeom lg;
        int lcfg = 0;
        if (multisplit && !weights && wrap && gxg != gx && nx_alloc >= gx && nx_alloc % LN_TX == 0 && nx_alloc < gx + LN_TX && win <= dsep &&
            win >= 8 && zero_grid && !option("tsc_oldlists") && std::fabs(offset) <= 2.0 * box / std::max(gxg, std::max(gy, gz)) &&
            (int64_t)(nx_alloc / LN_TX) * (gy / LN_TY) * (gz / LN_TZ) >= 4096 && lines_geometry(nx_alloc, gy, gz, zstride, lg, lcfg)) {
            g_lists.valid = false;
            const L3Win wn{1, gxg, win, xoff, xoff2};
            const int rc = lines3_build(pos, n, lg, lcfg, box, offset, wrap, 0, wrapped_out, wn);
            if (rc == 0) return lines3_deposit(grid, 0, zero_grid, norm, sub);
            if (rc < 0) return rc;
            if (wrapped_out) lines_wrapped |= *wrapped_out;
        }
    }
    const int ext = share ? 1 : 0;
    const bool reuse = share && list_mode == 2 && g_lists.valid && g_lists.pos == (const void *)pos && g_lists.n == n &&
                       g_lists.zstride == zstride && g_lists.gx == gx && g_lists.gy == gy && g_lists.gz == gz &&
                       g_lists.gxg == gxg && g_lists.xoff == xoff && g_lists.xoff2 == xoff2 && g_lists.cic == (CIC ? 1 : 0) && g_lists.box == box &&
                       sizeof(PT) == 4;
    if (!reuse) g_lists.valid = false;   // every build below overwrites the buffers the cache points into
    if (reuse) {
        entries = static_cast<Entry<PT> *>(g_lists.entries);
        nentries_total = g_lists.nentries;
        if (wrapped_out) *wrapped_out = lines_wrapped;
    } else if (multisplit) {
        const int ncoarse = (int)(((int64_t)ntiles + (1 << cshift) - 1) >> cshift);
        ABACUS_TRY(g_work.gcount.reserve((size_t)(MS_BINS + 1) * sizeof(unsigned int)));
        ABACUS_TRY(g_work.gstart.reserve((size_t)(MS_BINS + 1) * sizeof(int64_t)));
        unsigned int *gcount = g_work.gcount.as<unsigned int>();
        int64_t *gstart = g_work.gstart.as<int64_t>();
        HIP_TRY(hipMemsetAsync(gcount, 0, (size_t)(MS_BINS + 1) * sizeof(unsigned int), stream()));
        const int cgrid = (int)ceil_div(n, MS_CHUNK);
        // FAST: full mesh, power-of-two tiles obeying the four-cells-two-tiles rule, TSC (see for_each_tile_fast)
        const bool fastyz = !CIC && wrap && g.shx >= 0 && g.shy >= 0 && g.shz >= 0 && g.f4y && g.f4z && g.gy >= 8 && g.gz >= 8;
        const bool fast = fastyz && g.gxg == g.gx && g.xoff == 0 && g.xoff2 < 0 && g.f4x && g.gx >= 8;
        const bool fast_slab = fastyz && !fast && !option("tsc_noslabfast");   // an x-slab: y and z as on the full mesh
#define MS_COARSE(SC, name, ...)                                                                                                  \
    do {                                                                                                                          \
        if (fast) ABACUS_LAUNCH(name, (ms_coarse<PT, false, SC, 1>), dim3(cgrid), dim3(MS_BLOCK), 0, __VA_ARGS__);                \
        else if (fast_slab) ABACUS_LAUNCH(name, (ms_coarse<PT, false, SC, 2>), dim3(cgrid), dim3(MS_BLOCK), 0, __VA_ARGS__);      \
        else ABACUS_LAUNCH(name, (ms_coarse<PT, CIC, SC, 0>), dim3(cgrid), dim3(MS_BLOCK), 0, __VA_ARGS__);                       \
    } while (0)
        MS_COARSE(false, "tsc_ms_coarse_count", pos, n, weights, g, box, offset, wrap, cshift, ncoarse, gcount,
                  (const int64_t *)nullptr, (Entry<PT> *)nullptr, (unsigned int *)nullptr, flag, ext);
        ABACUS_TRY(exclusive_scan_u32(gcount, ncoarse, gstart, g_work.scan, 1));   // counters re-zeroed: cursors
        std::vector<int64_t> h_start((size_t)ncoarse + 1);
        int h_flag = 0;
        HIP_TRY(hipMemcpyAsync(h_start.data(), gstart, (size_t)(ncoarse + 1) * sizeof(int64_t), hipMemcpyDeviceToHost,
                               stream()));
        HIP_TRY(hipMemcpyAsync(&h_flag, flag, sizeof(int), hipMemcpyDeviceToHost, stream()));
        HIP_TRY(hipStreamSynchronize(stream()));
        if (wrapped_out) *wrapped_out = h_flag | lines_wrapped;
        g_wrapped_seen |= h_flag;
        const int64_t total = h_start[ncoarse];
        nentries_total = total;
        int64_t maxbucket = 0;
        for (int b = 0; b < ncoarse; b++) maxbucket = std::max(maxbucket, h_start[b + 1] - h_start[b]);
        const size_t t1 = (size_t)std::max<int64_t>(total, 1);
        ABACUS_TRY(g_work.stage_entry.reserve(t1 * sizeof(Entry<PT>)));
        ABACUS_TRY(g_work.stage_key.reserve(t1 * sizeof(unsigned int)));
        Entry<PT> *stage_entry = g_work.stage_entry.as<Entry<PT>>();
        unsigned int *stage_key = g_work.stage_key.as<unsigned int>();
        MS_COARSE(true, "tsc_ms_coarse_scatter", pos, n, weights, g, box, offset, 0, cshift, ncoarse, gcount,
                  (const int64_t *)gstart, stage_entry, stage_key, flag, ext);
#undef MS_COARSE
        if (cshift == 0) {   // every bucket is a tile already
            HIP_TRY(hipMemcpyAsync(tile_start, gstart, (size_t)(ntiles + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice,
                                   stream()));
            entries = stage_entry;
        } else {
            ABACUS_TRY(g_work.entries.reserve(t1 * sizeof(Entry<PT>)));
            entries = g_work.entries.as<Entry<PT>>();
            HIP_TRY(hipMemsetAsync(tile_count, 0, (size_t)(ntiles + 1) * sizeof(unsigned int), stream()));
            const dim3 fgrid((unsigned int)std::max<int64_t>(ceil_div(maxbucket, MS_FCHUNK), 1), (unsigned int)ncoarse);
            ABACUS_LAUNCH("tsc_ms_fine_count", (ms_fine<PT, false>), fgrid, dim3(MS_BLOCK), 0, (const int64_t *)gstart, cshift,
                          ntiles, (const Entry<PT> *)stage_entry, (const unsigned int *)stage_key, tile_count,
                          (const int64_t *)nullptr, (Entry<PT> *)nullptr);
            ABACUS_TRY(exclusive_scan_u32(tile_count, ntiles, tile_start, g_work.scan, 1));
            ABACUS_LAUNCH("tsc_ms_fine_scatter", (ms_fine<PT, true>), fgrid, dim3(MS_BLOCK), 0, (const int64_t *)gstart, cshift,
                          ntiles, (const Entry<PT> *)stage_entry, (const unsigned int *)stage_key, tile_count,
                          (const int64_t *)tile_start, entries);
        }
    } else {
        HIP_TRY(hipMemsetAsync(tile_count, 0, (size_t)(ntiles + 1) * sizeof(unsigned int), stream()));
        const int nblk = (int)std::min<int64_t>(std::max<int64_t>(ceil_div(n, TSC_BLOCK), 1), 256 * 16);
        if (n > 0)
            ABACUS_LAUNCH("tsc_bin_count", (tsc_bin<PT, false, CIC>), dim3(nblk), dim3(TSC_BLOCK), 0, pos, n, weights, g,
                          box, offset, wrap, tile_count, (const int64_t *)nullptr, (Entry<PT> *)nullptr, flag);
        // exclusive scan of the tile counts (also re-zeroes the counters: they become the FILL cursors)
        ABACUS_TRY(exclusive_scan_u32(tile_count, ntiles, tile_start, g_work.scan, 1));
        // list length (needed to size the entry buffer): one 8-byte read-back
        int64_t total = 0;
        HIP_TRY(hipMemcpyAsync(&total, tile_start + ntiles, sizeof(int64_t), hipMemcpyDeviceToHost, stream()));
        int h_flag = 0;
        HIP_TRY(hipMemcpyAsync(&h_flag, flag, sizeof(int), hipMemcpyDeviceToHost, stream()));
        HIP_TRY(hipStreamSynchronize(stream()));
        if (wrapped_out) *wrapped_out = h_flag | lines_wrapped;
        g_wrapped_seen |= h_flag;
        nentries_total = total;
        ABACUS_TRY(g_work.entries.reserve((size_t)std::max<int64_t>(total, 1) * sizeof(Entry<PT>)));
        entries = g_work.entries.as<Entry<PT>>();
        if (n > 0)
            ABACUS_LAUNCH("tsc_bin_fill", (tsc_bin<PT, true, CIC>), dim3(nblk), dim3(TSC_BLOCK), 0, pos, n, weights, g, box,
                          offset, 0, tile_count, (const int64_t *)tile_start, entries, flag);
    }
    if (share && list_mode == 1 && !reuse) {
        g_lists.valid = true;
        g_lists.pos = pos, g_lists.n = n, g_lists.zstride = zstride, g_lists.nentries = nentries_total;
        g_lists.gx = gx, g_lists.gy = gy, g_lists.gz = gz, g_lists.gxg = gxg, g_lists.xoff = xoff, g_lists.xoff2 = xoff2, g_lists.cic = CIC ? 1 : 0;
        g_lists.box = box, g_lists.entries = entries;
    }
    const int dbg = option("dbg_tsc");
    if constexpr (std::is_same<PT, float>::value && std::is_same<GT, float>::value) {
        if (!(dbg & 4) && ntiles >= 4096) {   // persistent workgroups: two 64-KiB tiles per CU
            int dev = 0, ncu = 256;
            HIP_TRY(hipGetDevice(&dev));
            HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
            // ranges of consecutive tiles: long enough to amortise the bounds staging, short enough to fill the chip
            const int range_len = (int)std::min<int64_t>(TP_RANGE, std::max<int64_t>(64, ceil_div(ntiles, (int64_t)ncu * 8)));
            const int nranges = (int)ceil_div(ntiles, range_len);
            const int grid_p = (int)std::min<int64_t>(nranges, (int64_t)ncu * 2);
            // dense lists: more threads per tile shorten the accumulation; sparse ones are flush-bound
            const bool dense = nentries_total / std::max<int64_t>(ntiles, 1) > 400;
            // ACC = float (32-KiB tiles, 4 workgroups per CU) was measured 1.6-4.5x SLOWER (16.3 vs 9.3 ms at 2048^3, both
            // compile to native ds_add_f32 / ds_add_f64): kept float64, which also makes the mesh reproducible run to run
            // full mesh, TSC, positions wrapped into the box by the list build (`wrap`): single-step cell wraps suffice
            const bool fastp_yz = !CIC && wrap && g.gy >= 8 && g.gz >= 8;
            const int fastp = !fastp_yz ? 0 : (g.gxg == g.gx && g.xoff == 0 && g.xoff2 < 0 && g.gx >= 8) ? 1 : (option("tsc_noslabfast") ? 0 : 2);
            // unweighted clouds (every addend in [0, 1]): fixed-point integer tile sums, see acc_add.  S leaves room for n addends
            int fxs = 40;
            while (fxs > 8 && (double)std::max<int64_t>(n, 1) * std::ldexp(1.0, fxs) >= 0x1p62) fxs--;
            const bool fixed = !weights && fxs >= 24;
            const double fxscale = fixed ? std::ldexp(1.0, fxs) : 0.0;
#define LAUNCH_P(NTP, ACC, GRID, FASTP)                                                                               \
    ABACUS_LAUNCH("tsc_tile_deposit", (tsc_tile_deposit_p<TX, TY, TZ, CIC && FASTP == 0, NTP, ACC, FASTP>), dim3(GRID), dim3(NTP), 0, \
                  (const Entry<float> *)entries, (int64_t)nentries_total, (const int64_t *)tile_start, (int)ntiles,   \
                  range_len, g, box, offset, grid, zero_grid, (float)norm, (float)sub, dbg, fxscale)
#define LAUNCH_PA(NTP, GRID, FASTP)                                       \
    do {                                                                  \
        if (fixed) LAUNCH_P(NTP, unsigned long long, GRID, FASTP);        \
        else LAUNCH_P(NTP, double, GRID, FASTP);                          \
    } while (0)
            if (dense && fastp == 1) LAUNCH_PA(512, grid_p, 1);
            else if (dense && fastp == 2) LAUNCH_PA(512, grid_p, 2);
            else if (dense) LAUNCH_PA(512, grid_p, 0);
            else if (fastp == 1) LAUNCH_PA(256, grid_p, 1);
            else if (fastp == 2) LAUNCH_PA(256, grid_p, 2);
            else LAUNCH_PA(256, grid_p, 0);
#undef LAUNCH_PA
#undef LAUNCH_P
            return 0;
        }
    }
    ABACUS_LAUNCH("tsc_tile_deposit", (tsc_tile_deposit<PT, GT, TX, TY, TZ, CIC>), dim3(ntiles), dim3(TSC_BLOCK), 0,
                  (const Entry<PT> *)entries, (const int64_t *)tile_start, g, box, offset, grid, zero_grid, (GT)norm, (GT)sub,
                  dbg);
    return 0;
}

template <typename PT, typename GT, bool CIC>
int deposit_host(void *pos_, int64_t n, const void *weights_, void *grid_, int gx, int gy, int gz, double box,
                 double offset, int wrap) {
    ABACUS_ENTER();
    const size_t cells = (size_t)gx * gy * gz;
    PT *dpos = nullptr, *dw = nullptr;
    GT *dgrid = nullptr;
    int rc = 0, wrapped = 0;
    do {
        // scratch blocks kept between calls (runtime.hip): a tsc_parallel call on NumPy arrays paid hipMalloc + hipFree of the
        // particles and the mesh every time
        if (scratch_acquire((void **)&dpos, std::max<size_t>(3 * n * sizeof(PT), 16)) != 0 ||
            scratch_acquire((void **)&dgrid, cells * sizeof(GT)) != 0 ||
            (weights_ && scratch_acquire((void **)&dw, std::max<size_t>(n * sizeof(PT), 16)) != 0)) {
            rc = fail("tsc: device allocation failed");
            break;
        }
        (void)hipMemcpyAsync(dpos, pos_, 3 * n * sizeof(PT), hipMemcpyHostToDevice, stream());
        if (weights_) (void)hipMemcpyAsync(dw, weights_, n * sizeof(PT), hipMemcpyHostToDevice, stream());
        (void)hipMemcpyAsync(dgrid, grid_, cells * sizeof(GT), hipMemcpyHostToDevice, stream());  // accumulate semantics
        rc = deposit_dev<PT, GT, CIC>(dpos, n, dw, dgrid, gx, gy, gz, gz, box, offset, wrap, 0, 0.0, &wrapped);
        if (rc) break;
        if (hipMemcpyAsync(grid_, dgrid, cells * sizeof(GT), hipMemcpyDeviceToHost, stream()) != hipSuccess) {
            rc = fail("tsc: D2H of the grid failed");
            break;
        }
        if (wrap && wrapped)  // the reference wraps the caller's array in place (tsc.py:171-173)
            (void)hipMemcpyAsync(pos_, dpos, 3 * n * sizeof(PT), hipMemcpyDeviceToHost, stream());
        if (hipStreamSynchronize(stream()) != hipSuccess) rc = fail("tsc: stream synchronisation failed");
    } while (0);
    if (dpos) scratch_release(dpos);
    if (dw) scratch_release(dw);
    if (dgrid) scratch_release(dgrid);
    return rc;
}

}  // namespace

namespace abacus {
void tsc_wrapped_reset() { g_wrapped_seen = 0; }
int tsc_wrapped_seen() { return g_wrapped_seen; }
void tsc_lines_defer(int on) { lines_defer_set(on); }
int tsc_lines_deferred_check() { return lines_deferred_check(); }
// used by power.hip: float32 deposit into a (possibly padded) device mesh with fused normalisation
// list_mode: 0 = lists for this deposit only; 1 = build lists that a following deposit of the same particles shifted by
// up to half a cell can reuse (call with offset 0); 2 = reuse them (rebuilds when anything changed)
int tsc_deposit_f32(float *pos, int64_t n, const float *w, float *grid, int nmesh, int64_t zstride, double box,
                    double offset, int wrap, double norm, int cic, int list_mode, double sub, int zero_grid) {
    if (cic)
        return deposit_dev<float, float, true>(pos, n, w, grid, nmesh, nmesh, nmesh, zstride, box, offset, 0, zero_grid, norm,
                                               nullptr, -1, 0, sub, list_mode);
    return deposit_dev<float, float, false>(pos, n, w, grid, nmesh, nmesh, nmesh, zstride, box, offset, wrap, zero_grid, norm,
                                            nullptr, -1, 0, sub, list_mode);
}
// float64 positions (and weights): the cloud weights are evaluated in the position dtype like the reference does
// (analysis/tsc.py:400 `ftype = positions.dtype.type`), the mesh stays float32
int tsc_deposit_f64pos(double *pos, int64_t n, const double *w, float *grid, int nmesh, int64_t zstride, double box,
                       double offset, int wrap, double norm, int cic, double sub) {
    if (cic)
        return deposit_dev<double, float, true>(pos, n, w, grid, nmesh, nmesh, nmesh, zstride, box, offset, 0, 1, norm,
                                                nullptr, -1, 0, sub, 0);
    return deposit_dev<double, float, false>(pos, n, w, grid, nmesh, nmesh, nmesh, zstride, box, offset, wrap, 1, norm,
                                             nullptr, -1, 0, sub, 0);
}
// float64 MESH (dtype=np.float64 of get_field / calc_power, analysis/power_spectrum.py:808,1148): cloud arithmetic in the position
// dtype, accumulation and normalisation in float64
int tsc_deposit_f64mesh(void *pos, int pos_f64, int64_t n, const void *w, double *grid, int nmesh, int64_t zstride, double box,
                        double offset, int wrap, double norm, int cic, double sub) {
    if (pos_f64) {
        if (cic) return deposit_dev<double, double, true>((double *)pos, n, (const double *)w, grid, nmesh, nmesh, nmesh, zstride, box, offset, 0, 1, norm, nullptr, -1, 0, sub, 0);
        return deposit_dev<double, double, false>((double *)pos, n, (const double *)w, grid, nmesh, nmesh, nmesh, zstride, box, offset, wrap, 1, norm, nullptr, -1, 0, sub, 0);
    }
    if (cic) return deposit_dev<float, double, true>((float *)pos, n, (const float *)w, grid, nmesh, nmesh, nmesh, zstride, box, offset, 0, 1, norm, nullptr, -1, 0, sub, 0);
    return deposit_dev<float, double, false>((float *)pos, n, (const float *)w, grid, nmesh, nmesh, nmesh, zstride, box, offset, wrap, 1, norm, nullptr, -1, 0, sub, 0);
}
// x-slab variant: `grid` holds planes [xoff, xoff + nx_local) (mod nmesh) of the global mesh, ghosts included - or, with
// xoff2 >= 0, two windows of nx_local / 2 planes each starting at xoff and xoff2 (folded slabs) -; written as
// rho*norm - sub (sub = 1: the overdensity's "-1" in every cell; a ghost block is then added to its owner as ghost + 1)
int tsc_deposit_slab_f32(float *pos, int64_t n, const float *w, float *grid, int nmesh, int xoff, int nx_local,
                         int64_t zstride, double box, double offset, int wrap, double norm, int cic, double sub, int xoff2, int nx_alloc) {
    if (cic)
        return deposit_dev<float, float, true>(pos, n, w, grid, nx_local, nmesh, nmesh, zstride, box, offset, 0, 1, norm,
                                               nullptr, nmesh, xoff, sub, 0, xoff2);
    return deposit_dev<float, float, false>(pos, n, w, grid, nx_local, nmesh, nmesh, zstride, box, offset, wrap, 1, norm,
                                            nullptr, nmesh, xoff, sub, 0, xoff2, nx_alloc);
}
int tsc_lines_clocks(unsigned long long *out32) {
    if (!g_lines_clk.p) {
        memset(out32, 0, 32 * sizeof(unsigned long long));
        return 0;
    }
    HIP_TRY(hipMemcpyAsync(out32, g_lines_clk.p, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipMemsetAsync(g_lines_clk.p, 0, 32 * sizeof(unsigned long long), stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}
int tsc_release_work() {
    g_lists.valid = false;
    g_l3.valid = false;
    ABACUS_TRY(g_lw.M.release());
    ABACUS_TRY(g_lw.tot.release());
    ABACUS_TRY(g_lw.tables.release());
    ABACUS_TRY(g_lw.staged.release());
    ABACUS_TRY(g_lw.C.release());
    ABACUS_TRY(g_lw.tile_start.release());
    ABACUS_TRY(g_lw.tile_cnt.release());
    ABACUS_TRY(g_lw.entries.release());
    ABACUS_TRY(g_lw.flag.release());
    ABACUS_TRY(g_work.tile_count.release());
    ABACUS_TRY(g_work.tile_start.release());
    ABACUS_TRY(g_work.entries.release());
    ABACUS_TRY(g_work.flag.release());
    ABACUS_TRY(g_work.scan.release());
    ABACUS_TRY(g_work.gcount.release());
    ABACUS_TRY(g_work.gstart.release());
    ABACUS_TRY(g_work.stage_entry.release());
    ABACUS_TRY(g_work.stage_key.release());
    return 0;
}
}  // namespace abacus

extern "C" {

int abacus_tsc_lines_clocks(unsigned long long *out32) {
    ABACUS_ENTER();
    if (!out32) return fail("abacus_tsc_lines_clocks: null argument");
    return tsc_lines_clocks(out32);
}

int abacus_tsc_deposit(void *pos, int64_t n, const void *weights, int pos_dtype, void *grid, int gx, int gy, int gz,
                       int grid_dtype, double box, double offset, int wrap) {
    if (!grid || (n > 0 && !pos)) return fail("abacus_tsc_deposit: null argument");
    if (pos_dtype == ABACUS_F32 && grid_dtype == ABACUS_F32)
        return deposit_host<float, float, false>(pos, n, weights, grid, gx, gy, gz, box, offset, wrap);
    if (pos_dtype == ABACUS_F64 && grid_dtype == ABACUS_F32)
        return deposit_host<double, float, false>(pos, n, weights, grid, gx, gy, gz, box, offset, wrap);
    if (pos_dtype == ABACUS_F32 && grid_dtype == ABACUS_F64)
        return deposit_host<float, double, false>(pos, n, weights, grid, gx, gy, gz, box, offset, wrap);
    if (pos_dtype == ABACUS_F64 && grid_dtype == ABACUS_F64)
        return deposit_host<double, double, false>(pos, n, weights, grid, gx, gy, gz, box, offset, wrap);
    return fail("abacus_tsc_deposit: unknown dtype code");
}

int abacus_tsc_deposit_dev(float *pos, int64_t n, const float *weights, float *grid, int gx, int gy, int gz,
                           double box, double offset, int wrap, int zero_grid, int cic) {
    ABACUS_ENTER();
    if (cic)
        return deposit_dev<float, float, true>(pos, n, weights, grid, gx, gy, gz, gz, box, offset, 0, zero_grid, 0.0,
                                               nullptr);
    return deposit_dev<float, float, false>(pos, n, weights, grid, gx, gy, gz, gz, box, offset, wrap, zero_grid, 0.0,
                                            nullptr);
}

int abacus_cic_deposit(const void *pos, int64_t n, const void *weights, int pos_dtype, float *grid, int gx, int gy,
                       int gz, double box) {
    if (!grid || (n > 0 && !pos)) return fail("abacus_cic_deposit: null argument");
    if (pos_dtype == ABACUS_F32)
        return deposit_host<float, float, true>(const_cast<void *>(pos), n, weights, grid, gx, gy, gz, box, 0.0, 0);
    if (pos_dtype == ABACUS_F64)
        return deposit_host<double, float, true>(const_cast<void *>(pos), n, weights, grid, gx, gy, gz, box, 0.0, 0);
    return fail("abacus_cic_deposit: unknown dtype code");
}

}  // extern "C"
