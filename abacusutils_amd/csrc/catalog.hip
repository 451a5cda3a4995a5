// Catalogue-side kernels upstream of the HOD: the bit-unpacking of Abacus particle subsamples
// and the local mass environment that prepare_sim ranks halos by.  gfx950 only; all three are streaming / cell-list
// kernels bounded by HBM bandwidth and latency, no matrix work.
//
//   unpack_rvint_k   rvint (3 x int32: 20-bit position | 12-bit velocity) -> pos, vel          (bitpacked.py:32-116)
//   unpack_pids_k    64-bit aux word -> pid, lagr_idx, lagr_pos, tagged, density               (bitpacked.py:118-330)
//   menv_*           M(< r_outer) - M(< r_inner) of neighbour halo mass around every halo above mcut (hod/menv.py:19-87;
//                    the reference's KD-tree ball query + gather-sum becomes a cell list with cells >= max r_outer)
//
// Arithmetic follows the reference's typing: the products are formed in float64 and rounded once into the output
// dtype (int64 * float64 for rvint; uint64 * float32 -> float64 for lagr_pos), so float32 outputs are bit-equal.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/abacus_hip.h"
#include "common.hpp"

namespace abacus {
int exclusive_scan_u32(unsigned int *counters, int64_t n, int64_t *out, DevBuf &scratch, int zero_counters);
}

using namespace abacus;

namespace {

bool on_device(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();   // plain host memory: not an error
        return false;
    }
    return a.type == hipMemoryTypeDevice;
}

// input staging: device pointers are used in place, host buffers are copied into `buf`
template <class T>
int stage_in(const T *src, size_t count, DevBuf &buf, const T **out) {
    if (!src || on_device(src)) {
        *out = src;
        return 0;
    }
    ABACUS_TRY(buf.reserve(std::max<size_t>(count, 1) * sizeof(T)));
    HIP_TRY(hipMemcpyAsync(buf.p, src, count * sizeof(T), hipMemcpyHostToDevice, stream()));
    *out = buf.as<T>();
    return 0;
}
// output staging: a device destination is written directly; for a host destination the kernel writes `buf`
struct OutStage {
    void *host = nullptr, *dev = nullptr;
    size_t bytes = 0;
    int prepare(void *dst, size_t nbytes, DevBuf &buf) {
        host = dev = nullptr;
        bytes = nbytes;
        if (!dst) return 0;
        if (on_device(dst)) {
            dev = dst;
            return 0;
        }
        ABACUS_TRY(buf.reserve(std::max<size_t>(nbytes, 16)));
        dev = buf.p;
        host = dst;
        return 0;
    }
    int finish() {
        if (host && bytes) HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, stream()));
        return 0;
    }
};

constexpr int UB = 256;

// ---- rvint: element-wise over the flat (3N) array, 4 elements (16 B) per thread -------------------------------------
// pos = (x >> 12) * (boxsize / 1e6)  [arithmetic shift], vel = ((x & 0xFFF) - 2048) * (6000 / 2048); both in float64
template <class F>
__global__ __launch_bounds__(UB) void unpack_rvint_k(const int *__restrict__ in, int64_t n3, double posscale,
                                                     double velscale, F *__restrict__ pos, F *__restrict__ vel) {
    const int64_t nvec = n3 / 4;
    for (int64_t v = (int64_t)blockIdx.x * UB + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * UB) {
        const int4 x = reinterpret_cast<const int4 *>(in)[v];
        const int e[4] = {x.x, x.y, x.z, x.w};
        F p[4], u[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            p[q] = (F)((double)(e[q] >> 12) * posscale);
            u[q] = (F)((double)((e[q] & 0xFFF) - 2048) * velscale);
        }
        if (pos) {
#pragma unroll
            for (int q = 0; q < 4; q++) pos[4 * v + q] = p[q];   // contiguous per thread: merged into 16/32-B stores
        }
        if (vel) {
#pragma unroll
            for (int q = 0; q < 4; q++) vel[4 * v + q] = u[q];
        }
    }
    // tail (n3 mod 4 elements)
    const int64_t t = nvec * 4 + (int64_t)blockIdx.x * UB + threadIdx.x;
    if (t < n3) {
        const int e = in[t];
        if (pos) pos[t] = (F)((double)(e >> 12) * posscale);
        if (vel) vel[t] = (F)((double)((e & 0xFFF) - 2048) * velscale);
    }
}

// ---- packed PIDs: one particle per thread -----------------------------------------------------------------------------
constexpr unsigned long long AUXX = 0x7FFFull, AUXY = 0x7FFF0000ull, AUXZ = 0x7FFF00000000ull;
constexpr unsigned long long AUXDENS = 0x07FE000000000000ull;

template <class F>
__global__ __launch_bounds__(UB) void unpack_pids_k(const unsigned long long *__restrict__ packed, int64_t n,
                                                    double inv_ppd, double half, long long *__restrict__ pid,
                                                    F *__restrict__ lagr_pos, short *__restrict__ lagr_idx,
                                                    unsigned char *__restrict__ tagged, F *__restrict__ density) {
    for (int64_t i = (int64_t)blockIdx.x * UB + threadIdx.x; i < n; i += (int64_t)gridDim.x * UB) {
        const unsigned long long a = packed[i];
        const unsigned long long ix = a & AUXX, iy = (a & AUXY) >> 16, iz = (a & AUXZ) >> 32;
        if (pid) pid[i] = (long long)(a & (AUXX | AUXY | AUXZ));
        if (lagr_idx) {
            lagr_idx[3 * i + 0] = (short)ix;
            lagr_idx[3 * i + 1] = (short)iy;
            lagr_idx[3 * i + 2] = (short)iz;
        }
        if (lagr_pos) {   // uint64 * float32 is float64 arithmetic in the reference (:312-314); one rounding on store
            lagr_pos[3 * i + 0] = (F)((double)ix * inv_ppd - half);
            lagr_pos[3 * i + 1] = (F)((double)iy * inv_ppd - half);
            lagr_pos[3 * i + 2] = (F)((double)iz * inv_ppd - half);
        }
        if (tagged) tagged[i] = (unsigned char)((a >> 48) & 1ull);
        if (density) {
            const unsigned long long d = (a & AUXDENS) >> 49;
            density[i] = (F)(d * d);
        }
    }
}

// ---- pack9: 9-byte records, a cell header (first byte 0xFF) governs the particles that follow it ----------------------
// The reference walks the records serially (pack9.py:58-123).  Here a chunk of P9_CH records is a workgroup: pass 1
// counts the particles and notes the last header of every chunk, a scan turns the counts into output offsets, pass 2
// stages the chunk in LDS, finds every record's governing header and output slot with workgroup scans (each thread owns
// P9_PER consecutive records), decodes, and writes the compacted positions / velocities through LDS in coalesced rows.
// Typing as Numba compiles the reference (pinned by tests/ref_data/test_pack9.asdf): invcpd, the cell centres and
// pscale are float64 expressions rounded to the output dtype; the per-particle `short * scale + centre` is arithmetic in
// the output dtype.
constexpr int P9_NT = 256, P9_PER = 6, P9_CH = P9_NT * P9_PER;   // 13.5 KB of records + <= 36 KB of staged output in LDS

__device__ __forceinline__ void p9_shorts(const unsigned char *c, int (&s)[6]) {
    s[0] = ((c[1] & 0x0F) | (c[0] << 4)) - 2048;
    s[1] = (((c[1] & 0xF0) << 4) | c[2]) - 2048;
    s[2] = ((c[4] & 0x0F) | (c[3] << 4)) - 2048;
    s[3] = (((c[4] & 0xF0) << 4) | c[5]) - 2048;
    s[4] = ((c[7] & 0x0F) | (c[6] << 4)) - 2048;
    s[5] = (((c[7] & 0xF0) << 4) | c[8]) - 2048;
}
// record r of a chunk staged in LDS as dwords: its 9 bytes start at byte 9 r, inside three consecutive aligned words -
// three LDS reads and two byte alignments instead of nine byte reads
__device__ __forceinline__ void p9_record(const unsigned int *words, int r, unsigned char (&c)[12]) {
    const int b = r * 9, w = b >> 2, sh = b & 3;
    const unsigned int a0 = words[w], a1 = words[w + 1], a2 = words[w + 2], a3 = sh ? words[w + 3] : 0u;
    const unsigned int v0 = __builtin_amdgcn_alignbyte(a1, a0, (unsigned int)sh), v1 = __builtin_amdgcn_alignbyte(a2, a1, (unsigned int)sh),
                       v2 = __builtin_amdgcn_alignbyte(a3, a2, (unsigned int)sh);
    c[0] = v0 & 0xFF, c[1] = (v0 >> 8) & 0xFF, c[2] = (v0 >> 16) & 0xFF, c[3] = v0 >> 24;
    c[4] = v1 & 0xFF, c[5] = (v1 >> 8) & 0xFF, c[6] = (v1 >> 16) & 0xFF, c[7] = v1 >> 24;
    c[8] = v2 & 0xFF;
}

template <class F>
struct P9Cell {
    F pscale, vscale, c[3];
};

template <class F>
__device__ __forceinline__ P9Cell<F> p9_header(const unsigned char *rec, F boxsize, F velz) {
    P9Cell<F> h;
    if (!rec) {   // no header yet: the reference's state is NaN (:66-71)
        h.pscale = h.vscale = h.c[0] = h.c[1] = h.c[2] = (F)NAN;
        return h;
    }
    int s[6];
    p9_shorts(rec, s);
    const F invcpd = (F)(1.0 / (double)(s[1] + 2000));
    const F csize = boxsize * invcpd;
    const double halfbox = (double)boxsize / 2.0;
    h.vscale = (F)((double)(s[2] + 2000) * 0.0005) * invcpd * velz;
#pragma unroll
    for (int d = 0; d < 3; d++) h.c[d] = (F)(((double)s[3 + d] + 2000.5) * (double)csize - halfbox);
    h.pscale = (F)(0.0005 * (double)csize);
    return h;
}

__device__ __forceinline__ void p9_stage(const unsigned char *__restrict__ data, int64_t rec0, int nrec_chunk,
                                         unsigned int *lds_words) {
    // the chunk starts at a multiple of 9 * P9_CH bytes: dword loads are aligned; the tail is read bytewise
    const int64_t byte0 = rec0 * 9;
    const int nbytes = nrec_chunk * 9, nwords = nbytes / 4;
    const unsigned int *g = reinterpret_cast<const unsigned int *>(data + byte0);
#pragma unroll 7
    for (int q = threadIdx.x; q < nwords; q += P9_NT) lds_words[q] = g[q];
    unsigned char *lb = reinterpret_cast<unsigned char *>(lds_words);
    for (int q = nwords * 4 + threadIdx.x; q < nbytes; q += P9_NT) lb[q] = data[byte0 + q];
}

__global__ __launch_bounds__(P9_NT) void pack9_count(const unsigned char *__restrict__ data, int64_t nrec, int64_t nchunk,
                                                     unsigned int *__restrict__ counts, int *__restrict__ last_hdr) {
    // ONE WAVE per chunk (a workgroup per chunk was dominated by its launch and its barrier: 13.5 KB of work), coalesced dword
    // loads - a chunk starts at a multiple of 9 * P9_CH bytes, 4-byte aligned -: byte k of word q starts a record when
    // 4 q + k is a multiple of 9; q advances by 64 per trip, 4 * 64 = 4 (mod 9)
    const int lane = threadIdx.x & 63;
    const int64_t chunk = (int64_t)blockIdx.x * (P9_NT / 64) + (threadIdx.x >> 6);
    if (chunk >= nchunk) return;
    const int64_t rec0 = chunk * P9_CH;
    const int m = (int)min((int64_t)P9_CH, nrec - rec0);
    unsigned int cnt = 0;
    int last = -1;
    const int nbytes = m * 9, nwords = nbytes >> 2;
    const unsigned int *g = reinterpret_cast<const unsigned int *>(data + rec0 * 9);
    int r9 = (4 * lane) % 9;
#pragma unroll 9
    for (int q = lane; q < nwords; q += 64) {
        const unsigned int v = g[q];
        const int k = r9 ? 9 - r9 : 0;                   // first record start at or after byte 4 q
        if (k < 4) {
            if (((v >> (8 * k)) & 0xFFu) == 0xFFu) last = max(last, (4 * q + k) / 9);
            else cnt++;
        }
        r9 += 4;
        if (r9 >= 9) r9 -= 9;
    }
    for (int b = nwords * 4 + lane; b < nbytes; b += 64)
        if (b % 9 == 0) {
            if (data[rec0 * 9 + b] == 0xFF) last = max(last, b / 9);
            else cnt++;
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        cnt += (unsigned int)__shfl_xor((int)cnt, off, 64);
        last = max(last, __shfl_xor(last, off, 64));
    }
    if (lane == 0) counts[chunk] = cnt, last_hdr[chunk] = last;
}

template <class F>
__global__ __launch_bounds__(P9_NT) void pack9_emit(const unsigned char *__restrict__ data, int64_t nrec, F boxsize, F velz,
                                                    const int64_t *__restrict__ chunk_off,
                                                    const int *__restrict__ last_hdr, F *__restrict__ pos,
                                                    F *__restrict__ vel) {
    __shared__ unsigned int recs[P9_CH * 9 / 4 + 4];
    __shared__ __align__(8) unsigned char stage_raw[P9_CH * 3 * sizeof(F)];
    __shared__ int wave_cnt[P9_NT / 64], wave_hdr[P9_NT / 64];
    __shared__ unsigned char carry_rec[12];
    __shared__ int carry_ok;
    F *stage = reinterpret_cast<F *>(stage_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t rec0 = (int64_t)blockIdx.x * P9_CH;
    const int m = (int)min((int64_t)P9_CH, nrec - rec0);
    p9_stage(data, rec0, m, recs);
    if (tid == 0) {   // the header in force at the start of this chunk: the last one of the nearest earlier chunk that has any
        int64_t c = (int64_t)blockIdx.x - 1;
        while (c >= 0 && last_hdr[c] < 0) c--;
        carry_ok = c >= 0;
        if (c >= 0) {
            const unsigned char *h = data + (c * P9_CH + last_hdr[c]) * 9;
            for (int q = 0; q < 9; q++) carry_rec[q] = h[q];
        }
    }
    __syncthreads();
    const unsigned char *lb = reinterpret_cast<const unsigned char *>(recs);
    // per-thread run of P9_PER consecutive records
    const int r0 = tid * P9_PER;
    int cnt = 0, lasth = -1;
#pragma unroll
    for (int q = 0; q < P9_PER; q++) {
        const int r = r0 + q;
        if (r < m) {
            if (lb[r * 9] == 0xFF) lasth = r;
            else cnt++;
        }
    }
    // workgroup scans: exclusive sum of cnt, exclusive running max of lasth
    int incl = cnt, hmax = lasth;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int a = __shfl_up(incl, off, 64), b = __shfl_up(hmax, off, 64);
        if (lane >= off) incl += a, hmax = max(hmax, b);
    }
    if (lane == 63) wave_cnt[wave] = incl, wave_hdr[wave] = hmax;
    __syncthreads();
    int base = 0, hprev = -1;
    for (int w = 0; w < wave; w++) base += wave_cnt[w], hprev = max(hprev, wave_hdr[w]);
    int total = 0;
    for (int w = 0; w < P9_NT / 64; w++) total += wave_cnt[w];
    const int excl = base + incl - cnt;
    int hbefore = __shfl_up(hmax, 1, 64);
    if (lane == 0) hbefore = -1;
    int cur_h = max(hprev, hbefore);   // header governing this thread's first record (chunk-local index, -1 = carried in)
    const int64_t out0 = chunk_off[blockIdx.x];

    for (int pass = 0; pass < 2; pass++) {   // 0: positions, 1: velocities (one LDS staging buffer)
        F *dst = pass == 0 ? pos : vel;
        if (!dst) continue;
        int h = cur_h, w = excl;
        P9Cell<F> cell = p9_header<F>(h >= 0 ? lb + h * 9 : (carry_ok ? carry_rec : nullptr), boxsize, velz);
#pragma unroll
        for (int q = 0; q < P9_PER; q++) {
            const int r = r0 + q;
            if (r >= m) break;
            unsigned char c[12];
            p9_record(recs, r, c);
            if (c[0] == 0xFF) {
                cell = p9_header<F>(c, boxsize, velz);
                continue;
            }
            int s[6];
            p9_shorts(c, s);
            if (pass == 0) {
#pragma unroll
                for (int d = 0; d < 3; d++) stage[w * 3 + d] = (F)s[d] * cell.pscale + cell.c[d];
            } else {
#pragma unroll
                for (int d = 0; d < 3; d++) stage[w * 3 + d] = (F)s[3 + d] * cell.vscale;
            }
            w++;
        }
        __syncthreads();
        for (int q = tid; q < total * 3; q += P9_NT) dst[out0 * 3 + q] = stage[q];
        __syncthreads();
    }
}

// ---- local mass environment --------------------------------------------------------------------------------------------
struct MenvGrid {
    int nc[3];
    int periodic;
    double lo[3], inv_cell[3], box;
};

// the reference's `pos = (pos + Lbox / 2.0) % Lbox` (menv.py:39) in the dtype of pos, with NumPy's remainder:
// fmod, then + Lbox when the sign differs from the divisor's, +0 for an exact multiple
template <class P>
__device__ __forceinline__ P menv_wrap(P v, double box) {
    const P L = (P)box, h = (P)(box / 2.0);
    const P a = v + h;
    P m = sizeof(P) == 4 ? (P)fmodf((float)a, (float)L) : (P)fmod((double)a, (double)L);
    if (m != (P)0) {
        if (m < (P)0) m += L;
    } else {
        m = (P)0;
    }
    return m;
}

template <class P>
__device__ __forceinline__ int menv_cell(const MenvGrid &g, const P *pos, int64_t i, double (&x)[3]) {
    int c[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        x[d] = (double)(g.periodic ? menv_wrap(pos[3 * i + d], g.box) : pos[3 * i + d]);
        int q = (int)floor((x[d] - g.lo[d]) * g.inv_cell[d]);
        c[d] = q < 0 ? 0 : (q >= g.nc[d] ? g.nc[d] - 1 : q);
    }
    return (c[0] * g.nc[1] + c[1]) * g.nc[2] + c[2];
}

template <class P>
__global__ __launch_bounds__(UB) void menv_count(const P *__restrict__ pos, int64_t n, MenvGrid g,
                                                 unsigned int *__restrict__ counts, unsigned int *__restrict__ cellid) {
    for (int64_t i = (int64_t)blockIdx.x * UB + threadIdx.x; i < n; i += (int64_t)gridDim.x * UB) {
        double x[3];
        const int c = menv_cell(g, pos, i, x);
        cellid[i] = (unsigned int)c;
        atomicAdd(&counts[c], 1u);
    }
}

// sorted record: x, y, z, mass (float64) + original index
template <class P, class M>
__global__ __launch_bounds__(UB) void menv_fill(const P *__restrict__ pos, const M *__restrict__ mass, int64_t n,
                                                const unsigned int *__restrict__ cellid,
                                                const int64_t *__restrict__ start, unsigned int *__restrict__ cursor,
                                                double4 *__restrict__ rec, int *__restrict__ orig, int periodic,
                                                double box) {
    for (int64_t i = (int64_t)blockIdx.x * UB + threadIdx.x; i < n; i += (int64_t)gridDim.x * UB) {
        const unsigned int c = cellid[i];
        const int64_t s = start[c] + atomicAdd(&cursor[c], 1u);
        P x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
        if (periodic) x = menv_wrap(x, box), y = menv_wrap(y, box), z = menv_wrap(z, box);
        rec[s] = make_double4((double)x, (double)y, (double)z, (double)mass[i]);
        orig[s] = (int)i;
    }
}

// one thread per halo in cell order; halos at or below mcut return 0 (menv.py:43,84-85)
template <class R>
__global__ __launch_bounds__(UB) void menv_sum(const double4 *__restrict__ rec, const int *__restrict__ orig, int64_t n,
                                               MenvGrid g, const int64_t *__restrict__ start,
                                               const R *__restrict__ r_inner, int inner_is_array,
                                               const R *__restrict__ r_outer, int outer_is_array, double mcut,
                                               double *__restrict__ out) {
    const int64_t s = (int64_t)blockIdx.x * UB + threadIdx.x;
    if (s >= n) return;
    const double4 me = rec[s];
    const int io = orig[s];
    if (!(me.w > mcut)) {
        out[io] = 0.0;
        return;
    }
    const double ri = (double)r_inner[inner_is_array ? io : 0], ro = (double)r_outer[outer_is_array ? io : 0];
    const double ri2 = ri * ri, ro2 = ro * ro;
    const bool has_inner = ri >= 0.0, has_outer = ro >= 0.0;   // a negative radius matches nothing, like the tree query
    int c[3];
    {
        const double x[3] = {me.x, me.y, me.z};
#pragma unroll
        for (int d = 0; d < 3; d++) {
            int q = (int)floor((x[d] - g.lo[d]) * g.inv_cell[d]);
            c[d] = q < 0 ? 0 : (q >= g.nc[d] ? g.nc[d] - 1 : q);
        }
    }
    const double hb = 0.5 * g.box;
    double so = 0.0, si = 0.0;
    // neighbour range per dimension: the 3-cell stencil, or every cell when a periodic dimension has fewer than 3
    int lo[3], hi[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        if (g.periodic && g.nc[d] < 3) lo[d] = 0, hi[d] = g.nc[d] - 1;
        else if (g.periodic) lo[d] = c[d] - 1, hi[d] = c[d] + 1;
        else lo[d] = max(c[d] - 1, 0), hi[d] = min(c[d] + 1, g.nc[d] - 1);
    }
    for (int ax = lo[0]; ax <= hi[0]; ax++) {
        const int cx = ax < 0 ? ax + g.nc[0] : (ax >= g.nc[0] ? ax - g.nc[0] : ax);
        for (int ay = lo[1]; ay <= hi[1]; ay++) {
            const int cy = ay < 0 ? ay + g.nc[1] : (ay >= g.nc[1] ? ay - g.nc[1] : ay);
            // the z neighbours are adjacent cells in memory except across the periodic wrap: walk them one by one
            for (int az = lo[2]; az <= hi[2]; az++) {
                const int cz = az < 0 ? az + g.nc[2] : (az >= g.nc[2] ? az - g.nc[2] : az);
                const int64_t cc = ((int64_t)cx * g.nc[1] + cy) * g.nc[2] + cz;
                const int64_t j0 = start[cc], j1 = start[cc + 1];
                for (int64_t j = j0; j < j1; j++) {
                    const double4 o = rec[j];
                    double dx = fabs(me.x - o.x), dy = fabs(me.y - o.y), dz = fabs(me.z - o.z);
                    if (g.periodic) {
                        if (dx > hb) dx = g.box - dx;
                        if (dy > hb) dy = g.box - dy;
                        if (dz > hb) dz = g.box - dz;
                    }
                    const double d2 = dx * dx + dy * dy + dz * dz;
                    if (has_outer && d2 <= ro2) so += o.w;
                    if (has_inner && d2 <= ri2) si += o.w;
                }
            }
        }
    }
    out[io] = so - si;
}

int grid_for(int64_t n) { return (int)std::min<int64_t>(std::max<int64_t>(ceil_div(n, UB), 1), 8192); }

}  // namespace

extern "C" int abacus_unpack_rvint(const int32_t *intdata, int64_t n, double boxsize, int out_f64, void *posout,
                                   void *velout) {
    ABACUS_ENTER();
    if (n < 0 || (n > 0 && !intdata)) return fail("abacus_unpack_rvint: null input");
    if (n == 0 || (!posout && !velout)) return 0;
    static DevBuf b_in, b_pos, b_vel;
    const int64_t n3 = 3 * n;
    const size_t fs = out_f64 ? 8 : 4;
    const int32_t *d_in;
    ABACUS_TRY(stage_in(intdata, (size_t)n3, b_in, &d_in));
    if (((uintptr_t)d_in & 15) != 0) return fail("abacus_unpack_rvint: device input must be 16-byte aligned");
    OutStage op, ov;
    ABACUS_TRY(op.prepare(posout, (size_t)n3 * fs, b_pos));
    ABACUS_TRY(ov.prepare(velout, (size_t)n3 * fs, b_vel));
    const double posscale = boxsize / 1e6, velscale = 6000.0 / 2048;   // (:104-105)
    const int grid = grid_for(n3 / 4 + 4);
    if (out_f64)
        ABACUS_LAUNCH("unpack_rvint", unpack_rvint_k<double>, dim3(grid), dim3(UB), 0, d_in, n3, posscale, velscale,
                      static_cast<double *>(op.dev), static_cast<double *>(ov.dev));
    else
        ABACUS_LAUNCH("unpack_rvint", unpack_rvint_k<float>, dim3(grid), dim3(UB), 0, d_in, n3, posscale, velscale,
                      static_cast<float *>(op.dev), static_cast<float *>(ov.dev));
    ABACUS_TRY(op.finish());
    ABACUS_TRY(ov.finish());
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

extern "C" int abacus_unpack_pids(const uint64_t *packed, int64_t n, double box, int64_t ppd, int out_f64, int64_t *pid,
                                  void *lagr_pos, int16_t *lagr_idx, uint8_t *tagged, void *density) {
    ABACUS_ENTER();
    if (n < 0 || (n > 0 && !packed)) return fail("abacus_unpack_pids: null input");
    if (ppd < 1) return fail("abacus_unpack_pids: ppd must be a positive integer");
    if (n == 0) return 0;
    static DevBuf b_in, b_pid, b_pos, b_idx, b_tag, b_den;
    const size_t fs = out_f64 ? 8 : 4;
    const uint64_t *d_in;
    ABACUS_TRY(stage_in(packed, (size_t)n, b_in, &d_in));
    OutStage o_pid, o_pos, o_idx, o_tag, o_den;
    ABACUS_TRY(o_pid.prepare(pid, (size_t)n * 8, b_pid));
    ABACUS_TRY(o_pos.prepare(lagr_pos, (size_t)n * 3 * fs, b_pos));
    ABACUS_TRY(o_idx.prepare(lagr_idx, (size_t)n * 6, b_idx));
    ABACUS_TRY(o_tag.prepare(tagged, (size_t)n, b_tag));
    ABACUS_TRY(o_den.prepare(density, (size_t)n * fs, b_den));
    // inv_ppd = float_dtype(box / ppd), half = float_dtype(box / 2) (:298-299), then float64 arithmetic
    const double q = box / (double)ppd, h = box / 2;
    const double inv_ppd = out_f64 ? q : (double)(float)q, half = out_f64 ? h : (double)(float)h;
    const int grid = grid_for(n);
    const unsigned long long *din = reinterpret_cast<const unsigned long long *>(d_in);
    if (out_f64)
        ABACUS_LAUNCH("unpack_pids", unpack_pids_k<double>, dim3(grid), dim3(UB), 0, din, n, inv_ppd, half,
                      static_cast<long long *>(o_pid.dev), static_cast<double *>(o_pos.dev),
                      static_cast<short *>(o_idx.dev), static_cast<unsigned char *>(o_tag.dev),
                      static_cast<double *>(o_den.dev));
    else
        ABACUS_LAUNCH("unpack_pids", unpack_pids_k<float>, dim3(grid), dim3(UB), 0, din, n, inv_ppd, half,
                      static_cast<long long *>(o_pid.dev), static_cast<float *>(o_pos.dev),
                      static_cast<short *>(o_idx.dev), static_cast<unsigned char *>(o_tag.dev),
                      static_cast<float *>(o_den.dev));
    ABACUS_TRY(o_pid.finish());
    ABACUS_TRY(o_pos.finish());
    ABACUS_TRY(o_idx.finish());
    ABACUS_TRY(o_tag.finish());
    ABACUS_TRY(o_den.finish());
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

extern "C" int abacus_unpack_pack9(const uint8_t *data, int64_t nrec, double boxsize, double velzspace_to_kms, int out_f64,
                                   void *posout, void *velout, int64_t *npart) {
    ABACUS_ENTER();
    if (nrec < 0 || (nrec > 0 && !data) || !npart) return fail("abacus_unpack_pack9: null argument");
    *npart = 0;
    if (nrec == 0) return 0;
    static DevBuf b_in, b_pos, b_vel, b_counts, b_last, b_off, scratch;
    const size_t fs = out_f64 ? 8 : 4;
    const uint8_t *d_in;
    ABACUS_TRY(stage_in(data, (size_t)nrec * 9, b_in, &d_in));
    if (((uintptr_t)d_in & 3) != 0) return fail("abacus_unpack_pack9: device input must be 4-byte aligned");
    OutStage op, ov;
    ABACUS_TRY(op.prepare(posout, (size_t)nrec * 3 * fs, b_pos));
    ABACUS_TRY(ov.prepare(velout, (size_t)nrec * 3 * fs, b_vel));
    const int64_t nchunk = ceil_div(nrec, P9_CH);
    if (nchunk >= ((int64_t)1 << 31)) return fail("abacus_unpack_pack9: too many records");
    ABACUS_TRY(b_counts.reserve((size_t)(nchunk + 1) * 4));
    ABACUS_TRY(b_last.reserve((size_t)nchunk * 4));
    ABACUS_TRY(b_off.reserve((size_t)(nchunk + 1) * 8));
    ABACUS_LAUNCH("pack9_count", pack9_count, dim3((unsigned int)ceil_div(nchunk, P9_NT / 64)), dim3(P9_NT), 0, d_in, nrec, nchunk,
                  b_counts.as<unsigned int>(), b_last.as<int>());
    ABACUS_TRY(exclusive_scan_u32(b_counts.as<unsigned int>(), nchunk, b_off.as<int64_t>(), scratch, 0));
    if (op.dev || ov.dev) {
        if (out_f64)
            ABACUS_LAUNCH("pack9_emit", pack9_emit<double>, dim3((unsigned int)nchunk), dim3(P9_NT), 0, d_in, nrec, boxsize,
                          velzspace_to_kms, b_off.as<int64_t>(), b_last.as<int>(), static_cast<double *>(op.dev),
                          static_cast<double *>(ov.dev));
        else
            ABACUS_LAUNCH("pack9_emit", pack9_emit<float>, dim3((unsigned int)nchunk), dim3(P9_NT), 0, d_in, nrec,
                          (float)boxsize, (float)velzspace_to_kms, b_off.as<int64_t>(), b_last.as<int>(),
                          static_cast<float *>(op.dev), static_cast<float *>(ov.dev));
    }
    int64_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, b_off.as<int64_t>() + nchunk, 8, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    *npart = total;
    // only the first npart rows are defined (the reference returns views of them)
    op.bytes = ov.bytes = (size_t)total * 3 * fs;
    ABACUS_TRY(op.finish());
    ABACUS_TRY(ov.finish());
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

namespace {

template <class P, class M, class R>
int menv_run(const P *pos, const M *mass, int64_t n, const R *r_inner, int inner_n, const R *r_outer, int outer_n,
             const MenvGrid &g, double mcut, double *out_host_or_dev) {
    static DevBuf b_pos, b_mass, b_ri, b_ro, b_counts, b_cellid, b_start, b_rec, b_orig, b_out, scratch;
    const P *d_pos;
    const M *d_mass;
    const R *d_ri, *d_ro;
    ABACUS_TRY(stage_in(pos, (size_t)n * 3, b_pos, &d_pos));
    ABACUS_TRY(stage_in(mass, (size_t)n, b_mass, &d_mass));
    ABACUS_TRY(stage_in(r_inner, (size_t)inner_n, b_ri, &d_ri));
    ABACUS_TRY(stage_in(r_outer, (size_t)outer_n, b_ro, &d_ro));
    const int64_t ncell = (int64_t)g.nc[0] * g.nc[1] * g.nc[2];
    ABACUS_TRY(b_counts.reserve((size_t)(ncell + 1) * 4));
    ABACUS_TRY(b_cellid.reserve((size_t)n * 4));
    ABACUS_TRY(b_start.reserve((size_t)(ncell + 1) * 8));
    ABACUS_TRY(b_rec.reserve((size_t)n * sizeof(double4)));
    ABACUS_TRY(b_orig.reserve((size_t)n * 4));
    OutStage o;
    ABACUS_TRY(o.prepare(out_host_or_dev, (size_t)n * 8, b_out));
    HIP_TRY(hipMemsetAsync(b_counts.p, 0, (size_t)(ncell + 1) * 4, stream()));
    const int grid = grid_for(n);
    ABACUS_LAUNCH("menv_count", (menv_count<P>), dim3(grid), dim3(UB), 0, d_pos, n, g, b_counts.as<unsigned int>(),
                  b_cellid.as<unsigned int>());
    ABACUS_TRY(exclusive_scan_u32(b_counts.as<unsigned int>(), ncell, b_start.as<int64_t>(), scratch, 1));
    ABACUS_LAUNCH("menv_fill", (menv_fill<P, M>), dim3(grid), dim3(UB), 0, d_pos, d_mass, n, b_cellid.as<unsigned int>(),
                  b_start.as<int64_t>(), b_counts.as<unsigned int>(), b_rec.as<double4>(), b_orig.as<int>(), g.periodic, g.box);
    ABACUS_LAUNCH("menv_sum", (menv_sum<R>), dim3((unsigned int)ceil_div(n, UB)), dim3(UB), 0, b_rec.as<double4>(),
                  b_orig.as<int>(), n, g, b_start.as<int64_t>(), d_ri, inner_n != 1, d_ro, outer_n != 1, mcut,
                  static_cast<double *>(o.dev));
    ABACUS_TRY(o.finish());
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

}  // namespace

// pos: (n,3) positions as the caller has them; when periodic the kernels apply the reference's shift into [0, Lbox)
// (menv.py:36-40, NumPy remainder in the dtype of pos).  r_inner / r_outer: one
// value (count 1) or one per halo (count n), in the `r_f64` precision.  Menv: (n) float64.
extern "C" int abacus_menv(const void *pos, int pos_f64, const void *mass, int mass_f64, int64_t n, const void *r_inner,
                           int64_t n_inner, const void *r_outer, int64_t n_outer, int r_f64, double r_outer_max,
                           double Lbox, int periodic, const double *lo, const double *hi, double mcut, double *Menv) {
    ABACUS_ENTER();
    if (n < 0) return fail("abacus_menv: negative count");
    if (n == 0) return 0;
    if (!pos || !mass || !r_inner || !r_outer || !Menv) return fail("abacus_menv: null argument");
    if (n >= ((int64_t)1 << 31)) return fail("abacus_menv: too many halos");
    if ((n_inner != 1 && n_inner != n) || (n_outer != 1 && n_outer != n))
        return fail("abacus_menv: radii must be scalars or one per halo");
    if (periodic && !(Lbox > 0)) return fail("abacus_menv: periodic box needs Lbox > 0");
    if (!periodic && (!lo || !hi)) return fail("abacus_menv: open geometry needs the bounding box");
    if (periodic && r_outer_max > 0.5 * Lbox) return fail("abacus_menv: r_outer exceeds half the box");
    MenvGrid g;
    g.periodic = periodic ? 1 : 0;
    g.box = periodic ? Lbox : 0.0;
    for (int d = 0; d < 3; d++) {
        const double a = periodic ? 0.0 : lo[d], b = periodic ? Lbox : hi[d];
        const double ext = std::max(b - a, 0.0);
        int nc = 1;
        if (r_outer_max > 0 && ext > 0) nc = (int)std::min(std::floor(ext / r_outer_max * 0.9999), 256.0);   // cell >= r_outer
        if (nc < 1) nc = 1;
        g.nc[d] = nc;
        g.lo[d] = a;
        g.inv_cell[d] = ext > 0 ? nc / ext : 0.0;
    }
    const int in_arr = (int)n_inner, out_arr = (int)n_outer;
#define MENV_CALL(P, M, R)                                                                                          \
    return menv_run<P, M, R>(static_cast<const P *>(pos), static_cast<const M *>(mass), n,                          \
                             static_cast<const R *>(r_inner), in_arr, static_cast<const R *>(r_outer), out_arr, g,  \
                             mcut, Menv)
    if (pos_f64) {
        if (mass_f64) {
            if (r_f64) MENV_CALL(double, double, double);
            MENV_CALL(double, double, float);
        }
        if (r_f64) MENV_CALL(double, float, double);
        MENV_CALL(double, float, float);
    }
    if (mass_f64) {
        if (r_f64) MENV_CALL(float, double, double);
        MENV_CALL(float, double, float);
    }
    if (r_f64) MENV_CALL(float, float, double);
    MENV_CALL(float, float, float);
#undef MENV_CALL
}
