// TSC particle -> mesh, "line" lists: what the list build of tsc_lines3.hpp and the tile deposit share.  Included by tsc.hip
// inside its anonymous namespace.  (_tsc_scatter, abacusnbody/analysis/tsc.py:394-507; _wrap_inplace :219-226.)
//
// Measurements that shaped it (profiles/r04/ubench_scatter.txt, scripts/ubench/scatter.hip): MI355X writes pieces of a 128-B
// line at a fraction of the rate of whole aligned lines (scattered aligned runs: 16 B 0.42, 32 B 1.03, 64 B 3.35, 128 B 5.2
// TB/s; a line written as two 64-B halves by consecutive stores of the same lanes: 3.4 - the L2 does not merge them).  So EVERY
// store of a list-building pass is a whole aligned line:
//   split_round    the streaming LDS multisplit of the build's two passes: a workgroup keeps the tail of every bucket's run that
//                  does not fill a line yet in LDS (`carry`) and stores only whole lines;
//   lines_colscan  exclusive scan over the chunks of particles: a chunk's run inside every block (exact, deterministic offsets);
//   lines_fscan    entries per (piece, tile) -> tile lists starting on line boundaries;
//   lines_deposit32 / lines_deposit   tile deposit from the packed entries, integer (fixed-point) LDS sums.
// An entry (8 bytes) is tile-relative: per dimension the nearest cell's index inside the tile (biased by one: a cloud reaches one
// cell over the tile's faces) and d = cell - p as a 16-bit fixed-point number.  p = (x + offset) * (n / L) is evaluated in
// float32 exactly as the reference does; d is a multiple of ulp(p), i.e. of 2^-16 or coarser wherever p >= 128 cells, so
// the 16-bit code is EXACT there and the cloud weights are the reference's float32 weights bit for bit; in the first 128
// cells of a dimension d is rounded to 2^-16 of a cell, up or down without bias (1.5e-5 of a cell at most: the reference's own
// tolerance on the mesh is rtol 1e-4, tests/test_tsc.py:136).  d = +1/2 (p exactly between two cells, round-half-even picked the
// upper) is stored as the lower cell with d = -1/2: the same three weights (1/2, 1/2, 0) on the same cells.
// (The second generation's own build - lines_count / lines_coarse / lines_fcount / lines_fine, one 16-byte record per (particle,
// tile) - was retired in round 6: docs/history.md.)

constexpr int LN_SHX = 4, LN_SHY = 4, LN_SHZ = 5;          // tile = 16 x 16 x 32 cells
constexpr int LN_TX = 16, LN_TY = 16, LN_TZ = 32;
constexpr int LN_ZP = LN_TZ + 4;                           // LDS row: two halo cells either side (no range tests along z)
constexpr unsigned long long LN_INVALID = ~0ull;

struct LGeom {
    int n[3];        // mesh cells
    int nt[3];       // tiles per dimension
    int sb[3];       // log2 of the block shape (tiles)
    int nb[3];       // blocks per dimension
    int nbuckets;    // blocks = coarse buckets
    int tpb;         // tiles per block
    int64_t zstride;
};

// In-cell offsets that need more than 16 bits (the first 128 cells of a dimension) are rounded WITHOUT BIAS: up with a
// probability equal to the fraction dropped, `u` in (0, 1) being a hash of the particle's three coordinates (ln_hash: a function of
// the input alone, so the mesh is reproducible; tsc_lines3.hpp: l3_S).  Round-to-nearest is not good enough: float32 catalogues sit
// on lattices - the benchmark's positions, 24-bit uniforms times L on a power-of-two mesh, are 2^-14-lattice values less one ulp 59 %
// of the time - on which every deterministic rule rounds one way; 4.5e-6 of a cell, the same for every particle below cell 128, is a
// mass dipole of half a particle across that plane and 1.3e-4 in the 2-mode bin of the 1e8-particle spectrum.
// three 10-bit uniforms in (0, 1) from the bits of a particle's coordinates
__device__ __forceinline__ void ln_hash(float x, float y, float z, float u[3]) {
    unsigned int h = __float_as_uint(x) ^ __builtin_rotateleft32(__float_as_uint(y), 11) ^ __builtin_rotateleft32(__float_as_uint(z), 21);
    h *= 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
#pragma unroll
    for (int a = 0; a < 3; a++) u[a] = ((float)((h >> (10 * a)) & 1023u) + 0.5f) * (1.f / 1024.f);
}
struct LnF3 {
    float x, y, z;   // 4-byte aligned: one global_load_dwordx3 per particle
};

// column b of M: exclusive scan over the chunks (in place) and the bucket's total.  One workgroup per bucket
__global__ __launch_bounds__(1024) void lines_colscan(unsigned int *__restrict__ M, int nchunk, int nbuckets,
                                                      unsigned int *__restrict__ tot) {
    __shared__ unsigned int wave_tot[16];
    const int b = blockIdx.x, c = threadIdx.x, lane = c & 63, wv = c >> 6;
    const unsigned int v = c < nchunk ? M[(int64_t)c * nbuckets + b] : 0u;
    unsigned int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    unsigned int before = 0;
    for (int w = 0; w < wv; w++) before += wave_tot[w];
    if (c < nchunk) M[(int64_t)c * nbuckets + b] = before + incl - v;
    if (c == 1023) tot[b] = before + incl;
}

// ---- the streaming split: whole lines out, tails carried in LDS -------------------------------------------------
// A workgroup streams its items in rounds.  Per round: count the new entries per bucket (LDS atomics); per bucket, with
// cc entries carried from earlier rounds and its run position `done`, find how many entries can leave as WHOLE lines
// (wlim: up to the last line boundary; everything in the final round); scan wlim -> slots of the round's output; copy the
// carried entries that leave to their slots, place the new ones at slot or carry; store the output with adjacent lanes on
// adjacent entries of a run - every store instruction covers whole, aligned lines of every bucket it touches.
// exclusive scan of in[0, NB) -> out[] by the NT threads of the workgroup (ITEMS consecutive values per thread)
template <int NB, int NT>
__device__ __forceinline__ unsigned int ln_block_scan(const unsigned int *in, unsigned int *out, unsigned int *wave_tot) {
    constexpr int ITEMS = (NB + NT - 1) / NT;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned int v[ITEMS], sum = 0;
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {
        const int idx = tid * ITEMS + q;
        v[q] = idx < NB ? in[idx] : 0u;
        sum += v[q];
    }
    unsigned int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    unsigned int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) {
        const unsigned int t = wave_tot[w];
        if (w < wv) before += t;
        total += t;
    }
    unsigned int run = before + incl - sum;
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {
        const int idx = tid * ITEMS + q;
        if (idx < NB) out[idx] = run;
        run += v[q];
    }
    return total;
}

// how an entry is kept in LDS: the staged records of the coarse pass are (entry lo, entry hi, key, 0) - three words, so
// that a round of twice as many particles fits the same LDS (the fourth word is restored on the way out)
template <typename E>
struct SplitKeep {
    typedef E type;
    static __device__ __forceinline__ E pack(const E &e) { return e; }
    static __device__ __forceinline__ E unpack(const E &e) { return e; }
};
struct LnU3 {
    unsigned int x, y, z;
};
template <>
struct SplitKeep<uint4> {
    typedef LnU3 type;
    static __device__ __forceinline__ LnU3 pack(const uint4 &e) { return LnU3{e.x, e.y, e.z}; }
    static __device__ __forceinline__ uint4 unpack(const LnU3 &e) { return make_uint4(e.x, e.y, e.z, 0u); }
};

template <typename E, int NB, int LINE, int SBUF, int NT>
struct SplitLds {
    static constexpr int ITEMS = (NB + NT - 1) / NT;   // buckets owned by a thread: tid * ITEMS + q
    typename SplitKeep<E>::type out[SBUF];
    typename SplitKeep<E>::type carry[NB * (LINE - 1)];
    unsigned int gdx[SBUF];         // global index of out[o]
    unsigned int lcnt[2][NB];       // new entries per bucket, by round parity
    unsigned int lcur[NB];          // placement cursors of the round
    unsigned int ccnt[NB], done[NB], base[NB];
    // placement of a bucket's new entries, one 16-byte read: rank k goes to out[x + k] while k < (int)y, else carry[z + k];
    // w: global index of out[o] = w + o
    uint4 pl[NB];
    unsigned int wave_tot[2][NT / 64];
};

// One round.  `count(f)`: f(bucket) for every entry of the calling thread's items; `place(f)`: f(bucket, entry) for the same
// entries (the items sit in registers).  Buckets [last_lo, last_hi) send out everything they hold (the drain after the last
// item).  Returns false - state untouched - when the round's output does not fit SBUF: the caller retries with fewer items.
// Four workgroup barriers; the stores of a round are not waited for: they overlap the loads and the counting of the next.
struct SplitNoMid {
    __device__ __forceinline__ void operator()() const {}
};
// `mid()` runs between the placement and the write-out: work that WAITS for global loads issued a round earlier belongs there.
// (Vector-memory operations retire in order and the compiler cannot count the write-out's stores - their number is not known
// at compile time - so a wait for a load that follows them is a wait for all of them: the third generation's phase clocks,
// scripts/gpu_lines_phases.sh, showed a quarter of a round spent in that wait.)
template <typename E, int NB, int LINE, int SBUF, int NT, typename COUNT, typename PLACE, typename MID = SplitNoMid>
__device__ __forceinline__ bool split_round(SplitLds<E, NB, LINE, SBUF, NT> &s, int nb, int par, int last_lo, int last_hi,
                                            E *__restrict__ dst, COUNT count, PLACE place, int dbg = 0,
                                            unsigned long long *clk = nullptr, MID mid = MID()) {
    constexpr int ITEMS = SplitLds<E, NB, LINE, SBUF, NT>::ITEMS;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // diagnostic (scripts/gpu_lines_phases.sh): shader-clock ticks of thread 0 per phase of the round, summed into clk[0..8]
    unsigned long long tk = clk ? __builtin_amdgcn_s_memtime() : 0ull;
    auto tick = [&](int ph) {
        if (clk && tid == 0) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            atomicAdd(clk + ph, t - tk);
            tk = t;
        }
    };
    count([&](int b) { atomicAdd(&s.lcnt[par][b], 1u); });
    tick(0);
    __syncthreads();
    tick(1);
    // owner step: how many entries of each owned bucket leave as whole lines
    unsigned int w[ITEMS], cc[ITEMS], lc[ITEMS], sum = 0;
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {
        const int b = tid * ITEMS + q;
        w[q] = 0u, cc[q] = 0u, lc[q] = 0u;
        if (b < nb) {
            cc[q] = s.ccnt[b], lc[q] = s.lcnt[par][b];
            const unsigned int avail = cc[q] + lc[q], pos0 = s.base[b] + s.done[b];
            if (b >= last_lo && b < last_hi) w[q] = avail;
            else {
                const unsigned int end = (pos0 + avail) & ~(unsigned int)(LINE - 1);
                w[q] = end > pos0 ? end - pos0 : 0u;
            }
        }
        sum += w[q];
    }
    unsigned int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) s.wave_tot[par][wv] = incl;
    tick(2);
    __syncthreads();
    tick(3);
    unsigned int before = 0, ot = 0;
#pragma unroll
    for (int k = 0; k < NT / 64; k++) {
        const unsigned int t = s.wave_tot[par][k];
        if (k < wv) before += t;
        ot += t;
    }
    if (ot > (unsigned int)SBUF) {
#pragma unroll
        for (int q = 0; q < ITEMS; q++)
            if (tid * ITEMS + q < NB) s.lcnt[par][tid * ITEMS + q] = 0u;
        __syncthreads();
        return false;
    }
    unsigned int run = before + incl - sum;
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {
        const int b = tid * ITEMS + q;
        if (b < NB) {
            s.lcur[b] = 0u;
            s.lcnt[par ^ 1][b] = 0u;
        }
        if (b < nb) {
            const unsigned int pos0 = s.base[b] + s.done[b];
            // w > 0 implies w > cc: a carry never reaches a line boundary
            s.pl[b] = make_uint4(run + cc[q], (unsigned int)((int)w[q] - (int)cc[q]), (unsigned int)(b * (LINE - 1)) + cc[q] - w[q], pos0 - run);
            s.ccnt[b] = cc[q] + lc[q] - w[q];
            s.done[b] += w[q];
            if (w[q] > 0u)                                             // the carried entries leave first
                for (unsigned int j = 0; j < cc[q]; j++) {
                    s.out[run + j] = s.carry[b * (LINE - 1) + j];
                    s.gdx[run + j] = pos0 + j;
                }
        }
        run += w[q];
    }
    tick(4);
    __syncthreads();
    tick(5);
    if (!(dbg & 32)) {
        // two ways to place: `place(f)` with f(bucket, entry) doing everything per entry (an LDS round trip per entry: the rank
        // comes back before the entry can go anywhere), or `place(rank, put)`: the caller takes the ranks of SEVERAL entries first
        // - their atomics are in flight together - and hands each entry over afterwards
        auto rank = [&](int b) { return atomicAdd(&s.lcur[b], 1u); };
        auto put = [&](int b, unsigned int k, const E &e) {
            const uint4 P = s.pl[b];
            if ((int)k < (int)P.y) {
                const unsigned int o = P.x + k;
                s.out[o] = SplitKeep<E>::pack(e);
                s.gdx[o] = P.w + o;
            } else {
                s.carry[P.z + k] = SplitKeep<E>::pack(e);
            }
        };
        if constexpr (std::is_invocable_v<PLACE, decltype(rank), decltype(put)>) place(rank, put);
        else
            place([&](int b, const E &e) {
                const uint4 P = s.pl[b];
                const unsigned int k = atomicAdd(&s.lcur[b], 1u);
                if ((int)k < (int)P.y) {
                    const unsigned int o = P.x + k;
                    s.out[o] = SplitKeep<E>::pack(e);
                    s.gdx[o] = P.w + o;
                } else {
                    s.carry[P.z + k] = SplitKeep<E>::pack(e);
                }
            });
    }
    tick(6);
    __syncthreads();
    tick(7);
    mid();
    tick(9);
    if (!(dbg & 16)) {
        // fixed trip count: the LDS reads of all of a thread's slots are in flight together, then its stores
        constexpr int WO = (SBUF + NT - 1) / NT;
#pragma unroll
        for (int j = 0; j < WO; j++) {
            const unsigned int o = (unsigned int)(j * NT) + tid;
            if (o < ot) dst[(size_t)s.gdx[o]] = SplitKeep<E>::unpack(s.out[o]);
        }
    }
    tick(8);
    return true;
}

template <typename E, int NB, int LINE, int SBUF, int NT>
__device__ __forceinline__ void split_init(SplitLds<E, NB, LINE, SBUF, NT> &s) {
    for (int b = threadIdx.x; b < NB; b += NT) s.ccnt[b] = 0u, s.done[b] = 0u, s.lcnt[0][b] = 0u, s.lcnt[1][b] = 0u, s.lcur[b] = 0u;
}

// after the last item: every carry leaves, slice by slice of buckets so that the output fits the buffer whatever they hold
template <typename E, int NB, int LINE, int SBUF, int NT>
__device__ __forceinline__ void split_drain(SplitLds<E, NB, LINE, SBUF, NT> &s, int nb, int &par, E *__restrict__ dst) {
    constexpr int SL = SBUF / (LINE - 1);
    for (int lo = 0; lo < nb; lo += SL, par ^= 1)
        split_round<E, NB, LINE, SBUF, NT>(s, nb, par, lo, min(lo + SL, nb), dst, [](auto) {}, [](auto) {});
}

// ---- fine level ------------------------------------------------------------------------------------------------
struct LnPiece {
    int bucket;
    unsigned int e0, e1;   // staged entries of the piece
    int first;             // 1: first piece of its bucket
};

__device__ __forceinline__ void ln_bucket_coords(int b, const LGeom &g, int &B0, int &B1, int &B2) {
    B2 = b % g.nb[2];
    B1 = (b / g.nb[2]) % g.nb[1];
    B0 = b / (g.nb[2] * g.nb[1]);
}

// one workgroup per bucket: C[piece][tile] -> the piece's offset inside the tile's list, list starts on line boundaries
// (16 entries), tile_start / tile_cnt of the bucket's tiles.  fstart[b]: first entry slot of the bucket (a multiple of 16)
template <int NBF>
__global__ __launch_bounds__(NBF) void lines_fscan(unsigned int *__restrict__ C, const int *__restrict__ piece_first, LGeom g,
                                                   const unsigned int *__restrict__ fstart, unsigned int *__restrict__ tile_start,
                                                   unsigned int *__restrict__ tile_cnt) {
    __shared__ unsigned int padded[NBF], start[NBF], wave_tot[NBF / 64];
    const int b = blockIdx.x, f = threadIdx.x;
    const int pa = piece_first[b], pb = piece_first[b + 1];
    unsigned int run = 0;
    if (f < g.tpb)
        for (int p = pa; p < pb; p++) {
            const unsigned int c = C[(int64_t)p * g.tpb + f];
            C[(int64_t)p * g.tpb + f] = run;
            run += c;
        }
    padded[f] = f < g.tpb ? (run + 15u) & ~15u : 0u;
    __syncthreads();
    ln_block_scan<NBF, NBF>(padded, start, wave_tot);
    __syncthreads();
    if (f < g.tpb) {
        const unsigned int st = fstart[b] + start[f];
        tile_start[(int64_t)b * g.tpb + f] = st;
        tile_cnt[(int64_t)b * g.tpb + f] = run;
        for (int p = pa; p < pb; p++) C[(int64_t)p * g.tpb + f] += st;
    }
}

// ---- tile deposit from packed entries ----------------------------------------------------------------------------
// The skeleton is tsc_tile_deposit_p's (persistent workgroups over ranges of consecutive tiles - consecutive in the BLOCKED
// order of the lists -, entries of tile t + 2 requested before tile t is flushed, the flush re-zeroes the tile); new:
// the entry decode (no float multiply, no rint, no periodic wrap: the list build resolved them), z rows with a halo of two
// cells either side so that no z index is ever tested, x / y cells outside the tile dropped by predication, integer sums.
constexpr int LD_RANGE = 512;

template <int NT>
__global__ __launch_bounds__(NT) void lines_deposit(const unsigned long long *__restrict__ entries, int64_t capacity,
                                                    const unsigned int *__restrict__ tile_start,
                                                    const unsigned int *__restrict__ tile_cnt, int ntiles, int range_len, LGeom g,
                                                    float *__restrict__ grid, int zero_grid, float norm, float sub, double fxscale,
                                                    int dbg) {
    constexpr int NPRE = 2;                                   // prefetched 16-B loads (two entries each) per thread
    constexpr int FL = LN_TX * LN_TY * (LN_TZ / 4) / NT;      // flush stores (16 B) per thread
    static_assert((LN_TX * LN_TY * (LN_TZ / 4)) % NT == 0 && 2 * FL + NPRE <= 60, "whole flush stores per thread");
    __shared__ __align__(16) unsigned long long tile[LN_TX * LN_TY * LN_ZP];
    __shared__ unsigned int bst[LD_RANGE + 2], bcn[LD_RANGE + 2];
    const int tid = threadIdx.x;
    {
        float4 *t4 = reinterpret_cast<float4 *>(tile);
        for (int q = tid; q < (int)(sizeof(tile) / 16); q += NT) t4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const double fxinv = 1.0 / fxscale;
    const int64_t lastpair = capacity >= 2 ? capacity / 2 - 1 : 0;   // 16-B units that may be read
    const int nranges = (ntiles + range_len - 1) / range_len;
    const int lgtpb = __ffs(g.tpb) - 1;
    for (int r = blockIdx.x; r < nranges; r += gridDim.x) {
        const int t0 = r * range_len, nt = min(range_len, ntiles - t0);
        __syncthreads();
        for (int q = tid; q < nt; q += NT) bst[q] = tile_start[t0 + q], bcn[q] = tile_cnt[t0 + q];
        __syncthreads();
        tsc_v4f X[NPRE], Y[NPRE];
        auto issue = [&](tsc_v4f(&set)[NPRE], int tt) {       // list starts are multiples of 16 entries: aligned 16-B loads
            const int64_t u0 = (int64_t)(bst[tt] >> 1);
#pragma unroll
            for (int q = 0; q < NPRE; q++)
                tsc_gload16_async(set[q], reinterpret_cast<const float4 *>(entries) + min(u0 + q * NT + tid, lastpair));
        };
        auto one = [&](unsigned long long e) {
            const unsigned int lo = (unsigned int)e, hi = (unsigned int)(e >> 32);
            const int lx = lo & 31, ly = (lo >> 5) & 31, lz = min((int)((lo >> 10) & 63), LN_TZ + 1);   // (clamped: an entry never indexes outside the tile's rows)
            const float dx = ((float)(lo >> 16) - 32768.f) * (1.f / 65536.f), dy = ((float)(hi & 0xffffu) - 32768.f) * (1.f / 65536.f),
                        dz = ((float)(hi >> 16) - 32768.f) * (1.f / 65536.f);
            float wx[3], wy[3], wz[3];
            const float d3[3] = {dx, dy, dz};
            float *w3[3] = {wx, wy, wz};
#pragma unroll
            for (int a = 0; a < 3; a++) {                    // _tsc_scatter's weights (tsc.py:428-451), float32, no contraction
                const float d = d3[a], tm = 0.5f + d, tp = 0.5f - d;
                w3[a][1] = 0.75f - d * d;
                w3[a][0] = 0.5f * (tm * tm);
                w3[a][2] = 0.5f * (tp * tp);
            }
            // cells lx - 2 .. lx (the nearest cell is lx - 1); z rows carry the halo: row index lz - 2 + 2 = lz .. lz + 2
            unsigned long long *zrow = tile + lz;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const int cx = lx - 2 + a;
                if ((unsigned)cx >= (unsigned)LN_TX) continue;
#pragma unroll
                for (int b = 0; b < 3; b++) {
                    const int cy = ly - 2 + b;
                    if ((unsigned)cy >= (unsigned)LN_TY) continue;
                    const float wxy = wx[a] * wy[b];
                    unsigned long long *cell = zrow + (cx * LN_TY + cy) * LN_ZP;
#pragma unroll
                    for (int c = 0; c < 3; c++) acc_add<unsigned long long>(cell + c, wxy * wz[c], fxscale);   // (wx wy) wz, tsc.py:471-507
                }
            }
        };
        bool prev_plain = false;
        auto process = [&](tsc_v4f(&cur)[NPRE], int tt) {
            const int tile_id = t0 + tt;
            const int b = tile_id >> lgtpb, f = tile_id & (g.tpb - 1);
            int B0, B1, B2;
            ln_bucket_coords(b, g, B0, B1, B2);
            const int fz = f & ((1 << g.sb[2]) - 1), fy = (f >> g.sb[2]) & ((1 << g.sb[1]) - 1), fx = f >> (g.sb[2] + g.sb[1]);
            const int ox = ((B0 << g.sb[0]) + fx) << LN_SHX, oy = ((B1 << g.sb[1]) + fy) << LN_SHY, oz = ((B2 << g.sb[2]) + fz) << LN_SHZ;
            const unsigned int cnt = bcn[tt];
            const int64_t e0 = bst[tt];
#pragma unroll
            for (int q = 0; q < NPRE; q++) tsc_touch(cur[q]);
#pragma unroll
            for (int q = 0; q < NPRE; q++) {
                const unsigned int k = 2u * (q * NT + tid);
                if (!(dbg & 1)) {
                    if (k < cnt) one(((unsigned long long)__float_as_uint(cur[q].y) << 32) | __float_as_uint(cur[q].x));
                    if (k + 1 < cnt) one(((unsigned long long)__float_as_uint(cur[q].w) << 32) | __float_as_uint(cur[q].z));
                }
            }
            for (unsigned int k = 2u * NPRE * NT + tid; k < cnt; k += NT) one(entries[e0 + k]);
            const bool extra = cnt > 2u * NPRE * NT;
            __syncthreads();
            const bool more = tt + 2 < nt;
            if (more) issue(cur, tt + 2);
            {
                constexpr int ZQ = LN_TZ / 4, ROWS = NT / ZQ;
                static_assert(NT % ZQ == 0 && ROWS % LN_TY == 0 && (LN_TX * LN_TY) % ROWS == 0, "flush mapping");
                constexpr int XSTEP = ROWS / LN_TY, STEPS = LN_TX / XSTEP;
                const int zq = tid & (ZQ - 1), yy = (tid / ZQ) & (LN_TY - 1), x0 = tid / (ZQ * LN_TY);
                unsigned long long *cell = &tile[(x0 * LN_TY + yy) * LN_ZP + 2 + 4 * zq];
                float *dst = grid + ((int64_t)(ox + x0) * g.n[1] + (oy + yy)) * g.zstride + oz + 4 * zq;
                const int64_t dstep = (int64_t)XSTEP * g.n[1] * g.zstride;
#pragma unroll
                for (int st = 0; st < STEPS; st++, cell += XSTEP * LN_TY * LN_ZP, dst += dstep) {
                    ulonglong2 *c2 = reinterpret_cast<ulonglong2 *>(cell);
                    const ulonglong2 a01 = c2[0], a23 = c2[1];
                    c2[0] = make_ulonglong2(0ull, 0ull);
                    c2[1] = make_ulonglong2(0ull, 0ull);
                    if (zq == 0) c2[-1] = make_ulonglong2(0ull, 0ull);          // the row's halo cells
                    if (zq == ZQ - 1) c2[2] = make_ulonglong2(0ull, 0ull);
                    double a[4] = {(double)a01.x * fxinv, (double)a01.y * fxinv, (double)a23.x * fxinv, (double)a23.y * fxinv};
                    if (!zero_grid) {
                        const float4 old = *reinterpret_cast<const float4 *>(dst);
                        a[0] += (double)old.x, a[1] += (double)old.y, a[2] += (double)old.z, a[3] += (double)old.w;
                    }
                    float v[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) v[c] = (float)a[c];
                    if (norm != 0.f) {
#pragma unroll
                        for (int c = 0; c < 4; c++) v[c] = v[c] * norm - sub;
                    }
                    if (!(dbg & 2)) *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
            __syncthreads();
            // see tsc_tile_deposit_p: [FL stores of tile tt-1] [NPRE loads of tile tt+2] [FL stores of this tile]
            const bool plain = zero_grid && !extra && !(dbg & 2);
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

// ---- the same deposit with 32-bit tile sums ------------------------------------------------------------------------
// The u64 deposit spends 0.95 ms of its 2.0 ms (BASELINE config 3) in LDS atomics and 1.1 ms in the flush, one after the
// other: two 74-KB workgroups per CU run in lock step.  Here a cell is a 32-bit fixed-point sum with a PER-TILE scale
// 2^S, S = 32 - floor(log2(entries of the tile)): no cell can overflow (an addend is at most 0.75^3 = 0.42), and the
// resolution is ~2e-6 of the tile's MEAN cell whatever its density (2^-23 absolute at the 950 entries of a config-3 tile;
// the reference's own float32 mesh carries 6e-8 relative per add, its tolerance on the mesh is rtol 1e-4).  Two z-adjacent
// cells form one aligned 64-bit word, so the three cells of a cloud row take TWO 64-bit LDS atomics (the low half cannot
// carry into the high one) instead of three: 18 per entry instead of 27.  The tile is 36 KB: four workgroups per CU,
// whose accumulate and flush phases overlap.  Integer sums: the mesh does not depend on the order of the entries.
constexpr int LD32_RANGE = 256;

template <int NT>
__global__ __launch_bounds__(NT) void lines_deposit32(const unsigned long long *__restrict__ entries, int64_t capacity,
                                                      const unsigned int *__restrict__ tile_start,
                                                      const unsigned int *__restrict__ tile_cnt, int ntiles, int range_len, LGeom g,
                                                      float *__restrict__ grid, int zero_grid, float norm, float sub, int dbg) {
    constexpr int NPRE = 3;                                   // prefetched 16-B loads (two entries each) per thread
    constexpr int FL = LN_TX * LN_TY * (LN_TZ / 4) / NT;      // flush stores (16 B) per thread
    static_assert((LN_TX * LN_TY * (LN_TZ / 4)) % NT == 0 && 2 * FL + NPRE <= 60, "whole flush stores per thread");
    __shared__ __align__(16) unsigned int tile[LN_TX * LN_TY * LN_ZP];
    __shared__ unsigned int bst[LD32_RANGE + 2], bcn[LD32_RANGE + 2];
    const int tid = threadIdx.x;
    {
        float4 *t4 = reinterpret_cast<float4 *>(tile);
        for (int q = tid; q < (int)(sizeof(tile) / 16); q += NT) t4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int64_t lastpair = capacity >= 2 ? capacity / 2 - 1 : 0;
    const int nranges = (ntiles + range_len - 1) / range_len;
    const int lgtpb = __ffs(g.tpb) - 1;
    for (int r = blockIdx.x; r < nranges; r += gridDim.x) {
        const int t0 = r * range_len, nt = min(range_len, ntiles - t0);
        __syncthreads();
        for (int q = tid; q < nt; q += NT) bst[q] = tile_start[t0 + q], bcn[q] = tile_cnt[t0 + q];
        __syncthreads();
        tsc_v4f X[NPRE], Y[NPRE];
        // the NPRE loads per thread are issued whatever the list holds (the counted vmcnt waits below rely on it); lanes
        // beyond the list read its first pair again - one request per wave instead of 1 KB of somebody else's entries
        // (2048^3: 128 entries per tile on average against the 1536 the loads cover, 4.6 GB of useless requests)
        auto issue = [&](tsc_v4f(&set)[NPRE], int tt) {
            const int64_t u0 = (int64_t)(bst[tt] >> 1);
            const int np2 = (int)((bcn[tt] + 1u) >> 1);
#pragma unroll
            for (int q = 0; q < NPRE; q++) {
                const int k = q * NT + tid;
                tsc_gload16_async(set[q], reinterpret_cast<const float4 *>(entries) + min(u0 + (k < np2 ? k : 0), lastpair));
            }
        };
        float fx = 1.f;                                       // 2^S of the tile being accumulated
        auto one = [&](unsigned long long e) {
            const unsigned int lo = (unsigned int)e, hi = (unsigned int)(e >> 32);
            const int lx = lo & 31, ly = (lo >> 5) & 31, lz = min((int)((lo >> 10) & 63), LN_TZ + 1);
            const float dx = ((float)(lo >> 16) - 32768.f) * (1.f / 65536.f), dy = ((float)(hi & 0xffffu) - 32768.f) * (1.f / 65536.f),
                        dz = ((float)(hi >> 16) - 32768.f) * (1.f / 65536.f);
            float wx[3], wy[3], wz[3];
            const float d3[3] = {dx, dy, dz};
            float *w3[3] = {wx, wy, wz};
#pragma unroll
            for (int a = 0; a < 3; a++) {                    // _tsc_scatter's weights (tsc.py:428-451), float32, no contraction
                const float d = d3[a], tm = 0.5f + d, tp = 0.5f - d;
                w3[a][1] = 0.75f - d * d;
                w3[a][0] = 0.5f * (tm * tm);
                w3[a][2] = 0.5f * (tp * tp);
            }
            // the three z weights in the four cells of the two aligned pairs they fall into (one of the outer two is empty): decided
            // once per entry, so that a row costs four scaled products (packed two by two) and no selects (SQ counters: the kernel
            // keeps the vector units 2/3 busy)
            typedef float ln_v2f __attribute__((ext_vector_type(2)));
            const bool odd = lz & 1;
            const ln_v2f wzA = {odd ? 0.f : wz[0], odd ? wz[0] : wz[1]}, wzB = {odd ? wz[1] : wz[2], odd ? wz[2] : 0.f};
            const ln_v2f fx2 = {fx, fx}, half2 = {0.5f, 0.5f};
            unsigned long long *zpair = reinterpret_cast<unsigned long long *>(tile) + (lz >> 1);   // cells lz .. lz + 2 of a halo'd row
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const int cx = lx - 2 + a;
                if ((unsigned)cx >= (unsigned)LN_TX) continue;
#pragma unroll
                for (int b = 0; b < 3; b++) {
                    const int cy = ly - 2 + b;
                    if ((unsigned)cy >= (unsigned)LN_TY) continue;
                    const float wxy = wx[a] * wy[b];
                    const ln_v2f w2 = {wxy, wxy};
                    // (wx wy) wz, tsc.py:471-507, as packed float32 operations (v_pk_mul_f32 / v_pk_fma_f32: the same roundings)
                    const ln_v2f sA = __builtin_elementwise_fma(w2 * wzA, fx2, half2), sB = __builtin_elementwise_fma(w2 * wzB, fx2, half2);
                    unsigned long long *cell = zpair + (cx * LN_TY + cy) * (LN_ZP / 2);
                    atomicAdd(cell, ((unsigned long long)(unsigned int)sA.y << 32) | (unsigned int)sA.x);
                    atomicAdd(cell + 1, ((unsigned long long)(unsigned int)sB.y << 32) | (unsigned int)sB.x);
                }
            }
        };
        bool prev_plain = false;
        auto process = [&](tsc_v4f(&cur)[NPRE], int tt) {
            const int tile_id = t0 + tt;
            const int b = tile_id >> lgtpb, f = tile_id & (g.tpb - 1);
            int B0, B1, B2;
            ln_bucket_coords(b, g, B0, B1, B2);
            const int fz = f & ((1 << g.sb[2]) - 1), fy = (f >> g.sb[2]) & ((1 << g.sb[1]) - 1), fx_ = f >> (g.sb[2] + g.sb[1]);
            const int ox = ((B0 << g.sb[0]) + fx_) << LN_SHX, oy = ((B1 << g.sb[1]) + fy) << LN_SHY, oz = ((B2 << g.sb[2]) + fz) << LN_SHZ;
            const unsigned int cnt = bcn[tt];
            const int64_t e0 = bst[tt];
            // a list beyond 2^17 entries (a pile-up: millions of particles in one tile) is accumulated and flushed in slices of
            // 2^17, the mesh taking the float32 sum of the slices: the scale of a slice never drops below 2^15 (with ONE scale
            // for 2.3e6 entries, 2^11, the addends below 2^-12 - the corners of every cloud - were lost: 6e-5 of the mass)
            constexpr unsigned int SLICE = 1u << 17;
            const unsigned int nslice = cnt > SLICE ? (cnt + SLICE - 1) / SLICE : 1u;
            const int S = min(32 - (31 - __clz((int)max(min(cnt, SLICE), 1u))), 30);
            fx = __uint_as_float((unsigned int)(127 + S) << 23);
            const float fxinv = __uint_as_float((unsigned int)(127 - S) << 23);
#pragma unroll
            for (int q = 0; q < NPRE; q++) tsc_touch(cur[q]);
#pragma unroll
            for (int q = 0; q < NPRE; q++) {
                const unsigned int k = 2u * (q * NT + tid);
                if (!(dbg & 1)) {
                    if (k < cnt) one(((unsigned long long)__float_as_uint(cur[q].y) << 32) | __float_as_uint(cur[q].x));
                    if (k + 1 < cnt) one(((unsigned long long)__float_as_uint(cur[q].w) << 32) | __float_as_uint(cur[q].z));
                }
            }
            const bool extra = cnt > 2u * NPRE * NT;
            const bool more = tt + 2 < nt;
            for (unsigned int sl = 0; sl < nslice; sl++) {
            const bool first_sl = sl == 0, last_sl = sl + 1 == nslice;
            {
                const unsigned int k0 = first_sl ? 2u * NPRE * NT : sl * SLICE, k1 = last_sl ? cnt : (sl + 1) * SLICE;
                for (unsigned int k = k0 + tid; k < k1; k += NT) one(entries[e0 + k]);
            }
            __syncthreads();
            if (more && last_sl) issue(cur, tt + 2);
            {
                constexpr int ZQ = LN_TZ / 4, ROWS = NT / ZQ;
                static_assert(NT % ZQ == 0 && ROWS % LN_TY == 0 && (LN_TX * LN_TY) % ROWS == 0, "flush mapping");
                constexpr int XSTEP = ROWS / LN_TY, STEPS = LN_TX / XSTEP;
                const int zq = tid & (ZQ - 1), yy = (tid / ZQ) & (LN_TY - 1), x0 = tid / (ZQ * LN_TY);
                unsigned int *cell = &tile[(x0 * LN_TY + yy) * LN_ZP + 2 + 4 * zq];     // 8-byte aligned: two 8-byte LDS accesses
                float *dst = grid + ((int64_t)(ox + x0) * g.n[1] + (oy + yy)) * g.zstride + oz + 4 * zq;
                const int64_t dstep = (int64_t)XSTEP * g.n[1] * g.zstride;
#pragma unroll
                for (int st = 0; st < STEPS; st++, cell += XSTEP * LN_TY * LN_ZP, dst += dstep) {
                    uint2 *c2 = reinterpret_cast<uint2 *>(cell);
                    const uint2 a01 = c2[0], a23 = c2[1];
                    c2[0] = make_uint2(0u, 0u);
                    c2[1] = make_uint2(0u, 0u);
                    if (zq == 0) c2[-1] = make_uint2(0u, 0u);                            // the row's halo cells
                    if (zq == ZQ - 1) c2[2] = make_uint2(0u, 0u);
                    float v[4] = {(float)a01.x * fxinv, (float)a01.y * fxinv, (float)a23.x * fxinv, (float)a23.y * fxinv};
                    if (!zero_grid || !first_sl) {
                        const float4 old = *reinterpret_cast<const float4 *>(dst);
                        v[0] += old.x, v[1] += old.y, v[2] += old.z, v[3] += old.w;
                    }
                    if (norm != 0.f && last_sl) {
#pragma unroll
                        for (int c = 0; c < 4; c++) v[c] = v[c] * norm - sub;
                    }
                    if (!(dbg & 2)) *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
            __syncthreads();
            if (!last_sl) tsc_wait_vmcnt<0>();      // the next slice's flush reads what this one wrote
            }
            const bool plain = zero_grid && !extra && !(dbg & 2);
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
