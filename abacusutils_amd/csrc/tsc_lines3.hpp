// TSC particle -> mesh, third generation of the list build ("block records"): included by tsc.hip after tsc_lines.hpp,
// whose streaming split (split_round), tile-list layout, fscan and deposit kernels it keeps.
// (_tsc_scatter, abacusnbody/analysis/tsc.py:394-507; _wrap_inplace :219-226; the half-cell-shifted second deposit of
// get_interlaced_field_fft, abacusnbody/analysis/power_spectrum.py:951-998.)
//
// What was wrong with the second generation (profiles/r04, rocprofv3 + SQ counters): its coarse pass is bound by
// vector-instruction issue - ~450 instructions per particle, half of them the per-particle loops over the 1 .. 8 tiles of a
// cloud, run twice (count, place) by every lane of a wave as often as its busiest lane needs - and it stages one 16-byte
// record per (particle, TILE): 1.35 records per particle written once and read twice.
// Here the coarse level stops at the BLOCK (8 x 8 x 8 or 16 x 8 x 8 tiles, the coarse bucket):
//   lines3_count    wraps the positions, counts (particle, block) records per chunk and block, and the tile entries each
//                   block will hold (what sizes the entry lists);
//   lines3_coarse   per particle and dimension ONE fixed-point coordinate S = floor(p 2^16) + 32767, p = (x + offset) n / L in
//                   float32 as the reference evaluates it (tsc.py:419-421): S >> 16 is the nearest cell and ~S & 0xffff
//                   the 16-bit code of the in-cell offset the tile entries carry (tsc_lines.hpp: exact wherever p >= 128
//                   cells, rounded without bias below).  A record is the three coordinates relative to its block
//                   (biased by two cells), one per block the cloud touches: 1.05 per particle;
//   lines3_fcount / lines3_fine   decode a record with shifts, enumerate the tiles of the cloud INSIDE the block (integer
//                   compares; a cloud's part in the neighbouring block is that block's record) and split by tile.
// The half-cell-shifted deposit of an interlaced pair is the same record with 0x8000 added to every coordinate: built with
// EXT (the 4-cell union of both clouds decides the blocks), one count + coarse pass serves both deposits; fcount / fine /
// deposit run per origin (ORG).  Where p is exact the shifted coordinate p + 1/2 is what float32 (x + d/2) n/L gives; elsewhere
// it differs from it by the reference's own rounding of that sum (<= 1 ulp of p).
//
// Entry format, tile lists, fscan and deposit: tsc_lines.hpp, unchanged.

struct L3Dim {
    int S;          // fixed-point coordinate + 32767 (cell = S >> 16)
    int Blo;        // unwrapped block index of the cloud's lowest cell (offset by n cells: non-negative)
    bool two;       // the cloud reaches into the next block
    int nt;         // tiles the cloud touches in this dimension (1 or 2)
};

// coordinate of one dimension; `u` in (0, 1) decides the rounding of the bits a 2^-16 grid drops
// ih16 = 65536 n / L: the power of two scales the float32 product (x + offset) n / L of tsc.py:419-421 exactly
__device__ __forceinline__ int l3_S(float c, float offset, float ih16, int n, float u) {
    float y = (c + offset) * ih16;
    // nearest cells -2 .. n + 2 (p in (-2.5, n + 2.5): positions inside the box at offsets up to two cells); garbage positions
    // (NaN, inf) stay inside the tables
    y = fminf(fmaxf(y, -163839.f), (float)(n + 2) * 65536.f + 32000.f);
    const float fl = floorf(y);                                // exact
    return (int)fl + ((y - fl) > u ? 1 : 0) + 32767;
}

template <bool EXT>
__device__ __forceinline__ void l3_dim(int S, int n, int lgbc, int sh, L3Dim &d) {
    const int i = S >> 16;                                     // [-2, n + 2]
    const int lo = i - 1 + n, hi = i + (EXT ? 2 : 1) + n;      // cloud cells, offset by n (>= n - 3 > 0)
    d.S = S;
    d.Blo = lo >> lgbc;
    d.two = d.Blo != (hi >> lgbc);
    d.nt = 1 + ((lo >> sh) != (hi >> sh) ? 1 : 0);
}
__device__ __forceinline__ int l3_wrapB(int B, int nbk) {      // B <= 2 nbk
    B -= B >= nbk ? nbk : 0;
    B -= B >= nbk ? nbk : 0;
    return B;
}

// x-windows of a slab-decomposed mesh (analysis/slab_power.py: a rank's folded pair of slabs with their ghost planes, stored
// back to back and padded to whole tiles): the local plane of global cell i is w * win + ((i - xoff_w) mod n).  A particle is
// listed when its whole cloud lies inside one window (the particles a rank owns always do: ghost planes >= the cloud's reach);
// the first window wins where both hold it; the others are skipped.  on = 0: the mesh is the whole periodic box.
struct L3Win {
    int on, n, win, xoff, xoff2;
};
// S of the x dimension -> the same coordinate in local planes; false: the particle is not listed
template <bool EXT>
__device__ __forceinline__ bool l3_window(const L3Win &wn, int &S) {
    if (!wn.on) return true;
    const int i = S >> 16, reach = EXT ? 2 : 1;
    int u = i - wn.xoff;
    u += u < 0 ? wn.n : 0;
    u -= u >= wn.n ? wn.n : 0;
    int base = 0;
    bool in = u >= 1 && u + reach <= wn.win - 1;
    if (!in && wn.xoff2 >= 0) {
        u = i - wn.xoff2;
        u += u < 0 ? wn.n : 0;
        u -= u >= wn.n ? wn.n : 0;
        in = u >= 1 && u + reach <= wn.win - 1;
        base = wn.win;
    }
    S += (u + base - i) * 65536;
    return in;
}

// ---- counting pass ---------------------------------------------------------------------------------------------
// M[c][b] = records of chunk c in block b; ent[b] += tile entries of block b (EXT: of the 4-cell union cloud, a bound for both
// origins).  Four particles per thread in flight (the second generation's one-particle loop ran at 2.7 TB/s: latency)
// What the counting pass needs of a coordinate is a function of its nearest cell alone: a table per dimension, built in LDS by
// every workgroup (3 (n + 5) words: cells -2 .. n + 2), replaces ~35 vector instructions per particle and dimension (block of
// the cloud's two ends, periodic wraps, the window map of a slab mesh) by one LDS read - the pass was bound by vector issue
// (SQ counters, profiles/r05): 170 instructions per particle.  Entry: bits 0..9 the block's bucket contribution / stride,
// 10..20 the bucket delta of the second block / stride + 1024, 21 two blocks, 22 two tiles, 23 not listed (slab windows)
template <bool EXT>
__device__ __forceinline__ unsigned int l3_count_entry(int i, int a, const LGeom &g, const L3Win &wn) {
    const int sh[3] = {LN_SHX, LN_SHY, LN_SHZ};
    const int lgbc = sh[a] + g.sb[a];
    int S = i * 65536 + 32768;                              // any coordinate whose nearest cell is i
    const bool listed = a == 0 ? l3_window<EXT>(wn, S) : true;
    L3Dim d;
    l3_dim<EXT>(S, g.n[a], lgbc, sh[a], d);
    const int B = l3_wrapB(max(d.Blo, 0), g.nb[a]);
    const int dB = (B + 1 == g.nb[a] ? 0 : B + 1) - B;
    return (unsigned int)B | ((unsigned int)(dB + 1024) << 10) | (d.two ? 1u << 21 : 0u) | (d.nt == 2 ? 1u << 22 : 0u) | (listed ? 0u : 1u << 23);
}

template <int NB, bool EXT>
__global__ __launch_bounds__(512) void lines3_count(float *__restrict__ pos, int64_t n, LGeom g, double box, float offA, int wrap,
                                                    int64_t CH, unsigned int *__restrict__ M, unsigned int *__restrict__ ent,
                                                    int *__restrict__ wrapped_flag, L3Win wn) {
    __shared__ unsigned int hrec[NB], hent[NB];
    extern __shared__ unsigned int l3_tab[];              // [3][tabn]
    const int tid = threadIdx.x;
    // (x: cells of the GLOBAL mesh - a slab's local mesh g.n[0] is only its windows)
    const int ncell[3] = {wn.on ? wn.n : g.n[0], g.n[1], g.n[2]};
    const int tabn = max(ncell[0], max(ncell[1], ncell[2])) + 5;
    for (int b = tid; b < NB; b += 512) hrec[b] = 0u, hent[b] = 0u;
    for (int q = tid; q < 3 * tabn; q += 512) {
        const int a = q / tabn, i = q - a * tabn - 2;
        l3_tab[q] = i <= ncell[a] + 2 ? l3_count_entry<EXT>(i, a, g, wn) : 0u;
    }
    __syncthreads();
    const float ih[3] = {(float)(ncell[0] / box) * 65536.f, (float)(g.n[1] / box) * 65536.f, (float)(g.n[2] / box) * 65536.f};
    const int bst[3] = {g.nb[1] * g.nb[2], g.nb[2], 1};
    const int64_t p0 = (int64_t)blockIdx.x * CH, p1 = min(p0 + CH, n);
    bool any_changed = false;
    // positions in [0, boxlo] are inside the box whatever float32 makes of `box`: the float64 comparisons of _wrap_inplace
    // (tsc.py:219-226) only for the others
    const float boxlo = nextafterf((float)box, 0.f);
    constexpr int U = 4;
    for (int64_t pb = p0; pb < p1; pb += U * 512) {
        LnF3 q[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
            const int64_t p = pb + k * 512 + tid;
            if (p < p1) q[k] = *reinterpret_cast<const LnF3 *>(pos + 3 * p);
        }
#pragma unroll
        for (int k = 0; k < U; k++) {
            const int64_t p = pb + k * 512 + tid;
            if (p >= p1) continue;
            if (wrap && !(fminf(fminf(q[k].x, q[k].y), q[k].z) >= 0.f && fmaxf(fmaxf(q[k].x, q[k].y), q[k].z) <= boxlo)) {
                bool ch = false;
                q[k].x = wrap1(q[k].x, box, ch), q[k].y = wrap1(q[k].y, box, ch), q[k].z = wrap1(q[k].z, box, ch);
                // EXT: this build also serves the second deposit of an interlaced pair, before which the reference wraps the
                // (wrapped) positions AGAIN (get_interlaced_field_fft calls tsc_parallel twice, power_spectrum.py:980-986): a
                // coordinate just below zero that the first wrap rounded up to `box` itself comes back as 0
                if (EXT) q[k].x = wrap1(q[k].x, box, ch), q[k].y = wrap1(q[k].y, box, ch), q[k].z = wrap1(q[k].z, box, ch);
                if (ch) {
                    *reinterpret_cast<LnF3 *>(pos + 3 * p) = q[k];
                    any_changed = true;
                }
            }
            const float c[3] = {q[k].x, q[k].y, q[k].z};
            unsigned int t[3];
#pragma unroll
            for (int a = 0; a < 3; a++) {
                // the cell alone: the rounding draw matters only when the dropped bits can carry into it (once in 65536)
                float y = (c[a] + offA) * ih[a];
                y = fminf(fmaxf(y, -163839.f), (float)(ncell[a] + 2) * 65536.f + 32000.f);      // as l3_S
                const float fl = floorf(y);
                int S = (int)fl + 32767;
                if ((S & 0xffff) == 0xffff && y > fl) {
                    float u[3];
                    ln_hash(c[0], c[1], c[2], u);
                    S += (y - fl) > u[a] ? 1 : 0;
                }
                t[a] = l3_tab[a * tabn + (S >> 16) + 2];
            }
            if ((t[0] >> 23) & 1u) continue;                   // not listed (outside the slab's windows)
            const int b0 = (int)(t[0] & 1023u) * bst[0] + (int)(t[1] & 1023u) * bst[1] + (int)(t[2] & 1023u);
            const int db[3] = {((int)((t[0] >> 10) & 2047u) - 1024) * bst[0], ((int)((t[1] >> 10) & 2047u) - 1024) * bst[1],
                               (int)((t[2] >> 10) & 2047u) - 1024};
            const unsigned int two = ((t[0] >> 21) & 1u) | (((t[1] >> 21) & 1u) << 1) | (((t[2] >> 21) & 1u) << 2);
            // tile entries of each block the cloud touches: a dimension with two blocks gives one tile to each
            const unsigned int nper = (((t[0] >> 21) & 1u) ? 1u : 1u + ((t[0] >> 22) & 1u)) * (((t[1] >> 21) & 1u) ? 1u : 1u + ((t[1] >> 22) & 1u)) *
                                      (((t[2] >> 21) & 1u) ? 1u : 1u + ((t[2] >> 22) & 1u));
            unsigned int e = 0u;                                  // every subset of the dimensions that have a second block
            do {
                const int b = b0 + ((e & 1u) ? db[0] : 0) + ((e & 2u) ? db[1] : 0) + ((e & 4u) ? db[2] : 0);
                atomicAdd(&hrec[b], 1u);
                atomicAdd(&hent[b], nper);
                e = (e - two) & two;
            } while (e);
        }
    }
    if (any_changed) *wrapped_flag = 1;
    __syncthreads();
    for (int b = tid; b < g.nbuckets; b += 512) {
        M[(int64_t)blockIdx.x * g.nbuckets + b] = hrec[b];
        if (hent[b]) atomicAdd(&ent[b], hent[b]);
    }
}

// ---- tables of a build, made on the device ------------------------------------------------------------------------
// What the host used to derive from the block totals between the counting and the scattering pass (a stream synchronise,
// a read-back, an upload: 0.15 ms of BASELINE config 3's step with the GPU idle): block starts of the staged records (gstart)
// and of the tile lists (fstart), both on line boundaries, and the pieces of at most PIECE records the fine level works on.
// The buffers are sized by the caller BEFORE the totals are known (from the previous build of the same mesh); need[] reports
// what was needed and need[3] != 0 says "did not fit": every later kernel of the build then does nothing, and the caller - who
// learns of it at its next synchronisation - runs the build again with exact sizes (tsc.hip: lines3_build, deferred mode).
__global__ __launch_bounds__(1024) void lines3_tables(const unsigned int *__restrict__ tot, const unsigned int *__restrict__ ent, const int *__restrict__ wflag,
                                                      int nb, int tpb, unsigned int PIECE, unsigned long long gs_cap, unsigned long long fs_cap, int np_cap,
                                                      unsigned int *__restrict__ gstart, unsigned int *__restrict__ fstart, int *__restrict__ piece_first,
                                                      LnPiece *__restrict__ pieces, unsigned int *__restrict__ need) {
    __shared__ unsigned long long sg[1024], sf[1024];
    __shared__ unsigned int sp[1024];
    __shared__ int bad;
    const int b = threadIdx.x;
    const unsigned int t = b < nb ? tot[b] : 0u, en = b < nb ? ent[b] : 0u;
    const unsigned long long g = b < nb ? ((unsigned long long)t + 15ull) & ~15ull : 0ull;
    const unsigned long long f = b < nb ? ((unsigned long long)en + 15ull * (unsigned long long)tpb + 15ull) & ~15ull : 0ull;
    const unsigned int npc = (t + PIECE - 1u) / PIECE;
    sg[b] = g, sf[b] = f, sp[b] = npc;
    if (b == 0) bad = 0;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {             // inclusive scans (a thousand values, once per build)
        const unsigned long long ag = b >= d ? sg[b - d] : 0ull, af = b >= d ? sf[b - d] : 0ull;
        const unsigned int ap = b >= d ? sp[b - d] : 0u;
        __syncthreads();
        sg[b] += ag, sf[b] += af, sp[b] += ap;
        __syncthreads();
    }
    const unsigned long long gs = sg[1023], fs = sf[1023];
    const unsigned int np = sp[1023];
    if (b == 0) {
        const int over = gs > gs_cap || fs > fs_cap || np > (unsigned int)np_cap || gs >= 0xfff00000ull || fs >= 0xfff00000ull;
        bad = over;
        need[0] = (unsigned int)min(gs, 0xffffffffull), need[1] = (unsigned int)min(fs, 0xffffffffull), need[2] = np, need[3] = (unsigned int)over;
        need[4] = (unsigned int)*wflag;
    }
    __syncthreads();
    const bool over = bad != 0;
    const unsigned int g0 = over ? 0u : (unsigned int)(sg[b] - g), f0 = over ? 0u : (unsigned int)(sf[b] - f);
    const int p0 = over ? 0 : (int)(sp[b] - npc);
    if (b < nb) gstart[b] = g0, fstart[b] = f0, piece_first[b] = p0;
    if (b == nb - 1) gstart[nb] = over ? 0u : (unsigned int)gs, fstart[nb] = over ? 0u : (unsigned int)fs, piece_first[nb] = over ? 0 : (int)np;
    if (!over && b < nb)
        for (unsigned int j = 0; j < npc; j++)
            pieces[p0 + (int)j] = LnPiece{b, g0 + j * PIECE, g0 + min((j + 1u) * PIECE, t), j == 0u ? 1 : 0};
    for (int p = (over ? 0 : (int)np) + b; p < np_cap; p += 1024) pieces[p] = LnPiece{0, 0u, 0u, 0};
}

// ---- coarse scatter: one record per (particle, block) ----------------------------------------------------------------
struct L3Item {
    unsigned int w[3];   // the record of the first block in every dimension
    unsigned int b0;     // its bucket; bit 31: no item
    int db[3];           // bucket delta of the second block of a dimension
    unsigned int two;    // bit a: dimension a has a second block
};
template <bool EXT>
__device__ __forceinline__ void l3_item(float x, float y, float z, float offset, const float ih[3], const LGeom &g, const int lgbc[3],
                                        const int bst[3], const L3Win &wn, L3Item &it) {
    const float c[3] = {x, y, z};
    const int sh[3] = {LN_SHX, LN_SHY, LN_SHZ};
    float u[3];
    ln_hash(x, y, z, u);
    it.b0 = 0u, it.two = 0u;
    bool listed = true;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        L3Dim d;
        int S = l3_S(c[a], offset, ih[a], (a == 0 && wn.on) ? wn.n : g.n[a], u[a]);
        if (a == 0) listed = l3_window<EXT>(wn, S);
        l3_dim<EXT>(S, g.n[a], lgbc[a], sh[a], d);
        const int B = l3_wrapB(d.Blo, g.nb[a]);
        it.b0 += (unsigned int)(B * bst[a]);
        it.db[a] = ((B + 1 == g.nb[a] ? 0 : B + 1) - B) * bst[a];
        it.two |= d.two ? 1u << a : 0u;
        it.w[a] = (unsigned int)(d.S + ((g.n[a] + 2 - (d.Blo << lgbc[a])) << 16));   // (cell in block + 2) << 16 | low bits of S
    }
    if (!listed) it.b0 = 0x80000000u, it.two = 0u;
}

template <int NB, int LINE, int SBUF, int NT, bool EXT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4))) void lines3_coarse(const float *__restrict__ pos, int64_t n, LGeom g, double box,
                                                                                           float offA, int64_t CH, const unsigned int *__restrict__ M,
                                                                                           const unsigned int *__restrict__ gstart,
                                                                                           uint4 *__restrict__ staged, unsigned long long *clk,
                                                                                           const unsigned int *__restrict__ need, L3Win wn) {
    __shared__ SplitLds<uint4, NB, LINE, SBUF, NT> s;
    const int tid = threadIdx.x, nb = g.nbuckets;
    if (need && need[3]) return;                          // the tables did not fit the buffers (lines3_tables): nothing is written
    split_init(s);
    for (int b = tid; b < NB; b += NT) s.base[b] = b < nb ? gstart[b] + M[(int64_t)blockIdx.x * nb + b] : 0u;
    __syncthreads();
    const float ih[3] = {(float)((wn.on ? wn.n : g.n[0]) / box) * 65536.f, (float)(g.n[1] / box) * 65536.f, (float)(g.n[2] / box) * 65536.f};
    const int lgbc[3] = {LN_SHX + g.sb[0], LN_SHY + g.sb[1], LN_SHZ + g.sb[2]};
    const int bst[3] = {g.nb[1] * g.nb[2], g.nb[2], 1};
    const int64_t p0 = (int64_t)blockIdx.x * CH, p1 = min(p0 + CH, n);
    // a round's output is about its input (1.05 - 1.1 records per particle); a round that does not fit (clouds piled up on
    // block corners: eight records per particle) is redone in G groups: a round's output is at most LINE times its new entries, so 8 x LINE x PMAX / G <= SBUF always fits
    constexpr int PMAX = SBUF * 3 / 4 / NT * NT;
    constexpr int PPT = PMAX / NT;
    constexpr int G = 64;
    static_assert(PPT >= 1 && 8 * (PMAX / G) * LINE <= SBUF && (G & (G - 1)) == 0, "buffer too small");
    int par = 0;
    LnF3 q[PPT];
    L3Item it[PPT];
    auto load = [&](int64_t s0) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const int64_t p = s0 + k * NT + tid;
            q[k] = LnF3{0.f, 0.f, 0.f};
            if (p < p1) q[k] = *reinterpret_cast<const LnF3 *>(pos + 3 * p);
        }
    };
    auto geometry = [&](int64_t s0) __attribute__((always_inline)) {          // (unconditional: an if / else over the items' fields left them in scratch memory)
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            l3_item<EXT>(q[k].x, q[k].y, q[k].z, offA, ih, g, lgbc, bst, wn, it[k]);
            const bool live = s0 + k * NT + tid < p1;
            it[k].b0 |= live ? 0u : 0x80000000u;
            it[k].two = live ? it[k].two : 0u;
        }
    };
    // pipeline: the particles of round r + 2 are requested behind the stores of round r; their geometry is evaluated in round
    // r + 1 BEFORE its write-out (split_round's `mid`: the items of round r + 1 are dead once they are placed), so the wait for
    // them never covers stores younger than a round
    load(p0);
    geometry(p0);
    load(p0 + PMAX);
    for (int64_t s0 = p0; s0 < p1; s0 += PMAX) {
        auto count = [&](int groups, int gi) {
            return [&, groups, gi](auto f) {
#pragma unroll
                for (int k = 0; k < PPT; k++) {
                    if ((it[k].b0 >> 31) || ((k * NT + tid) & (groups - 1)) != gi) continue;
                    unsigned int e = 0u;
                    do {
                        f((int)it[k].b0 + ((e & 1u) ? it[k].db[0] : 0) + ((e & 2u) ? it[k].db[1] : 0) + ((e & 4u) ? it[k].db[2] : 0));
                        e = (e - it[k].two) & it[k].two;
                    } while (e);
                }
            };
        };
        auto place = [&](int groups, int gi) {
            return [&, groups, gi](auto f) {
#pragma unroll
                for (int k = 0; k < PPT; k++) {
                    if ((it[k].b0 >> 31) || ((k * NT + tid) & (groups - 1)) != gi) continue;
                    unsigned int e = 0u;
                    do {   // the second block of a dimension sees the cell one block further down
                        f((int)it[k].b0 + ((e & 1u) ? it[k].db[0] : 0) + ((e & 2u) ? it[k].db[1] : 0) + ((e & 4u) ? it[k].db[2] : 0),
                          make_uint4(it[k].w[0] - ((e & 1u) ? 1u << (16 + lgbc[0]) : 0u), it[k].w[1] - ((e & 2u) ? 1u << (16 + lgbc[1]) : 0u),
                                     it[k].w[2] - ((e & 4u) ? 1u << (16 + lgbc[2]) : 0u), 0u));
                        e = (e - it[k].two) & it[k].two;
                    } while (e);
                }
            };
        };
        if (split_round<uint4, NB, LINE, SBUF, NT>(s, nb, par, 0, 0, staged, count(1, 0), place(1, 0), 0, clk, [&]() { geometry(s0 + PMAX); })) {
            par ^= 1;
        } else {
            for (int gi = 0; gi < G; gi++) {
                split_round<uint4, NB, LINE, SBUF, NT>(s, nb, par, 0, 0, staged, count(G, gi), place(G, gi));
                par ^= 1;
            }
            geometry(s0 + PMAX);
        }
        load(s0 + 2 * PMAX);
    }
    split_drain<uint4, NB, LINE, SBUF, NT>(s, nb, par, staged);
}

// ---- fine level: a record -> the tiles of its cloud inside the block ---------------------------------------------------
struct L3Rec {
    unsigned int key0;    // tile of the first tile in every dimension (inside the block)
    unsigned int h;       // bit a: dimension a has a second tile inside the block; bit 31: the cloud misses the block
    unsigned int lo, hi;  // the entry of the first tile
};
template <int ORG>
__device__ __forceinline__ void l3_decode(const uint4 &r, const LGeom &g, const int lgbc[3], L3Rec &o) {
    const unsigned int w[3] = {r.x + (ORG ? 0x8000u : 0u), r.y + (ORG ? 0x8000u : 0u), r.z + (ORG ? 0x8000u : 0u)};
    const int sh[3] = {LN_SHX, LN_SHY, LN_SHZ};
    const int ksh[3] = {g.sb[1] + g.sb[2], g.sb[2], 0};
    unsigned int key = 0u, h = 0u, l = 0u;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const int ir = (int)(w[a] >> 16) - 2, bc = 1 << lgbc[a];   // nearest cell relative to the block: [-2, bc + 1]
        const int lo = max(ir - 1, 0), hi = min(ir + 1, bc - 1);
        if (lo > hi) h |= 0x80000000u;                              // (only with EXT: the other origin's cloud alone reaches this block)
        const int ta = lo >> sh[a], tb = hi >> sh[a];
        key |= (unsigned int)ta << ksh[a];
        h |= ta != tb ? 1u << a : 0u;
        l |= (unsigned int)min(max(ir - (ta << sh[a]) + 1, 0), (1 << sh[a]) + 1) << (5 * a);   // nearest cell inside the first tile, biased by one
    }
    o.key0 = key, o.h = h;
    o.lo = l | ((~w[0] & 0xffffu) << 16);
    o.hi = (~w[1] & 0xffffu) | ((~w[2] & 0xffffu) << 16);
}
// tile key and entry of emission e (bit a: the second tile of dimension a)
__device__ __forceinline__ void l3_emit(const L3Rec &o, int e, const LGeom &g, unsigned int &key, unsigned long long &entry) {
    const unsigned int dk = ((e & 1) ? 1u << (g.sb[1] + g.sb[2]) : 0u) + ((e & 2) ? 1u << g.sb[2] : 0u) + ((e & 4) ? 1u : 0u);
    const unsigned int dl = ((e & 1) ? (unsigned int)LN_TX : 0u) + ((e & 2) ? (unsigned int)LN_TY << 5 : 0u) + ((e & 4) ? (unsigned int)LN_TZ << 10 : 0u);
    key = o.key0 + dk;
    entry = ((unsigned long long)o.hi << 32) | (o.lo - dl);
}

template <int NBF, int ORG>
__global__ __launch_bounds__(512) void lines3_fcount(const uint4 *__restrict__ staged, const LnPiece *__restrict__ pieces, LGeom g,
                                                     unsigned int *__restrict__ C) {
    __shared__ unsigned int hist[NBF];
    const int tid = threadIdx.x, tpb = g.tpb;
    const LnPiece pc = pieces[blockIdx.x];
    const int lgbc[3] = {LN_SHX + g.sb[0], LN_SHY + g.sb[1], LN_SHZ + g.sb[2]};
    for (int f = tid; f < NBF; f += 512) hist[f] = 0u;
    __syncthreads();
    constexpr int U = 4;
    for (unsigned int e0 = pc.e0; e0 < pc.e1; e0 += U * 512) {
        uint4 r[U];
#pragma unroll
        for (int k = 0; k < U; k++)
            if (e0 + k * 512 + tid < pc.e1) r[k] = staged[e0 + k * 512 + tid];
#pragma unroll
        for (int k = 0; k < U; k++) {
            if (e0 + k * 512 + tid >= pc.e1) continue;
            L3Rec o;
            l3_decode<ORG>(r[k], g, lgbc, o);
            if (o.h >> 31) continue;
            unsigned int e = 0u;                                  // every subset of the dimensions that have a second tile
            do {
                const unsigned int key = o.key0 + ((e & 1u) ? 1u << (g.sb[1] + g.sb[2]) : 0u) + ((e & 2u) ? 1u << g.sb[2] : 0u) + ((e & 4u) ? 1u : 0u);
                atomicAdd(&hist[min(key, (unsigned int)(NBF - 1))], 1u);
                e = (e - o.h) & o.h;
            } while (e);
        }
    }
    __syncthreads();
    for (int f = tid; f < tpb; f += 512) C[(int64_t)blockIdx.x * tpb + f] = hist[f];
}

template <int NBF, int LINE, int SBUF, int NT, int ORG>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4))) void lines3_fine(const uint4 *__restrict__ staged, const LnPiece *__restrict__ pieces, LGeom g,
                                                  const unsigned int *__restrict__ C, unsigned long long *__restrict__ entries,
                                                  unsigned long long *clk) {
    __shared__ SplitLds<unsigned long long, NBF, LINE, SBUF, NT> s;
    const int tid = threadIdx.x, nb = g.tpb;
    const LnPiece pc = pieces[blockIdx.x];
    const int lgbc[3] = {LN_SHX + g.sb[0], LN_SHY + g.sb[1], LN_SHZ + g.sb[2]};
    split_init(s);
    for (int f = tid; f < NBF; f += NT) s.base[f] = f < nb ? C[(int64_t)blockIdx.x * nb + f] : 0u;
    __syncthreads();
    constexpr int PMAX = SBUF * 5 / 8 / NT * NT;          // 1.3 entries out per record in
    constexpr int PPT = PMAX / NT;
    constexpr int G = 64;                                 // as in lines3_coarse: 8 x LINE x PMAX / G <= SBUF always fits
    static_assert(PPT >= 1 && 8 * (PMAX / G) * LINE <= SBUF && (G & (G - 1)) == 0, "buffer too small");
    int par = 0;
    uint4 nx[PPT];
    L3Rec q[PPT];
    auto load = [&](unsigned int s0) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const unsigned int e = s0 + k * NT + tid;
            nx[k] = make_uint4(0u, 0u, 0u, 0u);
            if (e < pc.e1) nx[k] = staged[e];
        }
    };
    if (pc.e0 >= pc.e1) return;
    auto decode = [&](unsigned int s0) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            l3_decode<ORG>(nx[k], g, lgbc, q[k]);
            q[k].h = s0 + k * NT + tid < pc.e1 ? q[k].h : 0x80000000u;
        }
    };
    // the same pipeline as lines3_coarse: records of round r + 2 requested behind the stores of round r, decoded in round r + 1
    // before its write-out
    load(pc.e0);
    decode(pc.e0);
    load(pc.e0 + PMAX);
    for (unsigned int s0 = pc.e0; s0 < pc.e1; s0 += PMAX) {
        auto count = [&](int groups, int gi) {
            return [&, groups, gi](auto f) {
#pragma unroll
                for (int k = 0; k < PPT; k++) {
                    if ((q[k].h >> 31) || ((k * NT + tid) & (groups - 1)) != gi) continue;
                    unsigned int e = 0u;
                    do {
                        unsigned int key;
                        unsigned long long en;
                        l3_emit(q[k], (int)e, g, key, en);
                        f((int)min(key, (unsigned int)(NBF - 1)));
                        e = (e - q[k].h) & q[k].h;
                    } while (e);
                }
            };
        };
        // (taking the ranks of a record's 1 .. 8 entries together - eight predicated slots per record, their LDS atomics in
        // flight at once - was measured: 1.83 ms against 1.12 for this per-lane loop at BASELINE config 3; a wave executes every
        // slot one of its lanes fills)
        auto place = [&](int groups, int gi) {
            return [&, groups, gi](auto f) {
#pragma unroll
                for (int k = 0; k < PPT; k++) {
                    if ((q[k].h >> 31) || ((k * NT + tid) & (groups - 1)) != gi) continue;
                    unsigned int e = 0u;
                    do {
                        unsigned int key;
                        unsigned long long en;
                        l3_emit(q[k], (int)e, g, key, en);
                        f((int)min(key, (unsigned int)(NBF - 1)), en);
                        e = (e - q[k].h) & q[k].h;
                    } while (e);
                }
            };
        };
        if (split_round<unsigned long long, NBF, LINE, SBUF, NT>(s, nb, par, 0, 0, entries, count(1, 0), place(1, 0), 0, clk, [&]() { decode(s0 + PMAX); })) {
            par ^= 1;
        } else {
            for (int gi = 0; gi < G; gi++) {
                split_round<unsigned long long, NBF, LINE, SBUF, NT>(s, nb, par, 0, 0, entries, count(G, gi), place(G, gi));
                par ^= 1;
            }
            decode(s0 + PMAX);
        }
        load(s0 + 2 * PMAX);
    }
    split_drain<unsigned long long, NBF, LINE, SBUF, NT>(s, nb, par, entries);
}
