// sb_thc_kernel.hip -- thermal heating contrast (the expanding-window land/sea mean
// difference of t0) on gfx950.   ref: generic/sea_breeze_diag.f90:188-216,
// python_wrapper/seabreezediag/seabreeze_diag_python.f90:187-221
//
// The reference re-sums a (2nn+1)^2 window from scratch at every radius nn until it holds
// both classes; only the last square matters.  Here a tile of 32 x TY cells plus a halo of
// H cells is held in LDS as three summed-area tables (all cells, land-side cells,
// land-side count), after which any square costs four LDS reads per table and the
// smallest valid radius is found by bisection (the "both classes present" predicate is
// monotone in nn).  Sums are taken about a per-tile offset c0, which cancels exactly in
// the difference of the two means and keeps the fp64 prefix sums small.
//
// One persistent 512-thread workgroup per CU (k_thc2, below): tiles that do not touch the
// coastal band (about 3 in 4) cost nothing, the others are dealt out from the compacted list
// of tile flags k_scan raised.
#include "sb_device.hpp"
#include "sb_launch.hpp"

#ifdef SB_STAMPS
#define SB_STAMP(i) \
    do { if (threadIdx.x == 0) job.stamps[(size_t)tile * SB_NSTAMP + (i)] = clock64(); } while (0)
#else
#define SB_STAMP(i) do { } while (0)
#endif

#define THC_MAXMINE 256          // active tiles one workgroup can own
// k_thc2 tiles are 32 longitudes wide: with the halo a staged row is exactly one 64-lane chunk (H = 16), so
// no lane of a staging load, an exp or an LDS write is padding.  48 latitudes by default: on the N1280 grid
// the coastal band touches 670 such tiles (575 of 64 rows, 1004 of 32) and tiles x staged rows is smallest
// there -- 47 us against 50 (64 rows) and 52 (32 rows).
#define THC2_TX 32
#define THC2_TY 48
#define THC2_TYL 64               // taller tiles, by sb_set_tile_rows only
#define THC2_TYS 32               // small grids (a band of a multi-GPU run, N512): half-height tiles, so that more
                                  // of the one-workgroup-per-CU grid has a tile and each tile is shorter
#define THC2_TY24 32              // tile rows with a halo of 24 (81 x 81 table entries)
#define THC2_TY32 16              // tile rows with a halo of 32: 81 x 97 table entries are what 160 KB of LDS hold
#ifndef THC2_NT
#define THC2_NT 512               // k_thc2: 8 waves per CU, so that a thread may hold 256 registers
#endif

// ------------------------------------------------------------------------------------
// Global-memory search for cells whose window outgrows the LDS tile (rare).  Rings are
// accumulated from the centre outwards; the first radius >= 1 at which the square holds
// both classes is the reference's final nn.  cap bounds the radius: the reference loop has
// none and never returns on a one-class grid (SURVEY.md §7 "Hard parts").
// ------------------------------------------------------------------------------------
// t0 of one cell of the ghost-celled frame: from the workspace (f2py flavour) or derived on the spot
template <typename T>
__device__ __forceinline__ T cell_t0(const DiagJob<T> &job, size_t idx, T sd, T rr) {
    if (!job.t0_fly) return job.t0[idx];
    return sb_t0<T>(job.theta[idx], job.z[idx], job.sigma[idx], sd, rr);
}

template <typename T>
__device__ T contrast_global(const DiagJob<T> &job, int x, int y, int cap, T sd, T rr, int &nn_used, bool &one_class) {
    const Geo g = job.g;
    int X, Y;
    bool has_l = false, has_s = false;
    if (sb_map_cell(g, x, y, X, Y)) {
        if (sb_bit(job.clsbits, g.nw, X, Y)) has_l = true; else has_s = true;
    }
    int nn = 0;
    bool found = false;
    while (nn < cap) {
        ++nn;
        for (int e = -nn; e <= nn; ++e) {
            const int xs[4] = {x + e, x + e, x - nn, x + nn};
            const int ys[4] = {y - nn, y + nn, y + e, y + e};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (sb_map_cell(g, xs[q], ys[q], X, Y)) {
                    if (sb_bit(job.clsbits, g.nw, X, Y)) has_l = true; else has_s = true;
                }
        }
        if (has_l && has_s) { found = true; break; }
    }
    nn_used = nn;
    one_class = !found;
    sb_map_cell(g, x, y, X, Y);
    const double c0 = (double)cell_t0(job, (size_t)Y * g.nxh + X, sd, rr);
    double sl = 0.0, ss = 0.0, nl = 0.0, ns = 0.0;
    for (int yy = y - nn; yy <= y + nn; ++yy)
        for (int xx = x - nn; xx <= x + nn; ++xx) {
            if (!sb_map_cell(g, xx, yy, X, Y)) continue;
            const double d = (double)cell_t0(job, (size_t)Y * g.nxh + X, sd, rr) - c0;
            if (sb_bit(job.clsbits, g.nw, X, Y)) { sl += d; nl += 1.0; } else { ss += d; ns += 1.0; }
        }
    return (T)(sl / nl - ss / ns);               // 0/0 -> NaN when a class is missing
}

// ------------------------------------------------------------------------------------
// The ordered list of active tiles, as every persistent 1024-thread workgroup builds it
// from the tile flags; returns how many entries this workgroup owns (s_mine[0..n)).
// Workgroups b and b+8 are observed to share an XCD (and its 4 MB L2); speed only, never
// correctness.  Each XCD therefore gets a contiguous eighth of the row-major list, dealt
// round-robin to its workgroups: tiles staged at the same time on one XCD are neighbours,
// and the halo cells they share are fetched from HBM once.
// ------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void thc_list_preload(const int *__restrict__ flags, int ntiles, int (&v)[SB_WAVE / (NT / SB_WAVE)]) {
    constexpr int K = SB_WAVE / (NT / SB_WAVE);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int t = k * NT + (int)threadIdx.x;
        v[k] = flags[t < ntiles ? t : ntiles - 1];
    }
}

// The position -> tile lookup of one chunk, by the first wave: lane e holds the count c and the inclusive
// prefix incl of entry e (tile order: k major, then wave); s_bm holds the entries' ballots.
template <int NT>
__device__ __forceinline__ void thc_list_lookup(int ch, int base, int c, int incl, int nmine, bool xcd_map, int per,
                                                int nper, const unsigned long long *s_bm, int *s_mine) {
    constexpr int NWV = NT / SB_WAVE, K = SB_WAVE / NWV, NE = K * NWV;
    const int lane = threadIdx.x & 63;
    const int xq = (int)blockIdx.x % 8, local = (int)blockIdx.x / 8;
    for (int j = 0; j < nmine; ++j) {
        const int pos = xcd_map ? xq * per + local + j * nper : (int)blockIdx.x + j * (int)gridDim.x;
        const int rel = pos - base;
        const uint64_t hit = __ballot(lane < NE && rel >= incl - c && rel < incl);
        if (hit) {                                           // wave-uniform: the position lies in this chunk
            const int e = __ffsll((unsigned long long)hit) - 1;
            int nth = rel - (__shfl(incl, e) - __shfl(c, e));   // which set bit of that entry's ballot
            const uint64_t bm = s_bm[e];
            int bit = 0;
#pragma unroll
            for (int w = 32; w > 0; w >>= 1) {               // select the nth set bit: halve the range six times
                const int cnt = __popcll((bm >> bit) & ((1ull << w) - 1ull));
                if (nth >= cnt) { nth -= cnt; bit += w; }
            }
            if (lane == 0) s_mine[j] = (ch * K + e / NWV) * NT + (e % NWV) * SB_WAVE + bit;
        }
    }
}

// Dynamic dealing (grids of at most one chunk = 4096 tiles): the ballots of the tile flags stay in LDS for
// the whole kernel and every wave keeps (count, inclusive prefix) of entry `lane`, so that any wave can turn
// a position of the row-major list of active tiles into a tile with thc_pos_to_tile.  Returns the number of
// active tiles.  Two barriers.
template <int NT>
__device__ __forceinline__ int thc_list_dynamic_init(const int *__restrict__ flags, int ntiles, int *s_wcnt,
                                                     unsigned long long *s_bm, const int *pre, int &c, int &incl) {
    constexpr int NWV = NT / SB_WAVE, K = SB_WAVE / NWV, NE = K * NWV;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int t = k * NT + tid;
        const int v = pre ? pre[k] : flags[t < ntiles ? t : ntiles - 1];
        const uint64_t bm = __ballot(t < ntiles && v != 0);
        if (lane == 0) { s_wcnt[k * NWV + wv] = __popcll(bm); s_bm[k * NWV + wv] = bm; }
    }
    __syncthreads();
    c = lane < NE ? s_wcnt[lane] : 0;
    incl = sb_wave_scan_add(c);
    __syncthreads();
    return __shfl(incl, SB_WAVE - 1);
}

// the tile at position pos (< number of active tiles) of the list; wave-uniform, every lane takes part
template <int NT>
__device__ __forceinline__ int thc_pos_to_tile(int pos, int c, int incl, const unsigned long long *s_bm) {
    constexpr int NWV = NT / SB_WAVE, K = SB_WAVE / NWV, NE = K * NWV;
    const int lane = threadIdx.x & 63;
    const uint64_t hit = __ballot(lane < NE && pos >= incl - c && pos < incl);
    const int e = hit ? __ffsll((unsigned long long)hit) - 1 : 0;
    int nth = pos - (__shfl(incl, e) - __shfl(c, e));        // which set bit of that entry's ballot
    const uint64_t bm = s_bm[e];
    int bit = 0;
#pragma unroll
    for (int w = 32; w > 0; w >>= 1) {                       // select the nth set bit: halve the range six times
        const int cnt = __popcll((bm >> bit) & ((1ull << w) - 1ull));
        if (nth >= cnt) { nth -= cnt; bit += w; }
    }
    return (e / NWV) * NT + (e % NWV) * SB_WAVE + bit;
}

// The ordered list of active tiles this workgroup owns -> s_mine[0..n).  s_wcnt: 64 ints, s_bm: 64
// ballot words.  A chunk is K * NT = 64 * 64 tiles whose flags are all loaded at once; `pre`: the flags of
// chunk 0 already loaded by thc_list_preload (issued early, so that other work hides the round trip), or
// nullptr.  A grid of at most one chunk (4096 tiles) needs two barriers and no second look at the flags.
template <int NT>
__device__ __forceinline__ int thc_build_list(const int *__restrict__ flags, int ntiles, int *s_mine, int *s_wcnt,
                                              unsigned long long *s_bm, const int *pre = nullptr) {
    constexpr int NWV = NT / SB_WAVE, K = SB_WAVE / NWV, NE = K * NWV;   // K flag loads in flight per thread
    static_assert(NE <= SB_WAVE, "the count table is prefixed by one wave");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nchunks = (ntiles + K * NT - 1) / (K * NT);
    const bool xcd_map = (gridDim.x % 8 == 0);
    const int nper = (int)gridDim.x / 8;
    const int xq = (int)blockIdx.x % 8, local = (int)blockIdx.x / 8;
    // ballots of one chunk -> LDS; returns after the barrier with (count, inclusive prefix) of entry `lane`
    auto chunk = [&](int ch, bool use_pre, int &c, int &incl) {
        int v[K];
        if (use_pre) {
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = pre[k];
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) {            // loads from clamped addresses, never under a branch:
                const int t = (ch * K + k) * NT + tid;   // a conditional load is waited for inside its branch
                v[k] = flags[t < ntiles ? t : ntiles - 1];
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint64_t bm = __ballot((ch * K + k) * NT + tid < ntiles && v[k] != 0);
            if (lane == 0) { s_wcnt[k * NWV + wv] = __popcll(bm); s_bm[k * NWV + wv] = bm; }
        }
        __syncthreads();
        c = lane < NE ? s_wcnt[lane] : 0;
        incl = sb_wave_scan_add(c);
    };
    auto owned = [&](int total_active, int per) {
        int n;
        if (xcd_map) {
            int cnt = total_active - xq * per;
            cnt = cnt < 0 ? 0 : (cnt > per ? per : cnt);
            n = cnt > local ? (cnt - 1 - local) / nper + 1 : 0;
        } else {
            n = total_active > (int)blockIdx.x ? (total_active - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
        }
        return n > THC_MAXMINE ? THC_MAXMINE : n;        // the launcher sizes the grid so this never binds
    };
    int c, incl;
    if (nchunks == 1) {
        chunk(0, pre != nullptr, c, incl);
        const int total_active = __shfl(incl, SB_WAVE - 1);
        const int per = (total_active + 7) / 8;
        const int nmine = owned(total_active, per);
        if (wv == 0) thc_list_lookup<NT>(0, 0, c, incl, nmine, xcd_map, per, nper, s_bm, s_mine);
        __syncthreads();
        return nmine;
    }
    int total_active = 0;
    for (int ch = 0; ch < nchunks; ++ch) {
        chunk(ch, ch == 0 && pre != nullptr, c, incl);
        total_active += __shfl(incl, SB_WAVE - 1);
        __syncthreads();
    }
    const int per = (total_active + 7) / 8;
    const int nmine = owned(total_active, per);
    int base = 0;
    for (int ch = 0; ch < nchunks; ++ch) {
        chunk(ch, false, c, incl);
        if (wv == 0) thc_list_lookup<NT>(ch, base, c, incl, nmine, xcd_map, per, nper, s_bm, s_mine);
        base += __shfl(incl, SB_WAVE - 1);
        __syncthreads();
    }
    return nmine;
}

// ====================================================================================
// k_thc2: the whole second half of a diag call in one persistent launch (halo H <= 16):
// sigmoid scalars (merge of k_scan's moments), t0 = theta - (gmma*z)*sigmoid(sigma) on the
// fly, the three summed-area tables, the radius search, thresholds and state update.
//
// Per tile the work is laid out so that nothing waits on HBM and no LDS pass is serial:
//   T0  band cells of the tile compacted into a list (full waves in the search however
//       ragged the band is); their state / wind loads are issued now
//   T1  the tile's inputs -- prefetched into registers while the PREVIOUS tile was being
//       processed -- become t0 - c0; a wave owns RPW consecutive rows, so the prefix along
//       latitude inside its band is RPW register adds per column, the land-side count is
//       prefixed along the row with ballot + popcount, and each wave leaves the column
//       totals of its band; then the NEXT tile's loads are issued
//   T2  exclusive prefix of the 16 band totals of every column (3*W short tasks)
//   T3  prefix along longitude: 8 threads per row, each a 12-cell run, run totals combined
//       by an 8-lane shuffle scan; the band offsets of T2 are added on the way in
//   T4  bisection on the count table, contrast from the two fp64 tables, thresholds,
//       state update (ref: generic/sea_breeze_diag.f90:188-216, :235-266)
// The barriers between the phases wait for LDS traffic only (lds_barrier), so the
// prefetched global loads stay in flight across them.
// ====================================================================================
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// the 64 bits of `bits` for the interior cells (x0 .. x0+63, y): at most two words of the plane
__device__ __forceinline__ uint64_t tile_row_bits(const uint64_t *__restrict__ bits, const Geo &g, int x0, int y) {
    const int X = x0 + g.h, Y = y + g.h;
    const int wi = X >> 6, sh = X & 63;
    const uint64_t *row = bits + (size_t)Y * g.nw;
    uint64_t w = row[wi] >> sh;
    if (sh && wi + 1 < g.nw) w |= row[wi + 1] << (64 - sh);
    return w;
}

// everything a thread holds of a tile between the issue of its loads and T1
template <typename T, int NC, int NCH, bool FLY>
struct ThcRegs {
    T th[NC];                      // theta (FLY) or t0
    T zz[FLY ? NC : 1], sg[FLY ? NC : 1];
    uint32_t lw[NC];               // the 32-bit half of the land-side word that holds the cell
    int xcol[NCH];                 // array column of each chunk, -1: no such cell
    unsigned okm;                  // bit k: cell k exists
    uint64_t bw0, bw1;             // the two words that hold the band bits of tile row tid (tid < TY)
    int bsh;                       // their shift, -1: no such row
    T c0;                          // the tile's offset
};

template <typename T, int TX, int TY, int H, bool FLY, int RPW, int NCH>
__device__ __forceinline__ void thc2_issue(const DiagJob<T> &job, int tile, ThcRegs<T, RPW * NCH, NCH, FLY> &R) {
    constexpr int W = TX + 2 * H;
    const Geo g = job.g;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ntx = job.thc_ntx;
    const int x0 = (tile % ntx) * TX, y0 = (tile / ntx) * TY;
    int X, Y;
    sb_map_cell(g, x0, y0, X, Y);
    const unsigned unxh = (unsigned)g.nxh;
    const unsigned i00 = (unsigned)Y * unxh + (unsigned)X;
    const bool fastx = g.nx > W + 2;             // one conditional add wraps every column of the tile
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c = ch * SB_WAVE + lane;
        const int xs = x0 - H + c;
        bool ok = c < W;
        int Xc = 0;
        if (g.bnd == BND_HALO) { Xc = xs + g.h; ok = ok && Xc >= 0 && Xc < g.nxh; }
        else if (fastx) {
            if (g.bnd == BND_WRAPPER) {
                int m = xs + 1;
                m = m < 0 ? m + g.nx : (m >= g.nx ? m - g.nx : m);
                Xc = (m < 1 ? 1 : m) - 1;
            } else Xc = xs < 0 ? xs + g.nx : (xs >= g.nx ? xs - g.nx : xs);
        } else {
            int Yd;
            sb_map_cell(g, xs, y0, Xc, Yd);
        }
        R.xcol[ch] = ok ? Xc : -1;
    }
    R.okm = 0;
#pragma unroll
    for (int ri = 0; ri < RPW; ++ri) {
        const int r = wv * RPW + ri;
        const int ys = y0 - H + r;
        int Yr;
        bool rowok = true;
        if (g.bnd == BND_HALO) { Yr = ys + g.h; rowok = Yr >= 0 && Yr < g.nyh; Yr = rowok ? Yr : 0; }
        else Yr = ys < 0 ? 0 : (ys >= g.ny ? g.ny - 1 : ys);
        const unsigned rowbase = (unsigned)Yr * unxh;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int k = ri * NCH + ch;
            const bool ok = rowok && R.xcol[ch] >= 0;
            const unsigned ii = ok ? rowbase + (unsigned)R.xcol[ch] : i00;     // i00: any cell that exists
            if constexpr (FLY) { R.th[k] = job.theta[ii]; R.zz[k] = job.z[ii]; }
            else R.th[k] = job.t0[ii];
            R.okm |= (ok ? 1u : 0u) << k;
        }
    }
    // Last, with no control flow behind them (a join would wait for them): the tile's offset -- any
    // common value conditions the sums, and theta at the tile origin needs no sigmoid -- and the band
    // words.  Every load of the tile is unconditional, from a clamped address: a load under a branch is
    // waited for inside the branch, one round trip after the other.
    if constexpr (FLY) R.c0 = job.theta[i00];
    else R.c0 = job.t0[i00];
    {
        const bool have = tid < TY && y0 + tid < g.rows;
        const int Xb = x0 + g.h, wi = Xb >> 6;
        const uint64_t *row = job.bandbits + (size_t)((have ? y0 + tid : y0) + g.h) * g.nw;
        R.bw0 = row[wi];
        R.bw1 = row[wi + 1 < g.nw ? wi + 1 : wi];
        R.bsh = have ? (Xb & 63) + (wi + 1 < g.nw ? 0 : 64) : -1;      // +64: there is no second word
    }
}

// The second, late part of a tile's loads: sigma and the land-side words.  Issued when the tile's
// turn starts (theta and z have been in flight since the previous tile's T1), so that the
// registers they land in are not live during the previous tile's search.
template <typename T, int TY, int H, bool FLY, int RPW, int NCH>
__device__ __forceinline__ void thc2_issue_late(const DiagJob<T> &job, int tile, ThcRegs<T, RPW * NCH, NCH, FLY> &R) {
    const Geo g = job.g;
    const int wv = threadIdx.x >> 6;
    const int y0 = (tile / job.thc_ntx) * TY;
    const uint32_t *cls32 = (const uint32_t *)job.clsbits;
#pragma unroll
    for (int ri = 0; ri < RPW; ++ri) {
        const int ys = y0 - H + wv * RPW + ri;
        int Yr;
        if (g.bnd == BND_HALO) { Yr = ys + g.h; Yr = (Yr >= 0 && Yr < g.nyh) ? Yr : 0; }
        else Yr = ys < 0 ? 0 : (ys >= g.ny ? g.ny - 1 : ys);
        const unsigned rowbase = (unsigned)Yr * (unsigned)g.nxh, wordbase = (unsigned)Yr * (unsigned)g.nw;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int k = ri * NCH + ch;
            const unsigned xc = ((R.okm >> k) & 1u) ? (unsigned)R.xcol[ch] : 0u;    // unconditional, clamped
            if constexpr (FLY) R.sg[k] = job.sigma[rowbase + xc];
            R.lw[k] = cls32[(size_t)(wordbase + (xc >> 6)) * 2 + ((xc >> 5) & 1u)];
        }
    }
}

template <typename T, int TX, int TY, int H, bool FLY, bool WF>     // WF: k_wind applies the update (job.wind_final)
__global__ __launch_bounds__(THC2_NT) void k_thc2(DiagJob<T> job, const Moments *__restrict__ partials, int nparts,
                                                  T *__restrict__ stats_out) {
    constexpr int NT = THC2_NT, NWV = NT / SB_WAVE;
    constexpr int W = TX + 2 * H, HT = TY + 2 * H, P = W + 1;
    constexpr int RPW = HT / NWV;                // consecutive table rows a wave owns
    constexpr int NCH = (W + SB_WAVE - 1) / SB_WAVE;
    constexpr int NC = RPW * NCH;
    constexpr int NSEG = 8, SEG = W / NSEG;      // the longitude prefix: 8 runs of SEG cells per row
    constexpr int CPT = (TX * TY + NT - 1) / NT; // list entries per thread in the search
    static_assert(HT % NWV == 0, "every wave owns the same number of rows");
    static_assert(W % NSEG == 0 && (2 * HT * NSEG) % SB_WAVE == 0, "row-pass task shape");
    static_assert(TX <= SB_WAVE && TY <= SB_WAVE && TY <= NT && NC <= 32, "tile shape");
    static_assert((size_t)W * HT < 65536, "u16 count table");
    __shared__ double sA[(HT + 1) * P];          // SAT of (t0 - c0), every cell
    __shared__ double sL[(HT + 1) * P];          // SAT of (t0 - c0), land-side cells
    __shared__ unsigned short sC[(HT + 1) * P];  // SAT of land-side count
    __shared__ double pA[NWV * W], pL[NWV * W];  // per-wave band totals of every column -> exclusive prefix
    __shared__ int pC[NWV * W];
    __shared__ int s_mine[THC_MAXMINE];
    __shared__ int s_wcnt[SB_WAVE];
    __shared__ uint64_t s_word[TY];
    __shared__ unsigned short s_cell[TX * TY];
    __shared__ Moments s_wpart[SB_STATS_NT / SB_WAVE];
    __shared__ T s_stats[4];
    __shared__ unsigned long long s_bmw[SB_WAVE];
    __shared__ unsigned short s_glob[TX * TY];   // cells whose window outgrows the tile (rare): handled after T4
    __shared__ int s_nglob, s_next;

    const Geo g = job.g;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ntx = job.thc_ntx, ntiles = job.thc_ntx * job.thc_nty;

#ifdef SB_STAMPS
    const long long t_begin = clock64(), w_begin = wall_clock64();
    long long t_pro[6] = {0, 0, 0, 0, 0, 0};
#define SB_PSTAMP(i) t_pro[i] = clock64() - t_begin
#else
#define SB_PSTAMP(i) do { } while (0)
#endif
    // k_scan's per-workgroup moments: loaded first, they land while the tile list is built
    constexpr int NPM = SB_STATS_NT / NT;
    Moments pm[NPM];
#pragma unroll
    for (int j = 0; j < NPM; ++j) {
        pm[j] = moments_empty();
        if constexpr (FLY) {
            if (nparts > 0) {              // wave-uniform; the load itself is unconditional (clamped)
                const int b = tid + j * NT;
                const Moments ld = partials[b < nparts ? b : nparts - 1];
                if (b < nparts) pm[j] = ld;
            }
        }
    }
    int preflags[SB_WAVE / NWV];
    thc_list_preload<NT>(job.tile_nnmax, ntiles, preflags);
    SB_PSTAMP(0);
    // sigmoid scalars: merged here from k_scan's per-workgroup moments, in the fixed order and
    // tree of k_moments_final, so every workgroup (and that kernel) gets the same bits.  The merge
    // runs while the tile flags are still on their way.
    T sd = T(0), rr = T(0);
    if constexpr (FLY) {
        if (nparts > 0) {
            // thread t stands for the threads t, t + NT, ... of that kernel's 1024; each of them is
            // reduced over its wave, the 16 wave results by the first wave
#pragma unroll
            for (int j = 0; j < NPM; ++j) {
                const int v = tid + j * NT;
                Moments mv = pm[j];
                if (j * NT < nparts) {         // wave-uniform; merging empties is the identity, so skip it
                    for (int b = v + SB_STATS_NT; b < nparts; b += SB_STATS_NT) mv = moments_merge(mv, partials[b]);
                    mv = wave_merge(mv);
                }
                if (lane == 0) s_wpart[v >> 6] = mv;
            }
            SB_PSTAMP(3);
            __syncthreads();
            SB_PSTAMP(4);
            Moments m = moments_empty();
            if (tid < SB_WAVE) {
                if (tid < SB_STATS_NT / SB_WAVE) m = s_wpart[tid];
                m = wave_merge(m);
            }
            if (tid == 0) {
                sigmoid_scalars<T>(m, s_stats);
                if (blockIdx.x == 0 && stats_out) sigmoid_scalars<T>(m, stats_out);
            }
        }
    }
    // Tiles are dealt dynamically when the flags fit one chunk (<= 4096 tiles): workgroup b starts with list
    // position b and draws further positions from a ticket (k_scan zeroes it) while it works, so that the
    // cost of a tile -- one to four search rounds, by its number of band cells -- evens out; the neighbours in
    // the list are then in flight on neighbouring workgroups at the same time.  Larger grids keep the static,
    // XCD-aware split of the list (s_mine).
    const bool dyn = ntiles <= (SB_WAVE / NWV) * NT && job.ticket != nullptr;
    int nmine = 0, nactive = 0, e_c = 0, e_incl = 0, tile = -1;
    if (dyn) {
        nactive = thc_list_dynamic_init<NT>(job.tile_nnmax, ntiles, s_wcnt, s_bmw, preflags, e_c, e_incl);
        if ((int)blockIdx.x < nactive) tile = thc_pos_to_tile<NT>((int)blockIdx.x, e_c, e_incl, s_bmw);
    } else {
        nmine = thc_build_list<NT>(job.tile_nnmax, ntiles, s_mine, s_wcnt, s_bmw, preflags);   // has barriers
        if (nmine > 0) tile = s_mine[0];
    }
    const int first_tile = tile;
    (void)first_tile;                          // used by the diagnostic build only
#ifdef SB_STAMPS
    const long long t_list = clock64();
#endif
    ThcRegs<T, NC, NCH, FLY> R;
    if (tile >= 0) thc2_issue<T, TX, TY, H, FLY, RPW, NCH>(job, tile, R);
    SB_PSTAMP(1);
    for (int i = tid; i < HT + 1; i += NT) { sA[i * P] = 0.0; sL[i * P] = 0.0; sC[i * P] = 0; }
    for (int i = tid; i < P; i += NT) { sA[i] = 0.0; sL[i] = 0.0; sC[i] = 0; }
    SB_PSTAMP(2);
    if constexpr (FLY) {
        if (nparts > 0) {
            sd = s_stats[0];                   // written before the list's barriers
            rr = s_stats[1];
        } else {
            sd = job.stats[0];
            rr = job.stats[1];
        }
    }
    __syncthreads();

#ifdef SB_STAMPS
    if (tid == 0 && first_tile >= 0) {       // prologue of this workgroup, kept with its first tile
        long long *st = job.stamps + (size_t)first_tile * SB_NSTAMP;
        st[8] = w_begin;                       // 100 MHz wall clock at this workgroup's first instruction
        st[9] = clock64() - t_begin;           // whole prologue, shader cycles
        st[10] = t_list - t_begin;             // tile list
        for (int i = 0; i < 6; ++i) st[16 + i] = t_pro[i];
    }
    if (lane == 0 && first_tile >= 0 && wv == NWV - 1)   // when did the last wave of the workgroup start?
        job.stamps[(size_t)first_tile * SB_NSTAMP + 12] = w_begin;
#endif
    for (int mi = 0; tile >= 0; ++mi) {
        const int x0 = (tile % ntx) * TX, y0 = (tile / ntx) * TY;
        // the position of the tile after this one: drawn now, it is back long before it is needed
        int next_pos = 0;
        if (dyn && tid == 0) next_pos = (int)gridDim.x + atomicAdd(job.ticket, 1);
        SB_STAMP(0);
        // ---- T0: compact the tile's band cells, issue their state loads ----------------------
        if (tid < TY) {
            uint64_t w = 0;
            if (R.bsh >= 0) {
                const int sh = R.bsh & 63;
                w = R.bw0 >> sh;
                if (sh && R.bsh < 64) w |= R.bw1 << (64 - sh);
                if (TX < 64) w &= (1ull << (TX & 63)) - 1ull;
            }
            s_word[tid] = w;
        }
        if (tid == 0) s_nglob = 0;
        SB_STAMP(24);
        thc2_issue_late<T, TY, H, FLY, RPW, NCH>(job, tile, R);     // sigma, land-side words: land under T0
        SB_STAMP(25);
        lds_barrier();
        SB_STAMP(26);
        int total;
        {
            // every wave prefixes the TY popcounts for itself; then lane (r, j) of a wave walks byte j of
            // the band word of one of the wave's rows -- at most 8 list entries per lane
            const int pc = lane < TY ? __popcll(s_word[lane]) : 0;
            const int incl = sb_wave_scan_add(pc);
            total = __shfl(incl, SB_WAVE - 1);
            const int excl = incl - pc;
            constexpr int NB = TX / 8;                           // bytes of a band word that belong to the tile
            constexpr int RPI = SB_WAVE / NB;                    // rows a wave covers per iteration
            for (int rb = wv * RPI; rb < TY; rb += NWV * RPI) {
                const int r = rb + lane / NB, j = lane % NB;
                const bool have = r < TY;
                const uint64_t w = s_word[have ? r : 0];
                int pos = __shfl(excl, have ? r : 0) + __popcll(w & ((1ull << (8 * j)) - 1ull));
                unsigned bits = have ? (unsigned)((w >> (8 * j)) & 0xffull) : 0u;
                while (bits) {
                    const int b = __ffs(bits) - 1;
                    s_cell[pos++] = (unsigned short)((r << 6) | (8 * j + b));
                    bits &= bits - 1;
                }
            }
        }
        lds_barrier();
        SB_STAMP(27);
        int cq[CPT];
        uint32_t ownw[CPT];            // the half-word of the land-side plane that holds the cell itself
        SbCellState<T> cst[CPT];
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const int i = tid + q * NT;
            cq[q] = i < total ? (int)s_cell[i] : -1;
            cst[q] = SbCellState<T>{T(0), T(0), T(0), T(0)};
            {   // unconditional loads; a thread without a cell reads the tile's first cell
                const int cc = cq[q] >= 0 ? cq[q] : 0;
                const int x = x0 + (cc & 63), y = y0 + (cc >> 6);
                if constexpr (!WF) cst[q] = sb_trigger_load<T>(job, (size_t)y * g.nx + x);
                const unsigned X = (unsigned)(x + g.h);
                ownw[q] = ((const uint32_t *)job.clsbits)[((size_t)(y + g.h) * g.nw + (X >> 6)) * 2 + ((X >> 5) & 1u)];
            }
        }
        SB_STAMP(1);
        // ---- T1: registers -> band-local column prefix -> LDS; band totals ---------------------
        {
            const double c0 = (double)R.c0;
            double runA[NCH], runL[NCH];
            int runC[NCH];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) { runA[ch] = 0.0; runL[ch] = 0.0; runC[ch] = 0; }
#pragma unroll
            for (int ri = 0; ri < RPW; ++ri) {
                const int r = wv * RPW + ri;
                unsigned carryC = 0;
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    const int k = ri * NCH + ch;
                    const int c = ch * SB_WAVE + lane;
                    const bool ok = (R.okm >> k) & 1u;
                    const int land = ok ? (int)((R.lw[k] >> (R.xcol[ch] & 31)) & 1u) : 0;
                    T t0v = R.th[k];
                    if constexpr (FLY) t0v = sb_t0<T>(R.th[k], R.zz[k], R.sg[k], sd, rr);   // ref :166-167
                    const double d = ok ? (double)t0v - c0 : 0.0;
                    const uint64_t lm = __ballot(land);
                    const unsigned cn = carryC + (unsigned)__popcll(lm & (~0ull >> (63 - lane)));
                    carryC += (unsigned)__popcll(lm);
                    runA[ch] += d;
                    runL[ch] += land ? d : 0.0;
                    runC[ch] += (int)cn;
                    if (c < W) {
                        const int o = (r + 1) * P + c + 1;
                        sA[o] = runA[ch];
                        sL[o] = runL[ch];
                        sC[o] = (unsigned short)runC[ch];
                    }
                }
            }
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int c = ch * SB_WAVE + lane;
                if (c < W) { pA[wv * W + c] = runA[ch]; pL[wv * W + c] = runL[ch]; pC[wv * W + c] = runC[ch]; }
            }
        }
        // the cells' own class, out of the words loaded in T0: resolved here so that nothing loaded
        // before the prefetch is first used after it (that use would wait for the whole prefetch)
        unsigned ownbits = 0;
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const int cc = cq[q] >= 0 ? cq[q] : 0;
            ownbits |= ((ownw[q] >> ((x0 + (cc & 63) + g.h) & 31)) & 1u) << q;
        }
        asm volatile("" : "+v"(ownbits));        // materialise here: the compiler would sink this into T4
        if (dyn && tid == 0) s_next = next_pos;
        lds_barrier();
        // the next tile's loads fly under T2..T4 (and under the other workgroups' staging)
        int next_tile = -1;
        if (dyn) {
            const int np = s_next;                               // read by everyone before the next write (4 barriers on)
            if (np < nactive) next_tile = thc_pos_to_tile<NT>(np, e_c, e_incl, s_bmw);
        } else if (mi + 1 < nmine) next_tile = s_mine[mi + 1];
        if (next_tile >= 0) thc2_issue<T, TX, TY, H, FLY, RPW, NCH>(job, next_tile, R);
        SB_STAMP(2);
        // ---- T2: exclusive prefix of the band totals along latitude ------------------------------
        for (int t = tid; t < 3 * W; t += NT) {
            const int a = t / W, c = t - a * W;
            if (a < 2) {
                double *pp = (a == 0 ? pA : pL) + c;
                double x[NWV];
#pragma unroll
                for (int b = 0; b < NWV; ++b) x[b] = pp[b * W];
                double e = 0.0;
#pragma unroll
                for (int b = 0; b < NWV; ++b) { pp[b * W] = e; e += x[b]; }
            } else {
                int *pp = pC + c;
                int x[NWV];
#pragma unroll
                for (int b = 0; b < NWV; ++b) x[b] = pp[b * W];
                int e = 0;
#pragma unroll
                for (int b = 0; b < NWV; ++b) { pp[b * W] = e; e += x[b]; }
            }
        }
        lds_barrier();
        SB_STAMP(3);
        // ---- T3: prefix along longitude (+ band offsets); count table: band offsets only ---------
        // 16 consecutive lanes take 16 consecutive rows of one run (the pitch is odd, so their 8-byte
        // elements fall in 16 different bank pairs); the four 16-lane groups of a wave and the two
        // tasks of a thread cover the 8 runs of those rows.  The two groups of a 32-lane read are given
        // runs whose starts are 16 elements apart modulo 32 (4 runs apart for 12-cell runs, 2 for
        // 8-cell runs), so that they do not share banks either.
        {
            static_assert((2 * HT) % 16 == 0 && NSEG == 8, "row-pass lane mapping");
            constexpr int NRG = 2 * HT / 16;         // 16-row groups over both fp64 tables
            constexpr int NRND = (NRG + NWV - 1) / NWV, SPARE = NRND * NWV - NRG;   // idle wave slots of the last round
            constexpr int DSEG = (SEG % 8 == 4) ? 4 : (SEG % 16 == 8) ? 2 : 1;
            constexpr int CT = HT * NSEG;            // count-table tasks (one run each)
            const int grp = lane >> 4;
            // runs of group q: first task, second task
            auto run0 = [](int q) { return DSEG == 4 ? (q >> 1) + ((q & 1) << 2) : DSEG == 2 ? (q >> 1) + ((q & 1) << 1) : q; };
            auto run1 = [&](int q) { return run0(q) + (DSEG == 4 ? 2 : 4); };
            auto count_tasks = [&](int first, int step) {
                for (int task = first; task < CT; task += step) {
                    const int row = (task & 15) + 16 * (task / (16 * NSEG)), seg = (task >> 4) & (NSEG - 1);
                    unsigned short *tab = sC + (row + 1) * P + 1 + seg * SEG;
                    const int *off = pC + (row / RPW) * W + seg * SEG;
#pragma unroll
                    for (int i = 0; i < SEG; ++i) tab[i] = (unsigned short)((int)tab[i] + off[i]);
                }
            };
            for (int rg = wv; rg < NRND * NWV; rg += NWV) {
                if (rg >= NRG) {                                 // wave-uniform: a spare slot takes count-table tasks
                    if (SPARE > 0) count_tasks((rg - NRG) * SB_WAVE + lane, SPARE * SB_WAVE);
                    continue;
                }
                const int trow = rg * 16 + (lane & 15);          // row over both tables
                const int a = trow / HT, row = trow - a * HT;
                double *trw = (a == 0 ? sA : sL) + (row + 1) * P + 1;
                const double *orw = (a == 0 ? pA : pL) + (row / RPW) * W;
                const int sg0 = run0(grp), sg1 = run1(grp);
                double v0[SEG], v1[SEG], s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int i = 0; i < SEG; ++i) {
                    v0[i] = trw[sg0 * SEG + i] + orw[sg0 * SEG + i];
                    v1[i] = trw[sg1 * SEG + i] + orw[sg1 * SEG + i];
                }
#pragma unroll
                for (int i = 0; i < SEG; ++i) { s0 += v0[i]; v0[i] = s0; s1 += v1[i]; v1[i] = s1; }
                // totals of the 8 runs of this row, in run order, then the exclusive offsets
                double tot[NSEG];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int src = (lane & 15) + 16 * q;
                    tot[run0(q)] = __shfl(s0, src);
                    tot[run1(q)] = __shfl(s1, src);
                }
                double e0 = 0.0, e1 = 0.0, acc = 0.0;
#pragma unroll
                for (int r = 0; r < NSEG; ++r) {
                    if (r == sg0) e0 = acc;
                    if (r == sg1) e1 = acc;
                    acc += tot[r];
                }
#pragma unroll
                for (int i = 0; i < SEG; ++i) {
                    trw[sg0 * SEG + i] = v0[i] + e0;
                    trw[sg1 * SEG + i] = v1[i] + e1;
                }
            }
            if (SPARE == 0) count_tasks(tid, NT);
        }
        lds_barrier();
        SB_STAMP(4);
        // ---- T4: smallest radius whose square holds both classes (bisection), contrast, update ---
        // Branch-free up to the final store: a thread without a cell probes around the tile's first
        // cell and discards the result (divergent control flow costs more in exec-mask bookkeeping
        // than the probes it would skip).
        int nnmax = 0;
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const bool valid = cq[q] >= 0;
            if (__ballot(valid) == 0) continue;                  // wave-uniform: no lane of this wave has a q-th cell
            const int cc = valid ? cq[q] : 0;
            const int lx = cc & 63, ly = cc >> 6;
            const int x = x0 + lx, y = y0 + ly;
            const int cx = lx + H, cy = ly + H;
            int lim = H;
            if (g.bnd == BND_HALO)                               // wave-uniform
                lim = min(lim, min(min(x + g.h, g.nx - 1 - x + g.h), min(y + g.h, g.ny - 1 - y + g.h)));
            const int limc = max(lim, 1);
            // land-side cells in the square of radius rad around the cell
            auto count = [&](int rad) {
                const int r0 = (cy - rad) * P, r1 = (cy + rad + 1) * P;
                const int a0 = cx - rad, a1 = cx + rad + 1;
                return (int)sC[r1 + a1] - (int)sC[r0 + a1] - (int)sC[r1 + a0] + (int)sC[r0 + a0];
            };
            int nlq = count(limc);
            const bool fnd = valid && lim >= 1 && nlq > 0 && nlq < (2 * limc + 1) * (2 * limc + 1);
            int lo = fnd ? 1 : limc, hi = limc;                  // nothing to bisect unless the widest square is mixed
            constexpr int ITER = (H <= 2 ? 1 : H <= 4 ? 2 : H <= 8 ? 3 : H <= 16 ? 4 : 5);
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int mid = (lo + hi) >> 1;
                const int nl = count(mid);
                const bool act = lo < hi, ok = nl > 0 && nl < (2 * mid + 1) * (2 * mid + 1);
                nlq = (act && ok) ? nl : nlq;
                hi = (act && ok) ? mid : hi;
                lo = (act && !ok) ? mid + 1 : lo;
            }
            const int nn = hi;
            const int r0 = (cy - nn) * P, r1 = (cy + nn + 1) * P;
            const int a0 = cx - nn, a1 = cx + nn + 1;
            const int area = (2 * nn + 1) * (2 * nn + 1);
            const double RL = (sL[r1 + a1] - sL[r0 + a1]) - (sL[r1 + a0] - sL[r0 + a0]);
            const double RA = (sA[r1 + a1] - sA[r0 + a1]) - (sA[r1 + a0] - sA[r0 + a0]);
            const T contrast = (T)(RL / (double)nlq - (RA - RL) / (double)(area - nlq));
            // the cell's own class (not the table's: the f2py boundary rule maps the centre of the
            // window at the last longitude to column 1)   ref :182-186, seabreeze_diag_python.f90:202
            const T mul = ((ownbits >> q) & 1u) ? T(1) : T(-1);
            if (fnd) {
                nnmax = max(nnmax, nn);
                if constexpr (WF) job.thc[(size_t)y * g.nx + x] = mul * contrast;          // ref :216; k_wind applies :235-266
                else sb_trigger_update<T>(job, (size_t)y * g.nx + x, mul * contrast, cst[q]);   // ref :216, :235-266
            } else if (valid) {                  // the window outgrows the tile: queue the cell
                s_glob[atomicAdd(&s_nglob, 1)] = (unsigned short)cc;
            }
        }
        // ---- cells on the global-memory path (none on a grid whose halo hint holds) ---------------
        lds_barrier();
        {
            const int nglob = s_nglob;
#pragma unroll 1
            for (int i = tid; i < nglob; i += NT) {
                const int cc = s_glob[i];
                const int x = x0 + (cc & 63), y = y0 + (cc >> 6);
                const size_t o = (size_t)y * g.nx + x;
                int cap = g.nx + g.ny;
                if (g.bnd == BND_HALO)
                    cap = min(min(x + g.h, g.nx - 1 - x + g.h), min(y + g.h, g.ny - 1 - y + g.h));
                bool one_class;
                int nn;
                const T contrast = contrast_global(job, x, y, cap, sd, rr, nn, one_class);
                atomicAdd(&job.counters[0], 1);
                if (one_class) atomicAdd(&job.counters[1], 1);
                nnmax = max(nnmax, nn);
                const T mul = sb_bit(job.clsbits, g.nw, x + g.h, y + g.h) ? T(1) : T(-1);
                if constexpr (WF) job.thc[o] = mul * contrast;
                else sb_trigger_update<T>(job, o, mul * contrast, sb_trigger_load<T>(job, o));
            }
        }
        // per-tile largest radius (diagnostic; reduced lazily by sb_last_counters); the flag k_scan
        // raised is 1, and a nonzero flag stays nonzero for workgroups still building their list
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nnmax = max(nnmax, __shfl_xor(nnmax, off));
        if (lane == 0 && nnmax > 1) atomicMax(&job.tile_nnmax[tile], nnmax);
        SB_STAMP(5);
        // no barrier here: the next tile's T0 touches only s_word / s_cell / s_nglob (all read before
        // the barrier above), and its two barriers stand between this search and the next table write
        tile = next_tile;
    }
#ifdef SB_STAMPS
    if (tid == 0 && first_tile >= 0) job.stamps[(size_t)first_tile * SB_NSTAMP + 11] = wall_clock64();
#endif
}

template <typename T, int TX, int TY, int H>
static void launch_thc2(const DiagJob<T> &job, int nblocks, const Moments *partials, int nparts, T *stats_out,
                        hipStream_t st) {
    const dim3 gr(nblocks), bl(THC2_NT);
    if (job.t0_fly && job.wind_final) hipLaunchKernelGGL((k_thc2<T, TX, TY, H, true, true>), gr, bl, 0, st, job, partials, nparts, stats_out);
    else if (job.t0_fly) hipLaunchKernelGGL((k_thc2<T, TX, TY, H, true, false>), gr, bl, 0, st, job, partials, nparts, stats_out);
    else if (job.wind_final) hipLaunchKernelGGL((k_thc2<T, TX, TY, H, false, true>), gr, bl, 0, st, job, partials, nparts, stats_out);
    else hipLaunchKernelGGL((k_thc2<T, TX, TY, H, false, false>), gr, bl, 0, st, job, partials, nparts, stats_out);
}

template <typename T>
hipError_t sb_launch_thc2(const DiagJob<T> &job, int H, int ncu, const Moments *partials, int nparts, T *stats_out,
                          hipStream_t st) {
    const int ntiles = job.thc_ntx * job.thc_nty;
    int nblocks = ncu;
    while ((ntiles + nblocks - 1) / nblocks + 8 > THC_MAXMINE) nblocks *= 2;
    if (H <= 8) launch_thc2<T, THC2_TX, THC2_TY, 8>(job, nblocks, partials, nparts, stats_out, st);
    else if (H <= 16 && job.thc_ty == THC2_TYS) launch_thc2<T, THC2_TX, THC2_TYS, 16>(job, nblocks, partials, nparts, stats_out, st);
    else if (H <= 16 && job.thc_ty == THC2_TYL) launch_thc2<T, THC2_TX, THC2_TYL, 16>(job, nblocks, partials, nparts, stats_out, st);
    else if (H <= 16) launch_thc2<T, THC2_TX, THC2_TY, 16>(job, nblocks, partials, nparts, stats_out, st);
    else if (H <= 24) launch_thc2<T, THC2_TX, THC2_TY24, 24>(job, nblocks, partials, nparts, stats_out, st);
    else launch_thc2<T, THC2_TX, THC2_TY32, 32>(job, nblocks, partials, nparts, stats_out, st);   // H == 32
    return hipGetLastError();
}
template hipError_t sb_launch_thc2<float>(const DiagJob<float> &, int, int, const Moments *, int, float *, hipStream_t);
template hipError_t sb_launch_thc2<double>(const DiagJob<double> &, int, int, const Moments *, int, double *, hipStream_t);

void sb_thc_tile_shape(int H, int nx, int rows, int ncu, int *tx, int *ty) {
    if (H <= 16) {
        *tx = THC2_TX;
        *ty = THC2_TY;
        // about a quarter of the tiles touch the coastal band: while even twice the tile count would leave
        // workgroups without a tile, use the 32-row tiles (H = 16 only)
        const long long full = (long long)((nx + THC2_TX - 1) / THC2_TX) * ((rows + THC2_TYL - 1) / THC2_TYL);
        if (H > 8 && full <= 2LL * ncu) *ty = THC2_TYS;
    }
    else if (H <= 24) { *tx = THC2_TX; *ty = THC2_TY24; }
    else { *tx = THC2_TX; *ty = THC2_TY32; }
}

