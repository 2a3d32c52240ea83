// sb_thc_kernel.hip -- thermal heating contrast (the expanding-window land/sea mean
// difference of t0) on gfx950.   ref: generic/sea_breeze_diag.f90:188-216,
// python_wrapper/seabreezediag/seabreeze_diag_python.f90:187-221
//
// The reference re-sums a (2nn+1)^2 window from scratch at every radius nn until it holds
// both classes; only the last square matters.  Here a tile of 32 x TY cells plus a halo of
// H cells is held in LDS as three summed-area tables (all cells, land-side cells,
// land-side count), after which any square costs four LDS reads per table and the
// smallest valid radius is found by a two-round search (the "both classes present"
// predicate is monotone in nn).  Sums are taken about a per-tile offset c0, which cancels
// exactly in the difference of the two means and keeps the fp64 prefix sums small.
//
// One persistent workgroup per CU (k_thc3, below) works through the list of active tiles
// k_prep compacted from the flags k_scan raised: tiles that do not touch the coastal band
// (about 3 in 4) cost nothing.
#include "sb_device.hpp"
#include "sb_launch.hpp"

// k_thc3 tiles are 32 longitudes wide: with the halo a staged row is exactly one 64-lane chunk (H = 16), so
// no lane of a staging load, an exp or an LDS write is padding.  48 latitudes by default: on the N1280 grid
// the coastal band touches 670 such tiles (575 of 64 rows, 1004 of 32) and tiles x staged rows is smallest there.
#define THC_TX 32
#define THC_TY 48
#define THC_TYL 64                // taller tiles, by sb_set_tile_rows only
#define THC_TYS 32                // small grids (a band of a multi-GPU run, N512): half-height tiles, so that more
                                  // of the one-workgroup-per-CU grid has a tile and each tile is shorter
#define THC_TY24 32               // tile rows with a halo of 24 (81 x 81 table entries)
#define THC_TY32 16               // tile rows with a halo of 32: 81 x 97 table entries are what 160 KB of LDS hold

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
__device__ __forceinline__ T contrast_global(const DiagJob<T> &job, int x, int y, int cap, T sd, T rr, int &nn_used,
                                          bool &one_class) {
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

// ====================================================================================
// k_thc3: t0 = theta - (gmma*z)*sigmoid(sigma) on the fly, the three summed-area tables, the
// radius search and the contrast (plus thresholds and state update in a band step).
//
// A wave owns RPW consecutive rows of the staged tile and a lane one column of them, so
//   * the prefix along latitude inside the wave's band of rows is register adds,
//   * the prefix along longitude is a wave scan (DPP; the count comes from ballot + popcount),
//   * what one band needs from the bands above it is one row of column totals per table.
// Per tile:
//   A1  the tile's inputs -- prefetched into registers while the PREVIOUS tile was processed --
//       become t0 - c0 (kept in registers) and the band totals; the NEXT tile's loads are issued
//   A2  [barrier] band cells compacted into a list; totals of the bands above added; second pass
//       over the registers writes the finished tables (one LDS store per entry, nothing read back)
//   A3  [barrier] two-round search on the count table (radii H/4, H/2, 3H/4, H, then the radii
//       inside the bracket), contrast from the fp64 tables, result
// The barriers wait for LDS traffic only, so the prefetched global loads stay in flight across them.
// ====================================================================================
#ifdef SB_STAMPS
#define SB_T(i) do { const long long t_now = clock64(); acc[i] += t_now - t_last; t_last = t_now; } while (0)
#else
#define SB_T(i) do { } while (0)
#endif

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// everything a thread holds of a tile between the issue of its loads and A1
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

template <typename T, int TX, int TY, int H, int NT, bool FLY, int RPW, int NCH>
__device__ __forceinline__ void thc_issue(const DiagJob<T> &job, int tile, ThcRegs<T, RPW * NCH, NCH, FLY> &R) {
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
    const uint32_t *cls32 = (const uint32_t *)job.clsbits;
    // Every load of the tile is unconditional, from a clamped address: a load under a branch is
    // waited for inside the branch, one round trip after the other.
#pragma unroll
    for (int ri = 0; ri < RPW; ++ri) {
        const int r = wv * RPW + ri;
        const int ys = y0 - H + r;
        int Yr;
        bool rowok = true;
        if (g.bnd == BND_HALO) { Yr = ys + g.h; rowok = Yr >= 0 && Yr < g.nyh; Yr = rowok ? Yr : 0; }
        else Yr = ys < 0 ? 0 : (ys >= g.ny ? g.ny - 1 : ys);
        const unsigned rowbase = (unsigned)Yr * unxh, wordbase = (unsigned)Yr * (unsigned)g.nw;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int k = ri * NCH + ch;
            const bool ok = rowok && R.xcol[ch] >= 0;
            const unsigned xc = ok ? (unsigned)R.xcol[ch] : 0u;
            const unsigned ii = ok ? rowbase + xc : i00;                       // i00: any cell that exists
            if constexpr (FLY) { R.th[k] = job.theta[ii]; R.zz[k] = job.z[ii]; R.sg[k] = job.sigma[ii]; }
            else R.th[k] = job.t0[ii];
            R.lw[k] = cls32[(size_t)(wordbase + (xc >> 6)) * 2 + ((xc >> 5) & 1u)];
            R.okm |= (ok ? 1u : 0u) << k;
        }
    }
    // the tile's offset -- any common value conditions the sums, and theta at the tile origin needs
    // no sigmoid -- and the band words
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

template <typename T, int TX, int TY, int H, int NT, bool FLY, bool WF>     // WF: k_wind applies the update (job.wind_final)
__global__ __launch_bounds__(NT) void k_thc3(DiagJob<T> job) {
    constexpr int NWV = NT / SB_WAVE;
    constexpr int W = TX + 2 * H, HT = TY + 2 * H, P = W + 1;
    constexpr int RPW = HT / NWV;                // consecutive table rows a wave owns
    constexpr int NCH = (W + SB_WAVE - 1) / SB_WAVE;
    constexpr int NC = RPW * NCH;
    constexpr int CPT = (TX * TY + NT - 1) / NT; // list entries per thread in the search
    constexpr int STEP = H / 4;                  // first-round radii: STEP, 2 STEP, 3 STEP, H
    static_assert(HT % NWV == 0, "every wave owns the same number of rows");
    static_assert(H % 4 == 0 && TX <= SB_WAVE && TY <= SB_WAVE && TY <= NT && NC <= 32, "tile shape");
    static_assert((size_t)W * HT < 65536, "u16 count table");
    // The next tile's inputs are prefetched into registers a whole tile ahead -- except where a thread holds two
    // chunks of every row in double precision (halos of 24 and 32 cells): there the prefetch registers would
    // spill, and the loads are issued when the tile's turn comes.
    constexpr bool PF = NCH == 1;
    __shared__ double sA[(HT + 1) * P];          // SAT of (t0 - c0), every cell
    __shared__ double sL[(HT + 1) * P];          // SAT of (t0 - c0), land-side cells
    __shared__ unsigned short sC[(HT + 1) * P];  // SAT of land-side count
    __shared__ double pA[NWV * W], pL[NWV * W];  // per-wave band totals of every column, prefixed along longitude
    __shared__ int pC[NWV * W];
    __shared__ uint64_t s_land[2][HT * NCH];     // land-side bits of every staged row; two buffers: the search of tile i reads
                                                 // them while a wave that is ahead already stages tile i + 1
    __shared__ uint64_t s_word[TY];              // band bits of every tile row
    __shared__ unsigned short s_cell[TX * TY];   // the tile's band cells, compacted

    const Geo g = job.g;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ntx = job.thc_ntx;
    const uint64_t le_mask = ~0ull >> (63 - lane);               // lanes 0 .. lane

#ifdef SB_STAMPS
    long long acc[SB_NSTAMP], t_last = clock64();
    const long long w_begin = wall_clock64();
    for (int i = 0; i < SB_NSTAMP; ++i) acc[i] = 0;
#endif
    // ---- prologue: the list of active tiles and the sigmoid scalars are k_prep's work -------------
    const int nactive = job.tile_list[0];
    const int G = (int)gridDim.x;
    int pos = (int)blockIdx.x;
    int tile = pos < nactive ? job.tile_list[1 + pos] : -1;
    ThcRegs<T, NC, NCH, FLY> R;
    if (PF && tile >= 0) thc_issue<T, TX, TY, H, NT, FLY, RPW, NCH>(job, tile, R);
    // the position after this one: its list entry is loaded a whole tile before it is needed
    int next_tile = pos + G < nactive ? job.tile_list[1 + pos + G] : -1;
    T sd = T(0), rr = T(0);
    if constexpr (FLY) { sd = job.stats[0]; rr = job.stats[1]; }
    for (int i = tid; i < HT + 1; i += NT) { sA[i * P] = 0.0; sL[i * P] = 0.0; sC[i * P] = 0; }
    for (int i = tid; i < P; i += NT) { sA[i] = 0.0; sL[i] = 0.0; sC[i] = 0; }
    __syncthreads();
    SB_T(0);                                   // prologue

    int par = 0;
    while (tile >= 0) {
        const int x0 = (tile % ntx) * TX, y0 = (tile / ntx) * TY;
        if (!PF) thc_issue<T, TX, TY, H, NT, FLY, RPW, NCH>(job, tile, R);
        // ---- A1: registers -> t0 - c0, band totals -------------------------------------------------
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
#ifdef SB_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the tile's loads have landed (diagnostic build only)
        SB_T(1);                               // waiting for the prefetch
        acc[8] += 1;                           // tiles
#endif
        double d[NC];                            // t0 - c0 of the thread's cells
        unsigned landbits = 0;                   // bit k: cell k is on the land side
        {
            const double c0 = (double)R.c0;
            double colA[NCH], colL[NCH];
            int colC[NCH];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) { colA[ch] = 0.0; colL[ch] = 0.0; colC[ch] = 0; }
#pragma unroll
            for (int ri = 0; ri < RPW; ++ri) {
                const int r = wv * RPW + ri;
                unsigned carryC = 0;
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    const int k = ri * NCH + ch;
                    const bool ok = (R.okm >> k) & 1u;
                    const unsigned land = ok ? ((R.lw[k] >> (R.xcol[ch] & 31)) & 1u) : 0u;
                    T t0v = R.th[k];
                    if constexpr (FLY) {
                        // the sigmoid only where a lane of the wave stands on land (wave-uniform branch)   ref :166-167
                        if (__ballot(ok && R.zz[k] != T(0)) != 0) t0v = sb_t0<T>(R.th[k], R.zz[k], R.sg[k], sd, rr);
                    }
                    d[k] = ok ? (double)t0v - c0 : 0.0;
                    const uint64_t lm = __ballot(land);
                    if (lane == 0) s_land[par][r * NCH + ch] = lm;
                    colA[ch] += d[k];
                    colL[ch] += land ? d[k] : 0.0;
                    colC[ch] += (int)(carryC + (unsigned)__popcll(lm & le_mask));
                    carryC += (unsigned)__popcll(lm);
                    landbits |= land << k;
                }
            }
            // the band's column totals, prefixed along longitude: what the bands below add to every entry
            double carA = 0.0, carL = 0.0;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int c = ch * SB_WAVE + lane;
                const double tA = sb_wave_scan_add_f64(colA[ch]) + carA, tL = sb_wave_scan_add_f64(colL[ch]) + carL;
                if (c < W) { pA[wv * W + c] = tA; pL[wv * W + c] = tL; pC[wv * W + c] = colC[ch]; }
                if (ch + 1 < NCH) { carA = sb_readlane_f64(tA, 63); carL = sb_readlane_f64(tL, 63); }
            }
        }
        SB_T(2);                               // A1 compute
        lds_barrier();
        SB_T(3);                               // barrier 1
        // the next tile's loads fly under A2 and A3 (and under the other workgroups' staging); the entry
        // after it is fetched now
        const int tile_after = next_tile;
        if (PF && tile_after >= 0) thc_issue<T, TX, TY, H, NT, FLY, RPW, NCH>(job, tile_after, R);
        pos += G;
        next_tile = pos + G < nactive ? job.tile_list[1 + pos + G] : -1;
        // ---- A2: band cells -> list; finished tables ---------------------------------------------------
        int total;
        {
            const int pc = lane < TY ? __popcll(s_word[lane < TY ? lane : 0]) : 0;
            const int incl = sb_wave_scan_add(pc);
            total = __shfl(incl, SB_WAVE - 1);
            const int excl = incl - pc;
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                const int i = tid + q * NT;
                const int r = i / TX, c = i % TX;
                const bool in = i < TX * TY;
                const uint64_t w = s_word[in ? r : 0];
                const int at = __shfl(excl, in ? r : 0) + __popcll(w & ((1ull << c) - 1ull));
                if (in && ((w >> c) & 1ull)) s_cell[at] = (unsigned short)((r << 6) | c);
            }
        }
        {
            double offA[NCH], offL[NCH];
            int offC[NCH];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int c = ch * SB_WAVE + lane < W ? ch * SB_WAVE + lane : 0;
                offA[ch] = 0.0; offL[ch] = 0.0; offC[ch] = 0;
                for (int b = 0; b < wv; ++b) {               // wave-uniform trip count; fixed order
                    offA[ch] += pA[b * W + c];
                    offL[ch] += pL[b * W + c];
                    offC[ch] += pC[b * W + c];
                }
            }
            double runA[NCH], runL[NCH];
            int runC[NCH];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) { runA[ch] = 0.0; runL[ch] = 0.0; runC[ch] = 0; }
#pragma unroll
            for (int ri = 0; ri < RPW; ++ri) {
                const int r = wv * RPW + ri;
                unsigned carryC = 0;
                double carA = 0.0, carL = 0.0;
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    const int k = ri * NCH + ch;
                    const int c = ch * SB_WAVE + lane;
                    const bool land = (landbits >> k) & 1u;
                    const uint64_t lm = __ballot(land);
                    runA[ch] += d[k];
                    runL[ch] += land ? d[k] : 0.0;
                    runC[ch] += (int)(carryC + (unsigned)__popcll(lm & le_mask));
                    carryC += (unsigned)__popcll(lm);
                    const double vA = sb_wave_scan_add_f64(runA[ch]) + carA, vL = sb_wave_scan_add_f64(runL[ch]) + carL;
                    if (c < W) {
                        const int o = (r + 1) * P + c + 1;
                        sA[o] = vA + offA[ch];
                        sL[o] = vL + offL[ch];
                        sC[o] = (unsigned short)(runC[ch] + offC[ch]);
                    }
                    if (ch + 1 < NCH) { carA = sb_readlane_f64(vA, 63); carL = sb_readlane_f64(vL, 63); }
                }
            }
        }
        SB_T(4);                               // A2 (issue of the next tile, list, tables)
        lds_barrier();
        SB_T(5);                               // barrier 2
        // ---- A3: smallest radius whose square holds both classes, contrast, result -----------------------
        // Two rounds of independent probes instead of a bisection: every probe of a round is issued before
        // the first is used, so a cell costs three LDS round trips (two for the radius, one for the sums).
        // QI list entries of a thread go through the rounds together (all of them while a SIMD holds only two
        // waves; one at a time at four waves per SIMD, where the other waves cover the round trips and the
        // register budget is half).
        {
            constexpr int QI = NT >= 1024 ? 1 : CPT;
            int nnmax = 0;
            unsigned slow = 0;                   // bit q: list entry q takes the global-memory path
            int ccall[CPT];
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                const int i = tid + q * NT;
                ccall[q] = i < total ? (int)s_cell[i] : -1;
            }
#pragma unroll
            for (int qb = 0; qb < CPT; qb += QI) {
                int ccq[QI], limq[QI];
                bool validq[QI];
                SbCellState<T> cst[QI];
                unsigned short t1[QI][4][4];                 // round 1: four corners of four squares
#pragma unroll
                for (int j = 0; j < QI; ++j) {
                    const int q = qb + j;
                    validq[j] = q < CPT && ccall[q < CPT ? q : 0] >= 0;
                    ccq[j] = validq[j] ? ccall[q < CPT ? q : 0] : 0;
                    const int lx = ccq[j] & 63, ly = ccq[j] >> 6;
                    const int x = x0 + lx, y = y0 + ly;
                    int lim = H;
                    if (g.bnd == BND_HALO)                       // wave-uniform
                        lim = min(lim, min(min(x + g.h, g.nx - 1 - x + g.h), min(y + g.h, g.ny - 1 - y + g.h)));
                    limq[j] = lim;
                    if constexpr (!WF) {
                        cst[j] = SbCellState<T>{T(0), T(0), T(0), T(0)};
                        if (__ballot(validq[j]) != 0) cst[j] = sb_trigger_load<T>(job, (size_t)y * g.nx + x);   // clamped: a cell of the tile
                    }
                    const int cx = lx + H, cy = ly + H;
                    const int limc = max(lim, 1);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int rad = min((k + 1) * STEP, limc);
                        const int r0 = (cy - rad) * P, r1 = (cy + rad + 1) * P, a0 = cx - rad, a1 = cx + rad + 1;
                        t1[j][k][0] = sC[r1 + a1]; t1[j][k][1] = sC[r0 + a1]; t1[j][k][2] = sC[r1 + a0]; t1[j][k][3] = sC[r0 + a0];
                    }
                }
                int loq[QI], hiq[QI], nlhi[QI];
                bool fndq[QI];
                unsigned short t2[QI][STEP > 1 ? STEP - 1 : 1][4];
#pragma unroll
                for (int j = 0; j < QI; ++j) {
                    const int limc = max(limq[j], 1);
                    // the smallest of the four squares that is mixed brackets the radius from above, the largest
                    // that is not from below
                    int lo = 1, hi = limc, nl_hi = 0;
                    bool got = false;
#pragma unroll
                    for (int k = 3; k >= 0; --k) {
                        const int rad = min((k + 1) * STEP, limc);
                        const int nl = (int)t1[j][k][0] - (int)t1[j][k][1] - (int)t1[j][k][2] + (int)t1[j][k][3];
                        const bool mixed = nl > 0 && nl < (2 * rad + 1) * (2 * rad + 1);
                        if (mixed) { hi = rad; nl_hi = nl; got = true; }
                        else if (rad < hi) lo = max(lo, rad + 1);
                    }
                    fndq[j] = validq[j] && limq[j] >= 1 && got;
                    if (!got) { lo = limc; hi = limc; }
                    loq[j] = lo; hiq[j] = hi; nlhi[j] = nl_hi;
                    const int lx = ccq[j] & 63, ly = ccq[j] >> 6, cx = lx + H, cy = ly + H;
#pragma unroll
                    for (int m = 0; m < STEP - 1; ++m) {
                        const int rad = min(lo + m, hi);
                        const int r0 = (cy - rad) * P, r1 = (cy + rad + 1) * P, a0 = cx - rad, a1 = cx + rad + 1;
                        t2[j][m][0] = sC[r1 + a1]; t2[j][m][1] = sC[r0 + a1]; t2[j][m][2] = sC[r1 + a0]; t2[j][m][3] = sC[r0 + a0];
                    }
                }
                int nnq[QI], nlq[QI];
                double sums[QI][8];
                uint64_t ownw[QI];
#pragma unroll
                for (int j = 0; j < QI; ++j) {
                    int nn = hiq[j], nl = nlhi[j];
#pragma unroll
                    for (int m = STEP - 2; m >= 0; --m) {
                        const int rad = min(loq[j] + m, hiq[j]);
                        const int c = (int)t2[j][m][0] - (int)t2[j][m][1] - (int)t2[j][m][2] + (int)t2[j][m][3];
                        if (c > 0 && c < (2 * rad + 1) * (2 * rad + 1)) { nn = rad; nl = c; }
                    }
                    nnq[j] = nn; nlq[j] = nl;
                    const int lx = ccq[j] & 63, ly = ccq[j] >> 6, cx = lx + H, cy = ly + H;
                    const int r0 = (cy - nn) * P, r1 = (cy + nn + 1) * P, a0 = cx - nn, a1 = cx + nn + 1;
                    sums[j][0] = sL[r1 + a1]; sums[j][1] = sL[r0 + a1]; sums[j][2] = sL[r1 + a0]; sums[j][3] = sL[r0 + a0];
                    sums[j][4] = sA[r1 + a1]; sums[j][5] = sA[r0 + a1]; sums[j][6] = sA[r1 + a0]; sums[j][7] = sA[r0 + a0];
                    ownw[j] = s_land[par][cy * NCH + (cx >> 6)];
                }
#pragma unroll
                for (int j = 0; j < QI; ++j) {
                    const int lx = ccq[j] & 63, ly = ccq[j] >> 6, cx = lx + H;
                    const int x = x0 + lx, y = y0 + ly;
                    const size_t o = (size_t)y * g.nx + x;
                    const int nn = nnq[j], area = (2 * nn + 1) * (2 * nn + 1);
                    const double RL = (sums[j][0] - sums[j][1]) - (sums[j][2] - sums[j][3]);
                    const double RA = (sums[j][4] - sums[j][5]) - (sums[j][6] - sums[j][7]);
                    const T contrast = (T)(RL / (double)nlq[j] - (RA - RL) / (double)(area - nlq[j]));
                    // the cell's own class: the table's centre, except that the f2py boundary rule maps the
                    // centre of the window at the last longitude to column 1   ref :182-186, seabreeze_diag_python.f90:202
                    bool own = (ownw[j] >> (cx & 63)) & 1ull;
                    if (g.bnd == BND_WRAPPER && x == g.nx - 1 && validq[j]) own = sb_bit(job.clsbits, g.nw, x + g.h, y + g.h);
                    const T mul = own ? T(1) : T(-1);
                    if (fndq[j]) {
                        nnmax = max(nnmax, nn);
                        if constexpr (WF) job.thc[o] = mul * contrast;                 // ref :216; k_wind applies :235-266
                        else sb_trigger_update<T>(job, o, mul * contrast, cst[j]);      // ref :216, :235-266
                    } else if (validq[j]) slow |= 1u << (qb + j);
                }
            }
            // cells whose window outgrows the tile (none on a grid whose halo hint holds): global-memory path,
            // one copy of its code for all list entries of the thread
            if (__ballot(slow != 0) != 0) {
#pragma unroll 1
                for (int q = 0; q < CPT; ++q) {
                    if (!((slow >> q) & 1u)) continue;
                    int cc = ccall[0];
#pragma unroll
                    for (int j = 1; j < CPT; ++j) cc = q == j ? ccall[j] : cc;
                    const int x = x0 + (cc & 63), y = y0 + (cc >> 6);
                    const size_t o = (size_t)y * g.nx + x;
                    int cap = g.nx + g.ny;
                    if (g.bnd == BND_HALO)
                        cap = min(min(x + g.h, g.nx - 1 - x + g.h), min(y + g.h, g.ny - 1 - y + g.h));
                    bool one_class;
                    int nng;
                    const T cg = contrast_global(job, x, y, cap, sd, rr, nng, one_class);
                    atomicAdd(&job.counters[0], 1);
                    if (one_class) atomicAdd(&job.counters[1], 1);
                    nnmax = max(nnmax, nng);
                    const T mulg = sb_bit(job.clsbits, g.nw, x + g.h, y + g.h) ? T(1) : T(-1);
                    if constexpr (WF) job.thc[o] = mulg * cg;
                    else sb_trigger_update<T>(job, o, mulg * cg, sb_trigger_load<T>(job, o));
                }
            }
            // per-tile largest radius (diagnostic; reduced lazily by sb_last_counters); the flag k_scan raised is 1
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) nnmax = max(nnmax, __shfl_xor(nnmax, off));
            if (lane == 0 && nnmax > 1) atomicMax(&job.tile_nnmax[tile], nnmax);
        }
        // no barrier here: the next tile's A1 writes s_word and the band totals (last read before the barrier
        // above) and the other s_land buffer; its first barrier stands before any table or list write
        SB_T(6);                               // A3
        tile = tile_after;
        par ^= 1;
    }
#ifdef SB_STAMPS
    if (tid == 0) {
        acc[9] = w_begin;
        acc[10] = wall_clock64();
        for (int i = 0; i < SB_NSTAMP; ++i) job.stamps[(size_t)blockIdx.x * SB_NSTAMP + i] = acc[i];
    }
#endif
}

template <typename T, int TX, int TY, int H, int NT>
static void launch_thc3(const DiagJob<T> &job, int nblocks, hipStream_t st) {
    const dim3 gr(nblocks), bl(NT);
    if (job.t0_fly && job.wind_final) hipLaunchKernelGGL((k_thc3<T, TX, TY, H, NT, true, true>), gr, bl, 0, st, job);
    else if (job.t0_fly) hipLaunchKernelGGL((k_thc3<T, TX, TY, H, NT, true, false>), gr, bl, 0, st, job);
    else if (job.wind_final) hipLaunchKernelGGL((k_thc3<T, TX, TY, H, NT, false, true>), gr, bl, 0, st, job);
    else hipLaunchKernelGGL((k_thc3<T, TX, TY, H, NT, false, false>), gr, bl, 0, st, job);
}

template <typename T>
hipError_t sb_launch_thc(const DiagJob<T> &job, int H, int ncu, int nt, hipStream_t st) {
    const int nblocks = ncu;                     // one persistent workgroup per CU
    // 512 threads (8 waves of up to 256 registers) or 1024 (16 waves of up to 128); the wide-halo tiles (two 64-lane
    // chunks per staged row) need the LDS the second set of band totals would take
    if (H <= 8) {
        if (nt == 1024) launch_thc3<T, THC_TX, THC_TY, 8, 1024>(job, nblocks, st);
        else launch_thc3<T, THC_TX, THC_TY, 8, 512>(job, nblocks, st);
    } else if (H <= 16 && job.thc_ty == THC_TYS) {
        if (nt == 1024) launch_thc3<T, THC_TX, THC_TYS, 16, 1024>(job, nblocks, st);
        else launch_thc3<T, THC_TX, THC_TYS, 16, 512>(job, nblocks, st);
    } else if (H <= 16 && job.thc_ty == THC_TYL) {
        if (nt == 1024) launch_thc3<T, THC_TX, THC_TYL, 16, 1024>(job, nblocks, st);
        else launch_thc3<T, THC_TX, THC_TYL, 16, 512>(job, nblocks, st);
    } else if (H <= 16) {
        if (nt == 1024) launch_thc3<T, THC_TX, THC_TY, 16, 1024>(job, nblocks, st);
        else launch_thc3<T, THC_TX, THC_TY, 16, 512>(job, nblocks, st);
    }
    else if (H <= 24) launch_thc3<T, THC_TX, THC_TY24, 24, 512>(job, nblocks, st);
    else launch_thc3<T, THC_TX, THC_TY32, 32, 512>(job, nblocks, st);   // H == 32
    return hipGetLastError();
}
template hipError_t sb_launch_thc<float>(const DiagJob<float> &, int, int, int, hipStream_t);
template hipError_t sb_launch_thc<double>(const DiagJob<double> &, int, int, int, hipStream_t);

void sb_thc_tile_shape(int H, int nx, int rows, int ncu, int *tx, int *ty) {
    if (H <= 16) {
        *tx = THC_TX;
        *ty = THC_TY;
        // about a quarter of the tiles touch the coastal band: while even twice the tile count would leave
        // workgroups without a tile, use the 32-row tiles (H = 16 only)
        const long long full = (long long)((nx + THC_TX - 1) / THC_TX) * ((rows + THC_TYL - 1) / THC_TYL);
        if (H > 8 && full <= 2LL * ncu) *ty = THC_TYS;
    }
    else if (H <= 24) { *tx = THC_TX; *ty = THC_TY24; }
    else { *tx = THC_TX; *ty = THC_TY32; }
}
