// sb_thc_kernel.hip -- thermal heating contrast (the expanding-window land/sea mean
// difference of t0) on gfx950.   ref: generic/sea_breeze_diag.f90:188-216,
// python_wrapper/seabreezediag/seabreeze_diag_python.f90:187-221
//
// The reference re-sums a (2nn+1)^2 window from scratch at every radius nn until it holds
// both classes; only the last square matters.  Here a tile of 32 x TY cells plus a halo of
// H cells is held in LDS as three summed-area tables (all cells, land-side cells,
// land-side count), after which any square costs four LDS reads per table and the
// smallest valid radius is found by bisection on the count table (the "both classes present"
// predicate is monotone in nn).  Sums are taken about a per-tile offset c0, which cancels
// exactly in the difference of the two means and keeps the fp64 prefix sums small.
//
// One persistent workgroup per CU (k_thc3, below) works through the active tiles -- ranked by itself from the flags
// k_scan raised (FOLD) or taken from the list k_prep compacted: tiles that do not touch the coastal band (about
// 3 in 4) cost nothing.  Default instance: 1024 threads, every tile loaded when its turn comes (PFX = false).
#include "sb_thc_common.hpp"

// k_thc3 tiles are 32 longitudes wide: with the halo a staged row is exactly one 64-lane chunk (H = 16), so
// no lane of a staging load, an exp or an LDS write is padding.  48 latitudes by default: on the N1280 grid
// the coastal band touches 670 such tiles (575 of 64 rows, 1004 of 32) and tiles x staged rows is smallest there.
#ifndef THC_SKIP
#define THC_SKIP 0                // diagnostic builds only: bit 0 no sigmoid, 1 no scans, 2 no radius probes, 3 no sums, 4 no prefetch loads
#endif
#ifndef THC_T0_BATCH
#define THC_T0_BATCH 1             // t0 of a wave's rows in one branch-free stretch (0: a wave-uniform branch per row)
#endif
#ifndef THC_SEARCH
#define THC_SEARCH 1               // radius search on the count table: 0 two rounds of independent probes, 1 bisection
#endif
#ifndef THC_PC16
#define THC_PC16 1                 // count table rows 16 banks apart (0: the pitch of the fp64 tables)
#endif
#ifndef THC_QI
#define THC_QI 0                  // list entries a thread takes through the search rounds together; 0: by workgroup size
#endif
#define THC_TX 32
#define THC_TY24 32               // tile rows with a halo of 24 (81 x 81 table entries)
#define THC_TY32 16               // tile rows with a halo of 32: 81 x 97 table entries are what 160 KB of LDS hold

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

// everything a thread holds of a tile between the issue of its loads and A1
template <typename T, int NC, int NCH, bool FLY>
struct ThcRegs {
    T th[NC];                      // theta (FLY) or t0
    T zz[FLY ? NC : 1], sg[FLY ? NC : 1];
    uint32_t lw[NC];               // the 32-bit half of the land-side word that holds the cell
    int xcol[NCH];                 // array column of each chunk, -1: no such cell
    unsigned rowok;                // bit ri: row ri of the wave exists (wave-uniform)
    uint64_t bw0, bw1;             // the two words that hold the band bits of tile row tid (tid < TY)
    int bsh;                       // their shift, -1: no such row
    T c0;                          // the tile's offset
};

// The loads of a tile come in two parts so that the second can be spread over the work of the tile before it:
// thc_issue_begin works out the lane's column offsets and issues the few loads that belong to the tile as a whole,
// thc_issue_row issues the loads of one of the wave's rows.  (Pushing all 41 loads of a thread through the
// texture-address unit in one burst takes 2.6 k cycles per tile -- 512 bytes per wave instruction at 64 bytes per
// clock, eight waves -- during which the waves stand at their load instructions.)
template <int NCH>
struct ThcCols {
    unsigned colb[NCH], clsb[NCH];               // byte offsets of the lane's column: in a field row, in a row of the bit plane
    int y0;                                      // first interior row of the tile (wave-uniform)
};

template <typename T, int TX, int TY, int H, int NT, bool FLY, int RPW, int NCH>
__device__ __forceinline__ void thc_issue_begin(const DiagJob<T> &job, const ThcBufs<FLY> &B, int tile,
                                                ThcRegs<T, RPW * NCH, NCH, FLY> &R, ThcCols<NCH> &C) {
    constexpr int W = TX + 2 * H;
    const Geo g = job.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int ntx = job.thc_ntx;
    const int ty = tile / ntx;                   // tile is wave-uniform (scalar division)
    const int x0 = (tile - ty * ntx) * TX, y0 = ty * TY;
    C.y0 = y0;
    int X, Y;
    sb_map_cell(g, x0, y0, X, Y);
    const unsigned i00 = (unsigned)Y * (unsigned)g.nxh + (unsigned)X;
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
        const unsigned xc = ok ? (unsigned)Xc : 0u;              // every load is unconditional, from a clamped address
        C.colb[ch] = xc * (unsigned)sizeof(T);
        C.clsb[ch] = (xc >> 5) * 4u;
    }
    R.rowok = 0;
    // the tile's offset -- any common value conditions the sums, and theta at the tile origin needs
    // no sigmoid -- and the band words
    R.c0 = sb_buf_ld<T>(B.th, 0u, i00 * (unsigned)sizeof(T));
    {
        const bool have = tid < TY && y0 + tid < g.rows;
        const int Xb = x0 + g.h, wi = Xb >> 6;
        const uint64_t *row = job.bandbits + (size_t)((have ? y0 + tid : y0) + g.h) * g.nw;
        R.bw0 = row[wi];
        R.bw1 = row[wi + 1 < g.nw ? wi + 1 : wi];
        R.bsh = have ? (Xb & 63) + (wi + 1 < g.nw ? 0 : 64) : -1;      // +64: there is no second word
    }
}

template <typename T, int H, bool FLY, int RPW, int NCH>
__device__ __forceinline__ void thc_issue_row(const DiagJob<T> &job, const ThcBufs<FLY> &B, const ThcCols<NCH> &C, int wvu,
                                              int ri, ThcRegs<T, RPW * NCH, NCH, FLY> &R) {
    const Geo g = job.g;
    const int ys = C.y0 - H + wvu * RPW + ri;    // scalar: the wave's row
    int Yr;
    bool rowok = true;
    if (g.bnd == BND_HALO) { Yr = ys + g.h; rowok = Yr >= 0 && Yr < g.nyh; Yr = rowok ? Yr : 0; }
    else Yr = ys < 0 ? 0 : (ys >= g.ny ? g.ny - 1 : ys);
    const unsigned rowb = (unsigned)Yr * (unsigned)g.nxh * (unsigned)sizeof(T), wordb = (unsigned)Yr * (unsigned)g.nw * 8u;
    R.rowok |= (rowok ? 1u : 0u) << ri;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int k = ri * NCH + ch;
        R.th[k] = sb_buf_ld<T>(B.th, C.colb[ch], rowb);
        if constexpr (FLY) { R.zz[k] = sb_buf_ld<T>(B.zz, C.colb[ch], rowb); R.sg[k] = sb_buf_ld<T>(B.sg, C.colb[ch], rowb); }
        R.lw[k] = __builtin_amdgcn_raw_buffer_load_b32(B.cls, C.clsb[ch], wordb, 0);
    }
}

template <typename T, int TX, int TY, int H, int NT, bool FLY, int RPW, int NCH>
__device__ __forceinline__ void thc_issue(const DiagJob<T> &job, const ThcBufs<FLY> &B, int tile, int wvu,
                                          ThcRegs<T, RPW * NCH, NCH, FLY> &R) {
    ThcCols<NCH> C;
    thc_issue_begin<T, TX, TY, H, NT, FLY, RPW, NCH>(job, B, tile, R, C);
#pragma unroll
    for (int ri = 0; ri < RPW; ++ri) thc_issue_row<T, H, FLY, RPW, NCH>(job, B, C, wvu, ri, R);
}

#define THC_MAXCNT 256            // FOLD: 64-bit words of the active-tile bit plane a workgroup can hold (the host checks)

// FOLD: the tile of row-major rank r among the active tiles, or -1 past the last, from the bit plane in LDS (bit b of
// word k = tile 64 k + b).  Popcounts, one wave scan per 64 words and ballots; wave-uniform result; every wave of the
// workgroup computes the same.
__device__ __forceinline__ int thc_fold_pick(const uint64_t *s_bits, int nwords, int r, int lane) {
    int tile = -1, run = 0;
#pragma unroll
    for (int k = 0; k < THC_MAXCNT / SB_WAVE; ++k) {
        if (k * SB_WAVE >= nwords) break;                        // wave-uniform
        const uint64_t w = (k * SB_WAVE + lane < nwords) ? s_bits[k * SB_WAVE + lane] : 0ull;
        const int pc = __popcll(w);
        const int incl = sb_wave_scan_add(pc);
        const int before = run + incl - pc;                      // active tiles before this word
        run += __builtin_amdgcn_readlane(incl, SB_WAVE - 1);
        const uint64_t hit = __ballot(before <= r && r < before + pc);
        if (hit) {                                               // wave-uniform; at most one lane of one k
            const int src = __ffsll((unsigned long long)hit) - 1;
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)w, src);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(w >> 32), src);
            const uint64_t word = ((uint64_t)hi << 32) | lo;
            const int n = r - __builtin_amdgcn_readlane(before, src);
            // the n-th set bit of the word: lane j looks at bit j
            const bool me = ((word >> lane) & 1ull) && __popcll(word & ((1ull << lane) - 1ull)) == n;
            const uint64_t sel = __ballot(me);
            tile = (k * SB_WAVE + src) * 64 + __ffsll((unsigned long long)sel) - 1;
        }
    }
    return tile;
}

// FOLD (single-domain host-model calls, job.fold): no k_prep between k_scan and this kernel.  Every workgroup
//   * turns k_scan's tile flags into a bit plane (ballots, one barrier) and finds the tiles of ranks blockIdx,
//     blockIdx + G, ... in it with popcounts and ballots alone (row-major order, so the dealing of tiles to
//     workgroups is exactly that of k_prep's list),
//   * adds up k_scan's shifted sums of sigma in k_prep's order (bit-identical scalars in every thread of every
//     workgroup) while its first tile's loads are in flight; workgroup 0 publishes the scalars,
//   * and the last SB_SEG_PARTS workgroups -- the ones the dealing gives one tile fewer whenever it is uneven --
//     compact one sub-list each of the segments that hold band cells for k_wind, after their tiles.
// PFX false: no register prefetch of the next tile.
// Since round 3 this kernel serves halos of 24 and 32 cells only (the strip kernel has the rest) and the library
// instantiates it with FOLD = false, PFX = false (sb_launch_thc): k_prep makes the tile list and the scalars; with two
// chunks per staged row a prefetched tile would not fit the registers anyway.  The FOLD and PFX paths stay in the
// source as what was measured in round 2 (DESIGN.md, round 2 in the history).
template <typename T, int TX, int TY, int H, int NT, bool FLY, bool WF, bool FOLD = false, bool PFX = true>     // WF: k_wind applies the update (job.wind_final)
__global__ __launch_bounds__(NT) void k_thc3(const int *__restrict__ tile_list, const T *__restrict__ stats, int G, DiagJob<T> job) {
    constexpr int NWV = NT / SB_WAVE;
    constexpr int W = TX + 2 * H, HT = TY + 2 * H, P = W + 1;
    // The count table has a pitch of its own: its 2-byte entries make consecutive rows of pitch W + 1 (65 entries =
    // 32.5 banks) land on the same LDS banks, and a half-wave of the search holds band cells of two or three
    // rows.  A pitch of 16 banks modulo 32 (where the LDS holds it) moves the rows apart: THC_PC16 below.
    constexpr int PC = THC_PC16 && (size_t)(HT + 1) * (((W + 1 + 31) / 64) * 64 + 32) * 2 + 2 * (size_t)(HT + 1) * P * 8 < 150000
                           ? ((W + 1 + 31) / 64) * 64 + 32 : P;
    constexpr int RPW = HT / NWV;                // consecutive table rows a wave owns
    constexpr int NCH = (W + SB_WAVE - 1) / SB_WAVE;
    constexpr int NC = RPW * NCH;
    constexpr int CPT = (TX * TY + NT - 1) / NT; // list entries per thread in the search
    constexpr int STEP = H / 4;                  // first-round radii: STEP, 2 STEP, 3 STEP, H
    static_assert(HT % NWV == 0, "every wave owns the same number of rows");
    static_assert(H % 4 == 0 && TX <= SB_WAVE && TY <= SB_WAVE && TY <= NT && NC <= 32, "tile shape");
    static_assert((size_t)W * HT < 65536, "u16 count table");
    // The next tile's inputs are prefetched into registers a whole tile ahead -- except where a thread holds two
    // chunks of every row (halos of 24 and 32 cells): there the prefetch registers would spill, and the loads are
    // issued when the tile's turn comes.
    constexpr bool PF = PFX && NCH == 1;
    __shared__ double sA[(HT + 1) * P];          // SAT of (t0 - c0), every cell
    __shared__ double sL[(HT + 1) * P];          // SAT of (t0 - c0), land-side cells
    __shared__ unsigned short sC[(HT + 1) * PC]; // SAT of land-side count (its own pitch, see PC)
    __shared__ double pA[NWV * W], pL[NWV * W];  // per-wave band totals of every column
    __shared__ int pC[NWV * W];
    __shared__ uint64_t s_land[2][HT * NCH];     // land-side bits of every staged row; two buffers: the search of tile i reads
                                                 // them while a wave that is ahead already stages tile i + 1
    __shared__ uint64_t s_word[TY];              // band bits of every tile row
    __shared__ unsigned short s_cell[TX * TY];   // the tile's band cells, compacted
    __shared__ uint64_t s_bits[FOLD ? THC_MAXCNT : 1];           // FOLD: the active tiles as a bit plane
    __shared__ Moments s_wpart[FOLD ? NT / SB_WAVE : 1];
    __shared__ int s_scan[NT / SB_WAVE];
    __shared__ T s_sdr[2];                                       // band step: the sigmoid scalars the first wave derived
    // (a tile shape that does not fit is a compile error here, not a launch that aborts on the device: an instance of
    // 219,528 bytes once got as far as the GPU box)
    static_assert(sizeof(sA) + sizeof(sL) + sizeof(sC) + sizeof(pA) + sizeof(pL) + sizeof(pC) + sizeof(s_land) + sizeof(s_word) +
                          sizeof(s_cell) + sizeof(s_bits) + sizeof(s_wpart) + sizeof(s_scan) + sizeof(s_sdr) + 64 <= 160 * 1024,
                  "k_thc3: the LDS of one workgroup");

    const Geo g = job.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: row offsets stay scalar
    const int ntx = job.thc_ntx;
    const uint64_t le_mask = ~0ull >> (63 - lane);               // lanes 0 .. lane

#ifdef SB_STAMPS
    long long acc[SB_NSTAMP], t_last = clock64();
    const long long w_begin = wall_clock64();
    for (int i = 0; i < SB_NSTAMP; ++i) acc[i] = 0;
#endif
    // ---- prologue: the list of active tiles and the sigmoid scalars are k_prep's work.  The kernel starts from
    // cold caches, so what counts is the number of dependent round trips: kernel arguments -> {this workgroup's
    // first two list entries, the sigmoid scalars} -> the first tile's inputs.  The list ends in -1 entries, so the
    // count of active tiles is not needed.
    int pos = (int)blockIdx.x;                 // G = gridDim.x, as an argument: the launch-size block is one more load
    // (the two pointers the first round trip needs are the kernel's leading arguments: the hardware preloads them
    // into scalar registers with the launch, -mllvm -amdgpu-kernarg-preload-count, so that round trip does not
    // wait for a load of the argument block)
    // every line of the argument block is touched by the first batch of scalar loads (a later first touch would be
    // one more cold round trip): the two pointers the compiler would otherwise fetch where they are first used
    asm volatile("" ::"s"(job.tile_nnmax), "s"(job.counters));
    int cand0 = -1, cand1 = -1;
    T sd = T(0), rr = T(0);
    Moments pm = moments_empty();
    double shift_c = 0.0;
    int fold_nwords = 0;
    if constexpr (!FOLD) {
        cand0 = tile_list[1 + pos];
        cand1 = tile_list[1 + pos + G];
        if constexpr (FLY) {
            if (job.ngath > 0) {
                // band step: the first wave merges the moments gathered from all ranks -- lane b takes rank b, then the
                // wave tree of k_merge_moments, so every workgroup of every rank derives the same bits -- and leaves
                // the scalars in LDS for the barrier that ends the prologue; workgroup 0 publishes them
                if (wv == 0) {
                    Moments m = moments_empty();
                    for (int b = lane; b < job.ngath; b += SB_WAVE) m = moments_merge(m, job.gath[b]);
                    m = wave_merge(m);
                    if (lane == 0) {
                        T st4[4];
                        sigmoid_scalars<T>(m, st4);
                        s_sdr[0] = st4[0]; s_sdr[1] = st4[1];
                        if (blockIdx.x == 0) { for (int i = 0; i < 4; ++i) job.stats_out[i] = st4[i]; }
                    }
                }
            } else { sd = stats[0]; rr = stats[1]; }
        }
    } else {
        // -- the tile flags, NT at a time, as a bit plane in LDS: word c NWV + wv = the ballot of chunk c in wave wv,
        //    i.e. bit b of word k is tile 64 k + b (row-major).  One load round, one barrier; from there on every
        //    wave works for itself (identical results in all of them, no further exchange).
        const int ntiles = job.thc_ntx * job.thc_nty;
        const int nch = (ntiles + NT - 1) / NT;                  // nch * NWV <= THC_MAXCNT words (host)
        if (job.fold_nparts > 0) {
            if (tid < job.fold_nparts) pm = job.fold_partials[tid];
            shift_c = (double)job.sigma[(size_t)g.h * g.nxh + g.h];
        } else { sd = stats[0]; rr = stats[1]; }
        unsigned mine = 0;
        for (int base = 0; base < nch; base += 8) {              // 8 loads in flight (clamped, so none is conditional)
            int f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = (base + j) * NT + tid;
                f[j] = job.tile_nnmax[i < ntiles ? i : ntiles - 1];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = (base + j) * NT + tid;
                mine |= (i < ntiles && f[j] != 0) ? 1u << (base + j) : 0u;
            }
        }
        for (int c = 0; c < nch; ++c) {
            const uint64_t b = __ballot((mine >> c) & 1u);
            if (lane == 0) s_bits[c * NWV + wv] = b;
        }
        __syncthreads();
        // only what the first tile's loads need stands before them (this code runs once, from a cold instruction
        // cache): the second tile, the table borders and the statistics follow the issue
        fold_nwords = nch * NWV;
        cand0 = thc_fold_pick(s_bits, fold_nwords, (int)blockIdx.x, lane);
    }
    const size_t fbytes = (size_t)g.nxh * g.nyh * sizeof(T);
    ThcBufs<FLY> B;
    B.th = sb_make_rsrc(FLY ? (const void *)job.theta : (const void *)job.t0, fbytes);
    B.zz = sb_make_rsrc(FLY ? (const void *)job.z : (const void *)job.t0, fbytes);
    B.sg = sb_make_rsrc(FLY ? (const void *)job.sigma : (const void *)job.t0, fbytes);
    B.cls = sb_make_rsrc(job.clsbits, (size_t)g.nyh * g.nw * 8);
    if constexpr (!FOLD) {
        for (int i = tid; i < HT + 1; i += NT) { sA[i * P] = 0.0; sL[i * P] = 0.0; sC[i * PC] = 0; }
        for (int i = tid; i < P; i += NT) { sA[i] = 0.0; sL[i] = 0.0; sC[i] = 0; }
    }
#ifdef SB_STAMPS
    acc[11] = clock64() - t_last;              // (k_prep path: LDS borders zeroed, no load waited for yet)
#endif
    int tile = __builtin_amdgcn_readfirstlane(cand0);
#ifdef SB_STAMPS
    acc[12] = clock64() - t_last;              // first list entry has arrived
#endif
    ThcRegs<T, NC, NCH, FLY> R;
    if (PF && tile >= 0) thc_issue<T, TX, TY, H, NT, FLY, RPW, NCH>(job, B, tile, wv, R);
#ifdef SB_STAMPS
    acc[13] = clock64() - t_last;              // first tile's loads issued
#endif
    if constexpr (FOLD) {
        for (int i = tid; i < HT + 1; i += NT) { sA[i * P] = 0.0; sL[i * P] = 0.0; sC[i * PC] = 0; }
        for (int i = tid; i < P; i += NT) { sA[i] = 0.0; sL[i] = 0.0; sC[i] = 0; }
        cand1 = thc_fold_pick(s_bits, fold_nwords, (int)blockIdx.x + G, lane);
        // the waves' totals of k_scan's shifted sums travel through the barrier below
        if (job.fold_nparts > 0) wave_total_shifted_store(pm, s_wpart);
    }
    int next_tile = __builtin_amdgcn_readfirstlane(cand1);
    __syncthreads();
    if constexpr (!FOLD && FLY) {
        if (job.ngath > 0) { sd = s_sdr[0]; rr = s_sdr[1]; }
    }
    if constexpr (FOLD) {
        if (job.fold_nparts > 0) {                               // uniform
            // k_scan's shifted sums, one per thread, added up in k_prep's order; every thread holds the totals and
            // derives the scalars itself (identical bits everywhere); workgroup 0 publishes them
            const Moments m = moments_of_shifted(shift_c, block_total_shifted_finish<NWV>(s_wpart));
            T st4[4];
            sigmoid_scalars<T>(m, st4);
            sd = st4[0]; rr = st4[1];
            if (blockIdx.x == 0 && tid == 0) { for (int i = 0; i < 4; ++i) job.stats_out[i] = st4[i]; }
        }
    }
    SB_T(0);                                   // prologue

    int par = 0;
    int fold_it = 0;
    while (tile >= 0) {
        const int tyi = tile / ntx;
        const int x0 = (tile - tyi * ntx) * TX, y0 = tyi * TY;
        if (!PF) thc_issue<T, TX, TY, H, NT, FLY, RPW, NCH>(job, B, tile, wv, R);
        // ---- A1: registers -> t0 - c0, column sums of the wave's band of rows -------------------------
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
            if constexpr (FLY && THC_T0_BATCH) {
                // t0 of all the wave's rows in one branch-free stretch: the rows' polynomial chains are independent,
                // and without a branch between them the scheduler interleaves them (two waves per SIMD hide little
                // of a 13-deep chain of fp64 fmas).  One wave-uniform branch for the lot: a wave whose rows are all
                // sea (z = 0 -> t0 = theta exactly) skips it.     ref :166-167
                bool anyland = false;
#pragma unroll
                for (int k = 0; k < NC; ++k) anyland |= R.zz[k] != T(0);
                if (!(THC_SKIP & 1) && __ballot(anyland) != 0) {
#pragma unroll
                    for (int k = 0; k < NC; ++k) R.th[k] = sb_t0<T>(R.th[k], R.zz[k], R.sg[k], sd, rr);
                }
            }
#pragma unroll
            for (int ri = 0; ri < RPW; ++ri) {
                const int r = wv * RPW + ri;
                const bool rowok = (R.rowok >> ri) & 1u;         // wave-uniform
                unsigned carryC = 0;
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    const int k = ri * NCH + ch;
                    const bool ok = rowok && R.xcol[ch] >= 0;
                    const unsigned land = ok ? ((R.lw[k] >> (R.xcol[ch] & 31)) & 1u) : 0u;
                    T t0v = R.th[k];
                    if constexpr (FLY && !THC_T0_BATCH) {
                        // the sigmoid only where a lane of the wave stands on land (wave-uniform branch)   ref :166-167
                        if (!(THC_SKIP & 1) && __ballot(ok && R.zz[k] != T(0)) != 0) t0v = sb_t0<T>(R.th[k], R.zz[k], R.sg[k], sd, rr);
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
            // the band's column sums (the counts already prefixed along longitude): what the bands below start from
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int c = ch * SB_WAVE + lane;
                if (c < W) { pA[wv * W + c] = colA[ch]; pL[wv * W + c] = colL[ch]; pC[wv * W + c] = colC[ch]; }
            }
        }
        SB_T(2);                               // A1 compute
        lds_barrier();
        SB_T(3);                               // barrier 1
        // the next tile's loads fly under A2 and A3 (and under the other workgroups' staging); the list entry
        // after it is fetched now, a whole tile before it is needed
        const int tile_after = __builtin_amdgcn_readfirstlane(next_tile);
        ThcCols<NCH> Cn;
        if (PF && tile_after >= 0) thc_issue_begin<T, TX, TY, H, NT, FLY, RPW, NCH>(job, B, tile_after, R, Cn);
        pos += G;
        if constexpr (FOLD && PF) { ++fold_it; next_tile = tile_after >= 0 ? thc_fold_pick(s_bits, fold_nwords, (int)blockIdx.x + (fold_it + 1) * G, lane) : -1; }
        else if constexpr (!FOLD) next_tile = tile_after >= 0 ? tile_list[1 + pos + G] : -1;   // (past a -1 entry nothing is read) made scalar where consumed
        // (FOLD without the prefetch: the tile after next is picked at the end of the iteration, where few registers are live)
        // ---- A2: band cells -> list; finished tables ---------------------------------------------------
        int total;
        {
            const int pc = lane < TY ? __popcll(s_word[lane < TY ? lane : 0]) : 0;
            const int incl = sb_wave_scan_add(pc);
            total = __builtin_amdgcn_readlane(incl, SB_WAVE - 1);
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
            // what the rows above this band sum to in every column: the running column prefix starts there, so
            // the scan along longitude of a row's running sums is the finished table row
            double runA[NCH], runL[NCH];
            int runC[NCH];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int c = ch * SB_WAVE + lane < W ? ch * SB_WAVE + lane : 0;
                runA[ch] = 0.0; runL[ch] = 0.0; runC[ch] = 0;
                if constexpr (NWV <= 8) {
                    // every band's entry is read (one LDS round trip for all of them), the bands above are added
                    // in their fixed order
                    double xa[NWV - 1], xl[NWV - 1];
                    int xc[NWV - 1];
#pragma unroll
                    for (int b = 0; b < NWV - 1; ++b) { xa[b] = pA[b * W + c]; xl[b] = pL[b * W + c]; xc[b] = pC[b * W + c]; }
#pragma unroll
                    for (int b = 0; b < NWV - 1; ++b)
                        if (b < wv) { runA[ch] += xa[b]; runL[ch] += xl[b]; runC[ch] += xc[b]; }     // wave-uniform condition
                } else {
                    for (int b = 0; b < wv; ++b) {               // scalar trip count; fixed order
                        runA[ch] += pA[b * W + c];
                        runL[ch] += pL[b * W + c];
                        runC[ch] += pC[b * W + c];
                    }
                }
            }
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
                    const double vA = ((THC_SKIP & 2) ? runA[ch] : sb_wave_scan_add_f64(runA[ch])) + carA, vL = ((THC_SKIP & 2) ? runL[ch] : sb_wave_scan_add_f64(runL[ch])) + carL;
                    if (c < W) {
                        const int o = (r + 1) * P + c + 1;
                        sA[o] = vA;
                        sL[o] = vL;
                        sC[(r + 1) * PC + c + 1] = (unsigned short)runC[ch];
                    }
                    if (ch + 1 < NCH) { carA = sb_readlane_f64(vA, 63); carL = sb_readlane_f64(vL, 63); }
                }
                // the next tile's loads of this row go out between the rows' scans, not in one burst
                if (PF && tile_after >= 0) thc_issue_row<T, H, FLY, RPW, NCH>(job, B, Cn, wv, ri, R);
            }
        }
        SB_T(4);                               // A2 (issue of the next tile, list, tables)
        lds_barrier();
        SB_T(5);                               // barrier 2
        // ---- A3: smallest radius whose square holds both classes, contrast, result -----------------------
        // THC_SEARCH 1 (default): the widest square, then a branch-free bisection on the count table -- 4 u16 reads per
        // probe, 1 + log2(H) dependent probes; QI list entries of a thread go through the probes together, so their
        // LDS round trips overlap (all of them while a SIMD holds only two waves; one at a time at four waves per
        // SIMD, where the other waves cover the round trips and the register budget is half).
        // THC_SEARCH 0 (kept for measurement): two rounds of independent probes at radii STEP, 2 STEP .. H and then
        // the STEP - 1 radii below the bracket: 3 round trips instead of 6, but 28 reads instead of 20 -- measured
        // slower (7.4k against 5.4k cycles per tile): this phase is bound by LDS issue, not by latency.
        {
            constexpr int QI = THC_QI > 0 ? (THC_QI < CPT ? THC_QI : CPT) : (NT >= 1024 ? 1 : CPT);
            int nnmax = 0;
            unsigned slow = 0;                   // bit q: list entry q takes the global-memory path
            int ccall[CPT];
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                const int i = tid + q * NT;
                ccall[q] = i < total ? (int)s_cell[i] : -1;
            }
            const bool limited = g.bnd == BND_HALO;          // wave-uniform
#pragma unroll
            for (int qb = 0; qb < CPT; qb += QI) {
                int ccq[QI], limq[QI], baseq[QI], basec[QI];
                bool validq[QI];
                SbCellState<T> cst[QI];
                int nl1[QI][4];                              // round 1: land-side count of four squares
#pragma unroll
                for (int j = 0; j < QI; ++j) {
                    const int q = qb + j;
                    validq[j] = q < CPT && ccall[q < CPT ? q : 0] >= 0;
                    ccq[j] = validq[j] ? ccall[q < CPT ? q : 0] : 0;
                    const int lx = ccq[j] & 63, ly = ccq[j] >> 6;
                    const int x = x0 + lx, y = y0 + ly;
                    int lim = H;
                    if (limited) lim = min(lim, min(min(x + g.h, g.nx - 1 - x + g.h), min(y + g.h, g.ny - 1 - y + g.h)));
                    limq[j] = lim;
                    if constexpr (!WF) {
                        cst[j] = SbCellState<T>{T(0), T(0), T(0), T(0)};
                        if (__ballot(validq[j]) != 0) cst[j] = sb_trigger_load<T>(job, (size_t)y * g.nx + x);   // clamped: a cell of the tile
                    }
                    baseq[j] = (ly + H) * P + lx + H;        // the cell's own table entry (row cy, column cx)
                    basec[j] = (ly + H) * PC + lx + H;       // ... in the count table
                    const int limc = max(lim, 1);
#pragma unroll
                    for (int k = (THC_SEARCH == 1 ? 3 : 0); k < 4; ++k) {
                        // C(r1,a1) - C(r0,a1) - C(r1,a0) + C(r0,a0), r0 = cy-rad, r1 = cy+rad+1, a0 = cx-rad, a1 = cx+rad+1
                        int rad = (k + 1) * STEP;
                        if (limited) rad = min(rad, limc);
                        const unsigned short *t = sC + basec[j];
                        if (THC_SKIP & 4) nl1[j][k] = k + 1; else
                        nl1[j][k] = (int)t[(rad + 1) * (PC + 1)] - (int)t[-rad * PC + rad + 1] - (int)t[(rad + 1) * PC - rad] + (int)t[-rad * (PC + 1)];
                    }
                }
                int loq[QI], hiq[QI], nlhi[QI], nl2[QI][STEP > 1 ? STEP - 1 : 1];
                bool fndq[QI];
                if constexpr (THC_SEARCH == 1) {
                    // bisection on the count table: 4 reads per probe, the widest square first, then ITER halvings
                    // (20 reads for a halo of 16 against the 28 of the two-round form, five dependent rounds against two;
                    // the rounds of the QI list entries overlap)
                    constexpr int ITER = (H <= 2 ? 1 : H <= 4 ? 2 : H <= 8 ? 3 : H <= 16 ? 4 : 5);
                    auto count = [&](int j, int rad) {
                        const unsigned short *t = sC + basec[j];
                        const int q = rad * (PC + 1);
                        return (int)t[q + PC + 1] - (int)t[-q + 2 * rad + 1] - (int)t[q + PC - 2 * rad] + (int)t[-q];
                    };
#pragma unroll
                    for (int j = 0; j < QI; ++j) {
                        const int limc = max(limq[j], 1);
                        const int nl = nl1[j][3];                // the square of radius min(H, lim)
                        const bool got = nl > 0 && nl < (2 * limc + 1) * (2 * limc + 1);
                        fndq[j] = validq[j] && limq[j] >= 1 && got;
                        loq[j] = got ? 1 : limc; hiq[j] = limc; nlhi[j] = nl;
                    }
#pragma unroll
                    for (int it = 0; it < ITER; ++it) {
                        int nlm[QI];
#pragma unroll
                        for (int j = 0; j < QI; ++j) nlm[j] = count(j, (loq[j] + hiq[j]) >> 1);
#pragma unroll
                        for (int j = 0; j < QI; ++j) {
                            const int mid = (loq[j] + hiq[j]) >> 1;
                            const bool act = loq[j] < hiq[j], ok = nlm[j] > 0 && nlm[j] < (2 * mid + 1) * (2 * mid + 1);
                            nlhi[j] = (act && ok) ? nlm[j] : nlhi[j];
                            hiq[j] = (act && ok) ? mid : hiq[j];
                            loq[j] = (act && !ok) ? mid + 1 : loq[j];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < QI; ++j)
                        for (int m = 0; m < STEP - 1; ++m) nl2[j][m] = 0;      // no second round: hi is the radius
                } else {
#pragma unroll
                for (int j = 0; j < QI; ++j) {
                    const int limc = max(limq[j], 1);
                    // the smallest of the four squares that is mixed brackets the radius from above, the largest
                    // that is not from below
                    int lo = 1, hi = limited ? limc : H, nl_hi = 0;
                    bool got = false;
#pragma unroll
                    for (int k = 3; k >= 0; --k) {
                        int rad = (k + 1) * STEP;
                        if (limited) rad = min(rad, limc);
                        const int nl = nl1[j][k];
                        const bool mixed = nl > 0 && nl < (2 * rad + 1) * (2 * rad + 1);
                        if (mixed) { hi = rad; nl_hi = nl; got = true; }
                        else if (rad < hi) lo = max(lo, rad + 1);
                    }
                    fndq[j] = validq[j] && limq[j] >= 1 && got;
                    // nothing found: probe below the widest square all the same (in bounds, results unused)
                    if (!got) { hi = limited ? limc : H; lo = max(1, hi - (STEP - 1)); }
                    loq[j] = lo; hiq[j] = hi; nlhi[j] = nl_hi;
                    // rad = lo + m: the four corners move by constant strides from the corners of radius lo
                    const unsigned short *t11 = sC + basec[j] + (lo + 1) * (PC + 1), *t01 = sC + basec[j] - lo * PC + lo + 1,
                                         *t10 = sC + basec[j] + (lo + 1) * PC - lo, *t00 = sC + basec[j] - lo * (PC + 1);
#pragma unroll
                    for (int m = 0; m < STEP - 1; ++m) {
                        if (THC_SKIP & 4) nl2[j][m] = m + 1; else
                        if (!limited || lo + m <= hi)
                            nl2[j][m] = (int)t11[m * (PC + 1)] - (int)t01[m * (1 - PC)] - (int)t10[m * (PC - 1)] + (int)t00[-m * (PC + 1)];
                        else nl2[j][m] = 0;
                    }
                }
                }
                int nnq[QI], nlq[QI];
                double sums[QI][8];
                uint64_t ownw[QI];
#pragma unroll
                for (int j = 0; j < QI; ++j) {
                    int nn = hiq[j], nl = nlhi[j];
#pragma unroll
                    for (int m = (THC_SEARCH == 1 ? -1 : STEP - 2); m >= 0; --m) {
                        const int rad = loq[j] + m;
                        const int c = nl2[j][m];
                        if (rad < hiq[j] && c > 0 && c < (2 * rad + 1) * (2 * rad + 1)) { nn = rad; nl = c; }
                    }
                    nnq[j] = nn; nlq[j] = nl;
                    const int b = baseq[j];
                    const int o11 = b + (nn + 1) * (P + 1), o01 = b - nn * P + nn + 1, o10 = b + (nn + 1) * P - nn, o00 = b - nn * (P + 1);
                    if (THC_SKIP & 8) { for (int e = 0; e < 8; ++e) sums[j][e] = (double)(o11 + e); } else {
                    sums[j][0] = sL[o11]; sums[j][1] = sL[o01]; sums[j][2] = sL[o10]; sums[j][3] = sL[o00];
                    sums[j][4] = sA[o11]; sums[j][5] = sA[o01]; sums[j][6] = sA[o10]; sums[j][7] = sA[o00]; }
                    const int cx = (ccq[j] & 63) + H, cy = (ccq[j] >> 6) + H;
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
                    // the two means by reciprocals of the (small, exact) counts: within an ulp of the quotients
                    const T contrast = (T)(RL * sb_inv((double)nlq[j]) - (RA - RL) * sb_inv((double)(area - nlq[j])));
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
            nnmax = sb_wave_max_to_last(nnmax);
            if (lane == SB_WAVE - 1 && nnmax > 1) atomicMax(&job.tile_nnmax[tile], nnmax);
        }
        // no barrier here: the next tile's A1 writes s_word and the band totals (last read before the barrier
        // above) and the other s_land buffer; its first barrier stands before any table or list write
        SB_T(6);                               // A3
        tile = tile_after;
        if constexpr (FOLD && !PF) { ++fold_it; next_tile = tile >= 0 ? thc_fold_pick(s_bits, fold_nwords, (int)blockIdx.x + (fold_it + 1) * G, lane) : -1; }
        par ^= 1;
    }
    if constexpr (FOLD) {
        // ---- k_wind's segment lists: sub-list `part` holds the segments with band cells of its contiguous range of the
        // band plane, in ascending order (exactly what k_prep writes) ----
        const unsigned nseg = (unsigned)g.nyh * (unsigned)g.nw;
        const unsigned cap = (unsigned)job.seg_cap;
        for (int part = G - 1 - (int)blockIdx.x; part < SB_SEG_PARTS; part += G) {
            if (part < 0) break;
            const unsigned s0 = (unsigned)part * cap, s1 = min(s0 + cap, nseg);
            const unsigned per = (cap + NT - 1) / NT;
            const unsigned a0 = min(s0 + (unsigned)tid * per, s1), a1 = min(a0 + per, s1);
            int cnt = 0;
            for (unsigned sg = a0; sg < a1; ++sg) cnt += job.bandbits[sg] != 0 ? 1 : 0;
            int total;
            int at = thc_block_excl_scan<NT>(cnt, s_scan, total);
            SbSegEntry *list = job.seg_list + (size_t)part * cap;
            for (unsigned sg = a0; sg < a1; ++sg) {
                const uint64_t w = job.bandbits[sg];
                if (w) { SbSegEntry e; e.word = w; e.seg = sg; e.pad = 0; list[at++] = e; }
            }
            if (tid == 0) job.seg_count[part] = total;
        }
    }
#ifdef SB_STAMPS
    if (tid == 0) {
        acc[9] = w_begin;
        acc[10] = wall_clock64();
        for (int i = 0; i < SB_NSTAMP; ++i) job.stamps[(size_t)blockIdx.x * SB_NSTAMP + i] = acc[i];
    }
    // the same sums as every wave of the first 64 workgroups saw them (rows 1024 ..): where do waves wait for each other?
    if (lane == 0 && blockIdx.x < 64)
        for (int i = 0; i < SB_NSTAMP; ++i) job.stamps[(size_t)(1024 + blockIdx.x * NWV + wv) * SB_NSTAMP + i] = acc[i];
#endif
}

template <typename T, int TX, int TY, int H, int NT>
static void launch_thc3(const DiagJob<T> &job, int nblocks, hipStream_t st) {
    const dim3 gr(nblocks), bl(NT);
    // (tile list and sigmoid scalars from k_prep, every tile loaded when its turn comes: FOLD = false, PFX = false)
    if (job.t0_fly && job.wind_final) hipLaunchKernelGGL((k_thc3<T, TX, TY, H, NT, true, true, false, false>), gr, bl, 0, st, (const int *)job.tile_list, job.stats, nblocks, job);
    else if (job.t0_fly) hipLaunchKernelGGL((k_thc3<T, TX, TY, H, NT, true, false, false, false>), gr, bl, 0, st, (const int *)job.tile_list, job.stats, nblocks, job);
    else if (job.wind_final) hipLaunchKernelGGL((k_thc3<T, TX, TY, H, NT, false, true, false, false>), gr, bl, 0, st, (const int *)job.tile_list, job.stats, nblocks, job);
    else hipLaunchKernelGGL((k_thc3<T, TX, TY, H, NT, false, false, false, false>), gr, bl, 0, st, (const int *)job.tile_list, job.stats, nblocks, job);
}

// LDS halos of 24 and 32 cells (distance fields made with windows of 17 .. 31 cells: the N2560 grid); radii up to 16
// are the strip kernel's (sb_strip_kernel.hip)
template <typename T>
hipError_t sb_launch_thc(const DiagJob<T> &job, int H, int ncu, hipStream_t st) {
    const int nblocks = ncu;                     // one persistent workgroup per CU
    // (the two instances that exist; anything else -- a halo the tables cannot hold, a tile grid other than the one
    // sb_thc_tile_shape gave for this halo -- is refused here rather than launched)
    int tx, ty;
    sb_thc_tile_shape(H, &tx, &ty);
    if ((H != 24 && H != 32) || job.thc_ty != ty || job.thc_txs != (tx == 32 ? 5 : 6) || job.strip) return hipErrorInvalidValue;
    if (H == 24) launch_thc3<T, THC_TX, THC_TY24, 24, 512>(job, nblocks, st);
    else launch_thc3<T, THC_TX, THC_TY32, 32, 512>(job, nblocks, st);
    return hipGetLastError();
}
template hipError_t sb_launch_thc<float>(const DiagJob<float> &, int, int, hipStream_t);
template hipError_t sb_launch_thc<double>(const DiagJob<double> &, int, int, hipStream_t);

void sb_thc_tile_shape(int H, int *tx, int *ty) {
    *tx = THC_TX;
    *ty = H <= 24 ? THC_TY24 : THC_TY32;
}
