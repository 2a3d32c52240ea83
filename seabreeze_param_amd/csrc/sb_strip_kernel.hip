// sb_strip_kernel.hip -- thermal heating contrast (the expanding-window land/sea mean difference of t0)
// on gfx950 for search radii up to 16 cells: marching strips.
// ref: generic/sea_breeze_diag.f90:166-167 (t0), :188-216 (window search and contrast),
//      python_wrapper/seabreezediag/seabreeze_diag_python.f90:187-221
//
// The reference re-sums a (2nn+1)^2 window from scratch at every radius nn until it holds both classes; only the
// last square matters, and "holds both classes" is monotone in nn.  So summed-area tables (all cells, land-side
// cells, land-side count) answer any square with four reads per table and a bisection on the count table finds nn.
//
// Layout.  The grid is cut into strips of 32 owned longitudes; with the halo of 16 cells either side a staged row
// is exactly one 64-lane wave (lane = column).  A strip is cut into blocks of 16 rows (one row per wave of the
// 1024-thread workgroup).  k_scan raises a flag per (strip, block) that holds a coastal-band cell; the flags are laid
// out strip-major with one virtual block above and below every strip, so that "the blocks before and after an
// active block" is plain bit arithmetic on a bit plane.  Every workgroup takes an equal share of the active blocks
// in that order -- consecutive blocks down a strip -- and MARCHES: it stages block after block into a ring of 128
// table rows in LDS, queries the band cells of block j once block j+1 is in the ring, and never stages a row twice
// inside a run (where the tile kernel re-staged a 16-row halo above and below every 48-row tile).
//
// Arithmetic.  t0 is turned into 64-bit fixed point (2^-40 K: below the spacing of doubles near 300 K by 16, so
// window sums are EXACT integer sums of values rounded once) -- the prefix sums are then integer adds, which
// (a) wrap harmlessly, so the tables need no per-tile offset and the ring never has to be re-based,
// (b) take two full-rate DPP instructions per scan step (v_add_co / v_addc with the lane shift fused), where the
//     fp64 scan took two DPP moves and a half-rate v_add_f64,
// (c) make the result independent of how the grid is cut into strips, blocks, bands or GPUs, bit for bit.
//
// Per block (one barrier, LDS only: the prefetched global loads stay in flight):
//   S1  every wave: its row's inputs (prefetched three blocks ahead into registers) -> t0 -> fixed point ->
//       prefix along longitude of {all, land-side, land-side count} -> ring row (not yet summed along latitude);
//       without a stored plan waves 8-15 also list the band cells of the block that is queried in this step
//   --- barrier ---
//   S2  waves 5-7: one table each, prefix along latitude of the 16 new rows (running column totals in registers)
//       waves 0-7: 64 band cells each of the block staged two steps ago: window radius, contrast -> thc
//       (the other waves go straight on to S1 of the next block)
//
// The plan -- shares, order of steps, cell lists with every window's radius, count and class -- is stored in device
// memory and used again for as long as k_scan finds the band and land-side planes unchanged (see `cached` below).
// DESIGN.md section 2.4 has the measurements behind the choices, and what hipcc does with loads kept in flight.
#include "sb_thc_common.hpp"
#include "sb_strip_common.hpp"

#define STRIP_H 16                // halo of the tables = largest radius answered from LDS
#define STRIP_W 64                // staged columns = lanes
#define STRIP_SW (STRIP_W - 2 * STRIP_H)
#define STRIP_C 16                // rows per block = waves per workgroup
#define STRIP_NT 1024
#define STRIP_RING 128            // ring rows (8 blocks): a query of block j reads rows of blocks j-2 .. j+1 while
                                  // block j+3 may already be written
#define STRIP_P (STRIP_W + 1)     // table pitch: column 0 is the zero column
#define STRIP_MAXW 768            // 64-bit words of the position plane a workgroup can hold (49,151 positions)
#define STRIP_SCHED 384           // steps of one round of a workgroup
#define STRIP_ROUND 90            // active blocks of one round (at most 3 x 90 staged blocks + 90 drain + 3 warm-up + 2 padding steps)
#define STRIP_FB 40               // fractional bits of the fixed-point t0
#define STRIP_DEPTH 3             // blocks of inputs in flight per wave


// Diagnostic build (-DSB_STAMPS: `make stamps`, tools/stamp_strip.py): every wave leaves the 100 MHz wall clock at a few
// marks -- one scalar clock read and one exec-masked store each, no registers held (an earlier version summed shader
// clocks per phase in 32 registers per lane: the spills that caused distorted what it measured).
#if defined(SB_STAMPS) && !defined(SB_STAMPS_WIND)
#define SB_T(i) do { if (lane == 0) job.stamps[(size_t)(blockIdx.x * (STRIP_NT / SB_WAVE) + wv) * SB_NSTAMP + (i)] = wall_clock64(); } while (0)
#else
#define SB_T(i) do { } while (0)
#endif

// The fp64 constants of a staged row -- the logistic's argument reduction and Taylor coefficients, the fixed-point
// conversion -- live in constant memory and are fetched by scalar loads where they are used (two s_load_dwordx16 per
// row, from the scalar cache).  As literals they are loop invariants the compiler keeps in some 40 scalar registers for
// the whole march, and the registers it then has to spill cost the staging code a quarter of its vector instructions
// (v_readlane reloads).  The pointer is made opaque per use so that the loads are not hoisted.
__constant__ double sb_strip_k[20] = {
    1.4426950408889634, 0.6931471805599453, 2.3190468138462996e-17, -708.0, 700.0,            // 0..4: log2(e), ln2 hi, lo, clamps
    1.6666666666666666e-01, 4.1666666666666664e-02, 8.333333333333333e-03, 1.388888888888889e-03,     // 5..8: 1/3! .. 1/6!
    1.984126984126984e-04, 2.48015873015873e-05, 2.7557319223985893e-06, 2.755731922398589e-07,     // 9..12: 1/7! .. 1/10!
    2.505210838544172e-08,                                                                         // 13: 1/11!
    -1024.0, 1024.0, 0x1p40, 0x1.8p52,                                                               // 14..17: fixed point
    -0.0060956, 0.0};                                                                               // 18: gmma (ref :138)
typedef const __attribute__((address_space(4))) double *sb_cdp;
#ifndef STRIP_KTAB
#define STRIP_KTAB 0              // 0: the constants as literals (measured faster by 1.6 us: the scalar loads stall the row);
                                  // 1: by scalar loads from the table
#endif
#if STRIP_KTAB
#define SB_K(i, lit) k[i]
#else
#define SB_K(i, lit) (lit)
#endif

// 1 / (1 + exp(y)): sb_logistic_of_neg<double> (sb_device.hpp) operation for operation -- the same bits --, its
// constants optionally from the table
__device__ __forceinline__ double strip_logistic_of_neg(double y, sb_cdp k) {
    y = fmin(fmax(y, -708.0), 700.0);                         // (literals: known not to be NaN, no canonicalisation)
    const double n = __builtin_rint(y * SB_K(0, 1.4426950408889634));
    double r = __builtin_fma(-n, SB_K(1, 0.6931471805599453), y);
    r = __builtin_fma(-n, SB_K(2, 2.3190468138462996e-17), r);                           // |r| <= ln2/2
    // exp(r), degree 11, as E(r^2) + r O(r^2): two Horner chains of five, every multiply-add with ONE constant (a
    // scalar operand; a second one would have to be copied to vector registers first)
    const double r2 = r * r;
    double e = __builtin_fma(r2, SB_K(12, 2.755731922398589e-07), SB_K(10, 2.48015873015873e-05));               // 1/10!, 1/8!
    double o = __builtin_fma(r2, SB_K(13, 2.505210838544172e-08), SB_K(11, 2.7557319223985893e-06));               // 1/11!, 1/9!
    e = __builtin_fma(e, r2, SB_K(8, 1.388888888888889e-03));  o = __builtin_fma(o, r2, SB_K(9, 1.984126984126984e-04));      // 1/6!, 1/7!
    e = __builtin_fma(e, r2, SB_K(6, 4.1666666666666664e-02));  o = __builtin_fma(o, r2, SB_K(7, 8.333333333333333e-03));      // 1/4!, 1/5!
    e = __builtin_fma(e, r2, 0.5);   o = __builtin_fma(o, r2, SB_K(5, 1.6666666666666666e-01));      // 1/2!, 1/3!
    e = __builtin_fma(e, r2, 1.0);   o = __builtin_fma(o, r2, 1.0);
    const double p = __builtin_fma(o, r, e);
    const double x = 1.0 + ldexp(p, (int)n);
    double q = __builtin_amdgcn_rcp(x);                       // v_rcp_f64 (2^-24 relative) and two Newton steps, as sb_logistic_of_neg
    q = __builtin_fma(q, __builtin_fma(-x, q, 1.0), q);
    q = __builtin_fma(q, __builtin_fma(-x, q, 1.0), q);
    return q;
}
// t0 = theta - (gmma*z)*sigmoid(sigma)   ref: generic/sea_breeze_diag.f90:166-167,478-480 (as sb_t0)
// (no branch for sea level: there gmma * z is a signed zero, the product with the -- finite or NaN -- sigmoid too, and
// theta comes back bit for bit, or NaN exactly where the reference's expression gives NaN)
__device__ __forceinline__ double strip_t0(double theta, double z, double sigma, double sd, double r, sb_cdp k) {
    return theta - ((SB_K(18, -0.0060956) * z) * strip_logistic_of_neg(-sd * (sigma - r), k));
}
__device__ __forceinline__ float strip_t0(float theta, float z, float sigma, float sd, float r, sb_cdp) {
    return sb_t0<float>(theta, z, sigma, sd, r);
}

// t0 (K) -> fixed point, BIASED: fma rounds x * 2^40 + 1.5 * 2^52 to an integer held in the mantissa (|x| < 2048 K:
// anything a temperature can be; beyond, the sums are garbage, not a fault), and the bits of that double are the value
// plus the constant SB_FIX_BIAS.  The bias stays in the tables -- a window of n cells holds n of them, which the
// query takes out again -- so a staged cell costs one instruction here instead of five (clamps, 64-bit subtraction).
#define SB_FIX_BIAS 0x4338000000000000ull
__device__ __forceinline__ u64 sb_to_fixed(double x, sb_cdp k) {
    return (u64)__double_as_longlong(__builtin_fma(x, SB_K(16, 0x1p40), SB_K(17, 0x1.8p52)));
}

// what a lane holds of one staged row between the issue of its loads and S1
template <typename T, bool FLY>
struct StripRegs {
    T th;                          // theta (FLY) or t0
    T zz, sg;                      // z, sigma (FLY only)
    uint32_t lw;                   // the 32-bit half of the land-side word that holds the cell
    uint32_t lbit;                 // the cell's bit in lw; 0: no such cell
};

// FLY: t0 from theta, z, sigma while staging.  The contrast goes to thc; thresholds and state update are k_wind's (a
// cell's winds and state loaded here, next to the prefetched blocks of the march, would drain them at every step).
template <typename T, bool FLY>
__global__ __launch_bounds__(STRIP_NT) void k_strip(char *plan, const int *plan_gen, const Moments *fold_partials, int G, StripJob<T> job) {
    // (the three pointers the first loads of the kernel hang on come as leading arguments: with
    // -amdgpu-kernarg-preload-count=7 they are in scalar registers when the wave starts, and the plan header, the
    // change counter and k_scan's partial sums are requested without waiting for a load of the argument block)
    constexpr int H = STRIP_H, W = STRIP_W, SW = STRIP_SW, C = STRIP_C, P = STRIP_P, NWV = STRIP_NT / SB_WAVE;
    constexpr int RM = STRIP_RING - 1;
    static_assert(NWV == C && SW == 32, "one staged row per wave, 32 owned columns");
    static_assert(STRIP_SCHED == SB_PLAN_SCHED, "a stored plan holds one round's steps");
    __shared__ u64 sA[STRIP_RING * P];                 // prefix sums of t0 (fixed point), every cell
    __shared__ u64 sL[STRIP_RING * P];                 // ... land-side cells
    __shared__ unsigned short sC[STRIP_RING * P];      // ... land-side count (modulo 2^16: a window holds < 2^16 cells)
    __shared__ u64 s_land[STRIP_RING];                 // land-side bits of every ring row
    __shared__ u64 s_bits[STRIP_MAXW];                 // the active blocks as a bit plane
    __shared__ uint2 s_ent[STRIP_SCHED];               // steps of the round: x = position | flags, y = strip << 16 | block
                                                       // within the padded strip
    __shared__ unsigned short s_cell[3][SW * C];       // the band cells of the block a step queries, compacted (three steps in flight)
    __shared__ Moments s_wpart[NWV];
    __shared__ int s_scan[NWV];
    __shared__ int s_misc[12];                         // [0] steps of the round, [1..3] entries of the three cell lists, [4] a cell was marked,
                                                       // [5], [6] the share (ranks of active blocks), [7] totals of the plane,
                                                       // [8] the plan of this call is stored (incl. its cell lists),
                                                       // [9] query steps of the plan (a band step's update)
    __shared__ T s_sdr[2];
    static_assert(sizeof(u64) * (2 * STRIP_RING * P + STRIP_RING + STRIP_MAXW) + 2 * STRIP_RING * P + 8 * STRIP_SCHED +
                          6 * SW * C + sizeof(Moments) * NWV + 4 * NWV + 48 + 16 <= 160 * 1024,
                  "k_strip: LDS of one workgroup");

    const Geo g = job.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int npad = job.nty + 2;                      // blocks of a strip incl. the virtual ones above and below
    const int npos = job.ntx * npad;
    const unsigned npad_magic = 0xffffffffu / (unsigned)npad + 1u;       // floor(p / npad) = umulhi(p, magic) for p < 2^16

    SB_T(0);                                             // start
    // ---- prologue 1: the flags k_scan raised, 1024 at a time, as a bit plane (word c NWV + wv = ballot of chunk c) ----
    T sd = T(0), rr = T(0);
    Moments pm = moments_empty();
    double shift_c = 0.0;
    const bool fold_stats = job.fold && job.fold_nparts > 0;
    if (fold_stats) {
        if (tid < job.fold_nparts) pm = fold_partials[tid];
        shift_c = (double)job.sigma[(size_t)g.h * g.nxh + g.h];
    } else if (FLY && job.ngath > 0) {
        // band step: the first wave merges the moments gathered from all ranks in rank order (one tree on every
        // workgroup of every rank: identical scalars everywhere); workgroup 0 publishes them
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
    } else if (FLY) { sd = job.stats[0]; rr = job.stats[1]; }
    const int nch = (npos + STRIP_NT - 1) / STRIP_NT;            // <= STRIP_MAXW / 16 (host)
    const int nwords = nch * NWV;
    // ---- the plan.  Which blocks this workgroup marches over, in which order, and where their band cells lie follows
    // from the band plane alone, and a coast does not move: the plan is stored in device memory (steps and cell
    // lists), and k_scan -- which rewrites the plane every call -- compares each word with the one it replaces and
    // leaves the number of the last call that saw a difference.  A plan stored by that call or a later one is used as
    // it is: no flags are read, nothing is planned, no list is built.
    typedef const __attribute__((address_space(4))) int *cintp;
    char *const plan_wg = plan + (size_t)blockIdx.x * SB_PLAN_STRIDE;
    unsigned *const plan_lists = (unsigned *)(plan_wg + SB_PLAN_LIST_OFF);
    // (the stored plan's steps travel WITH its header -- one round trip, not two; used only if the plan stands)
    const uint2 plan_ent = ((const uint2 *)(plan_wg + SB_PLAN_ENT_OFF))[tid < STRIP_SCHED ? tid : 0];
    const int plan_stored = ((cintp)plan_wg)[0], plan_nst = min(((cintp)plan_wg)[1], STRIP_SCHED);
    const int plan_rb = ((cintp)plan_wg)[2], plan_re = ((cintp)plan_wg)[3];
    const bool cached = job.plan_use != 0 && plan_stored != 0 && *(cintp)plan_gen <= plan_stored;      // uniform
    auto load_plane = [&]() {
        u64 mine = 0;
        for (int base = 0; base < nch; base += 8) {              // 8 loads in flight (clamped, so none is conditional)
            int f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = (base + j) * STRIP_NT + tid;
                f[j] = job.flags[i < npos ? i : npos - 1];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = (base + j) * STRIP_NT + tid;
                mine |= (i < npos && f[j] != 0) ? 1ull << (base + j) : 0ull;
            }
        }
        for (int c = 0; c < nch; ++c) {
            const u64 b = __builtin_amdgcn_ballot_w64((mine >> c) & 1ull);
            if (lane == 0) s_bits[c * NWV + wv] = b;
        }
    };
    if (__builtin_expect(!cached, 0)) load_plane();
    else {
        if (tid < plan_nst) s_ent[tid] = plan_ent;      // (at most STRIP_SCHED < 1024 steps)
        if (tid == 0) { s_misc[0] = plan_nst; s_misc[5] = plan_rb; s_misc[6] = plan_re; s_misc[8] = 0; }
    }
    // the zero column of the three tables (never written again) while the flags travel
    for (int i = tid; i < STRIP_RING; i += STRIP_NT) { sA[i * P] = 0; sL[i * P] = 0; sC[i * P] = 0; }
    if (tid == 0) s_misc[4] = 0;
    if (fold_stats) wave_total_shifted_store(pm, s_wpart);
    SB_T(1);                                             // first barrier reached
    __syncthreads();
    SB_T(2);                                             // ... passed
    // ---- prologue 2, WAVE 0 ALONE (the others wait at one barrier): this workgroup's share and the schedule of its first
    // round.  Shares are equal in COST, in strip-major order.  The marks of tools/stamp_strip.py, fitted over the 256
    // workgroups of the headline grid, give a workgroup's life as 1.25 us per staged block (an active block or a
    // neighbour of one) + 0.56 us per active block (its band cells are queried) + 0.68 us per run (drain step, restart):
    // weights 4 : 2 : 2.  Workgroup b takes the active blocks at which the running cost lies in [b, b + 1) T / G.  (An
    // equal share of active blocks left the workgroup with the most short runs with 15 staged blocks against a mean of
    // 10; an equal share of staged blocks still had lives of 20 .. 30 us around a mean of 25.)
    // Per word of the plane: active and staged blocks before it, packed (low / high 16 bits) -- the array lies where
    // the cell lists of the march will (they are not in use while a round is planned) -- and the cost before it, where
    // the first rows of the first table will (their zero column is written again behind the planning).
    unsigned *s_pre = (unsigned *)&s_cell[0][0];
    unsigned *s_cost = (unsigned *)&sA[0];
    constexpr int CW_STAGED = 4, CW_ACTIVE = 2, CW_RUN = 2;
    static_assert(sizeof(s_cell) >= sizeof(unsigned) * STRIP_MAXW, "the prefix array fits where the cell lists lie");
    auto wave_sync = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };      // this wave's LDS writes have landed
    auto stage_word = [&](int k) -> u64 {                // staged = act | act << 1 | act >> 1 (virtual blocks separate the strips)
        const u64 a = s_bits[k], pv = k > 0 ? s_bits[k - 1] : 0ull, nx = k + 1 < nwords ? s_bits[k + 1] : 0ull;
        return a | (a << 1) | (pv >> 63) | (a >> 1) | (nx << 63);
    };
    auto run_starts = [&](int k, u64 sw) -> u64 {        // staged blocks of word k whose predecessor is not staged
        const u64 swp = k > 0 ? stage_word(k - 1) : 0ull;
        return sw & ~((sw << 1) | (swp >> 63));
    };
    int tot_packed = 0, tot_cost = 0;
    auto make_prefix = [&](bool with_cost) {             // one wave
        int run = 0, crun = 0;
        for (int k0 = 0; k0 < nwords; k0 += SB_WAVE) {
            const int k = k0 + lane;
            const u64 a = k < nwords ? s_bits[k] : 0ull, sw = k < nwords ? stage_word(k) : 0ull;
            const int v = (int)((unsigned)__popcll(a) | (unsigned)__popcll(sw) << 16);
            const int incl = sb_wave_scan_add(v);
            if (k < nwords) s_pre[k] = (unsigned)(run + incl - v);
            run += __builtin_amdgcn_readlane(incl, SB_WAVE - 1);
            if (with_cost) {
                const int c = CW_STAGED * __popcll(sw) + CW_ACTIVE * __popcll(a) + CW_RUN * __popcll(k < nwords ? run_starts(k, sw) : 0ull);
                const int cincl = sb_wave_scan_add(c);
                if (k < nwords) s_cost[k] = (unsigned)(crun + cincl - c);
                crun += __builtin_amdgcn_readlane(cincl, SB_WAVE - 1);
            }
        }
        tot_packed = run;
        tot_cost = crun;
        if (lane == 0) s_misc[7] = run;
        wave_sync();
    };
    // the word that holds rank t of the packed prefix (hi: staged, else active) and the rank inside it; wave-uniform
    auto find_word = [&](int t, bool hi, int &n) -> int {
        int kk = -1;
        n = 0;
        for (int k0 = 0; k0 < nwords; k0 += SB_WAVE) {
            const int k = k0 + lane;
            const unsigned pa = k < nwords ? s_pre[k] : 0u, pb = k + 1 < nwords ? s_pre[k + 1] : (unsigned)tot_packed;
            const int lo = (int)(hi ? pa >> 16 : pa & 0xffffu), up = (int)(hi ? pb >> 16 : pb & 0xffffu);
            const u64 hit = __builtin_amdgcn_ballot_w64(k < nwords && lo <= t && t < up);
            if (hit) {
                const int src = __ffsll((unsigned long long)hit) - 1;
                kk = k0 + src;
                n = t - __builtin_amdgcn_readlane(lo, src);
                break;
            }
        }
        return kk;
    };
    auto nth_bit = [&](u64 word, int n) -> int {         // position of the n-th set bit (lane j looks at bit j)
        const bool me = ((word >> lane) & 1ull) && __popcll(word & ((1ull << lane) - 1ull)) == n;
        return __ffsll((unsigned long long)__builtin_amdgcn_ballot_w64(me)) - 1;
    };
    // active blocks in front of the position at which the running cost reaches t (all of them beyond the total)
    auto act_before_cost = [&](int t) -> int {
        const int nact = tot_packed & 0xffff;
        if (t >= tot_cost) return nact;
        for (int k0 = 0; k0 < nwords; k0 += SB_WAVE) {
            const int k = k0 + lane;
            const int lo = k < nwords ? (int)s_cost[k] : 0x7fffffff, up = k + 1 < nwords ? (int)s_cost[k + 1] : tot_cost;
            const u64 hit = __builtin_amdgcn_ballot_w64(k < nwords && lo <= t && t < up);
            if (hit) {                                   // wave-uniform: the word in which the cost crosses t
                const int src = __ffsll((unsigned long long)hit) - 1, kk = k0 + src;
                const u64 a = sb_uniform64(s_bits[kk]), sw = sb_uniform64(stage_word(kk)), rs = sb_uniform64(run_starts(kk, stage_word(kk)));
                const u64 upto = lane == 63 ? ~0ull : (1ull << (lane + 1)) - 1ull;       // bits 0 .. lane
                const int cum = __builtin_amdgcn_readlane(lo, src) + CW_STAGED * __popcll(sw & upto) + CW_ACTIVE * __popcll(a & upto) + CW_RUN * __popcll(rs & upto);
                const u64 over = __builtin_amdgcn_ballot_w64(cum > t);
                const int bpos = over ? __ffsll((unsigned long long)over) - 1 : 63;       // the first position behind the crossing
                return (int)(__builtin_amdgcn_readfirstlane((int)s_pre[kk]) & 0xffff) + __popcll(a & ((1ull << bpos) - 1ull));
            }
        }
        return nact;
    };
    // The schedule of the round that holds the active blocks of ranks [ra, rb): the staged positions in ascending
    // order with their flags, a drain step behind every run, three warm-up steps in front (they stage nothing and only
    // issue the loads of the first three blocks, so that every load of the march is issued at the same three program
    // points -- see `step`), padded to a multiple of three.  One lane per position, 64 positions at a time.
    auto make_schedule = [&](int ra, int rb) {           // one wave; leaves the number of steps in s_misc[0]
        int n0, n1;
        const int k0w = find_word(ra, false, n0), k1w = find_word(rb - 1, false, n1);
        const int p0 = k0w < 0 ? -1 : k0w * 64 + nth_bit(sb_uniform64(s_bits[k0w < 0 ? 0 : k0w]), n0);
        const int p1 = k1w < 0 ? -1 : k1w * 64 + nth_bit(sb_uniform64(s_bits[k1w < 0 ? 0 : k1w]), n1);
        wave_sync();                                     // (the prefix array may be overwritten from here on)
        if (p0 < 1 || p1 < p0) { if (lane == 0) s_misc[0] = 0; return; }     // (cannot happen: position 0 is virtual)
        if (lane < STRIP_DEPTH) s_ent[lane] = make_uint2(SCH_DRAIN | SCH_IDLE, 0u);
        if (lane < 3) s_misc[1 + lane] = 0;
        int n_out = STRIP_DEPTH;
        for (int c = p0 - 1; c <= p1 + 1; c += SB_WAVE) {
            const int pp = c + lane;
            // active blocks (of this round) at positions pp - 2 .. pp + 2: bits 0 .. 4
            unsigned win = 0;
#pragma unroll
            for (int d = 0; d < 5; ++d) {
                const int q = pp + d - 2;
                const bool in = q >= p0 && q <= p1;
                const u64 w = s_bits[in ? q >> 6 : 0];
                win |= (in && ((w >> (q & 63)) & 1ull)) ? 1u << d : 0u;
            }
            const bool st = pp <= p1 + 1 && (win & 0xeu) != 0u;                   // pp - 1, pp, pp + 1
            // (a run never crosses from one strip into the next: the virtual block below a strip ends it, the one above the
            // next strip starts afresh -- the blocks of a run are queried by their position within ONE strip; found in
            // round 4 on a grid whose band reaches the first and the last row)
            const int sp = (int)__umulhi((unsigned)(pp < 0 ? 0 : pp), npad_magic), jpp = pp - sp * npad;
            const bool st_prev = (win & 0x7u) != 0u && jpp != 0, st_next = (win & 0x1cu) != 0u && jpp != npad - 1;
            const bool en = st && !st_next;
            const u64 ms = __builtin_amdgcn_ballot_w64(st), me = __builtin_amdgcn_ballot_w64(en);
            const u64 below = (1ull << lane) - 1ull;
            const int at = n_out + __popcll(ms & below) + __popcll(me & below);
            n_out += __popcll(ms) + __popcll(me);
            if (st) {
                const unsigned sjv = ((unsigned)sp << 16) | (unsigned)jpp;
                const unsigned e = (unsigned)pp | ((win & 1u) ? SCH_Q2 : 0u) | (st_prev ? 0u : SCH_RESTART);
                if (at < STRIP_SCHED) s_ent[at] = make_uint2(e, sjv);
                if (en && at + 1 < STRIP_SCHED) s_ent[at + 1] = make_uint2((unsigned)pp | SCH_DRAIN | ((win & 2u) ? SCH_Q1 : 0u), sjv);
            }
        }
        // (padded to a multiple of three with steps that do nothing: the march has no early exit -- with one, the
        // compiler's count of the loads in flight collapses and it drains the queue every third step)
        // (The padding steps' dummy loads go where the last real step's went -- same strip, same block: no column
        // arithmetic, lines that are in the cache.  A padding step cost 0.45 us, and the longest-lived workgroups of
        // the headline grid have two.)
        n_out = min(n_out, STRIP_SCHED - 2);
        const int n_pad = (n_out + STRIP_DEPTH - 1) / STRIP_DEPTH * STRIP_DEPTH;
        wave_sync();
        const unsigned last_sj = s_ent[n_out - 1].y;         // (n_out >= STRIP_DEPTH + 1 here)
        if (lane < n_pad - n_out) s_ent[n_out + lane] = make_uint2(SCH_DRAIN | SCH_IDLE, last_sj);
        if (lane == 0) s_misc[0] = n_pad;
    };
    if (__builtin_expect(!cached, 0)) {
        if (wv == 0) {
            make_prefix(true);
            const int rb0 = act_before_cost((int)(((long long)blockIdx.x * tot_cost) / G));
            const int re0 = blockIdx.x + 1 == (unsigned)G ? (tot_packed & 0xffff) : act_before_cost((int)(((long long)(blockIdx.x + 1) * tot_cost) / G));
            if (lane == 0) { s_misc[5] = rb0; s_misc[6] = re0; s_misc[0] = 0; }
            if (rb0 < re0) make_schedule(rb0, min(rb0 + STRIP_ROUND, re0));
            wave_sync();
            // the plan goes to device memory: the steps, each query step with the number of its cell list (the lists
            // themselves are written by the waves that query them); a share of several rounds, or of more query steps
            // than a stored plan holds, is planned every call
            const int nstv = __builtin_amdgcn_readfirstlane(s_misc[0]);
            int nq = 0;
            uint2 *eg = (uint2 *)(plan_wg + SB_PLAN_ENT_OFF);
            for (int c0 = 0; c0 < nstv; c0 += SB_WAVE) {
                const int i = c0 + lane;
                uint2 v = s_ent[i < nstv ? i : 0];
                const bool q = i < nstv && !(v.x & SCH_IDLE) && (v.x & ((v.x & SCH_DRAIN) ? SCH_Q1 : SCH_Q2)) != 0u;
                const u64 m = __builtin_amdgcn_ballot_w64(q);
                const int qi = nq + __popcll(m & ((1ull << lane) - 1ull));
                nq += __popcll(m);
                if (q) v.x |= (unsigned)(qi & (SB_PLAN_NQ - 1)) << SCH_QI_SHIFT;
                if (i < nstv) { s_ent[i] = v; eg[i] = v; }
            }
            const bool ok = re0 - rb0 <= STRIP_ROUND && nq <= SB_PLAN_NQ;
            if (lane == 0) {
                int *h = (int *)plan_wg;
                h[1] = nstv; h[2] = rb0; h[3] = re0;
                h[0] = ok ? job.call_id : 0;
                s_misc[8] = ok ? 1 : 0;
            }
        }
        __syncthreads();
        if (tid < 8) sA[tid * P] = 0;                    // (the cost array lay over the first rows' zero column)
        SB_T(3);                                         // planned
    }
    const bool store_lists = __builtin_amdgcn_readfirstlane(s_misc[8]) != 0;
    // the cell list of the next step's query, from the stored plan (one entry per lane of the eight querying waves)
    unsigned qc = ~0u;
    const unsigned qc_off = (unsigned)(min(wv, C / 2 - 1) * SB_WAVE + lane);
    const int r_begin = __builtin_amdgcn_readfirstlane(s_misc[5]), r_end = __builtin_amdgcn_readfirstlane(s_misc[6]);

    const bool fastx = g.nx > W + 2;                   // one conditional add wraps every column of a staged row
    const bool limited = g.bnd == BND_HALO;
    // how far a window round (x, y) may reach inside a ghost-celled frame: the ghost width beyond the interior -- except
    // in the directions in which a band's frame is not an edge at all (round the circle; beyond a pole)
    const int big_reach = 1 << 20;
    auto frame_reach = [&](int x, int y) __attribute__((always_inline)) -> int {
        const int rx = (g.band & GEO_BAND_EW) ? big_reach : min(x + g.h, g.nx - 1 - x + g.h);
        const int rs = (g.band & GEO_BAND_SOUTH) ? big_reach : y + g.h, rn = (g.band & GEO_BAND_NORTH) ? big_reach : g.ny - 1 - y + g.h;
        return min(rx, min(rs, rn));
    };

    // the lane's column of the strip the loads are issued for: byte offsets in a field row and in a row of the
    // land-side plane, bit in the 32-bit word (-1: no such cell); recomputed when the strip changes
    int cc_strip = -1;
    unsigned cc_colb = 0, cc_clsb = 0, cc_lbit = 0;

    // loads of row wv of block jp of `strip`
    // (always four loads, also behind the end of the schedule and for a drain step, from clamped addresses: the
    // compiler counts the loads in flight per program point, and a path without them would make it drain the queue)
    auto issue = [&](StripRegs<T, FLY> &R, unsigned sj) __attribute__((always_inline)) {
        const int strip = (int)(sj >> 16), jp = (int)(sj & 0xffffu);
        if (strip != cc_strip) {                         // wave-uniform
            cc_strip = strip;
            const int xs = strip * SW - H + lane;
            bool ok = true;
            int Xc = 0;
            if (g.bnd == BND_HALO) {
                int xw = xs;
                if (g.band & GEO_BAND_EW) xw = xs < 0 ? xs + g.nx : (xs >= g.nx ? xs - g.nx : xs);   // (a band holds whole circles; nx > 64 + 2)
                Xc = xw + g.h; ok = Xc >= 0 && Xc < g.nxh;
            }
            else if (fastx) {
                if (g.bnd == BND_WRAPPER) {
                    int m = xs + 1;
                    m = m < 0 ? m + g.nx : (m >= g.nx ? m - g.nx : m);
                    Xc = (m < 1 ? 1 : m) - 1;
                } else Xc = xs < 0 ? xs + g.nx : (xs >= g.nx ? xs - g.nx : xs);
            } else {
                int Yd;
                sb_map_cell(g, xs, 0, Xc, Yd);
            }
            const unsigned xc = ok ? (unsigned)Xc : 0u;  // every load is unconditional, from a clamped address
            cc_colb = xc * (unsigned)sizeof(T);
            cc_clsb = (xc >> 5) * 4u;
            cc_lbit = ok ? 1u << (xc & 31u) : 0u;
        }
        const int ys = (jp - 1) * C + wv;               // interior row (may lie outside the grid: clamped or absent)
        int Yr;
        bool rowok = true;
        if (g.bnd == BND_HALO) {
            int yw = ys;
            if ((g.band & GEO_BAND_SOUTH) && yw < 0) yw = 0;          // beyond a pole: the edge row again (the latitude clamp)
            if ((g.band & GEO_BAND_NORTH) && yw >= g.ny) yw = g.ny - 1;
            Yr = yw + g.h; rowok = Yr >= 0 && Yr < g.nyh; Yr = rowok ? Yr : 0;
        }
        else Yr = ys < 0 ? 0 : (ys >= g.ny ? g.ny - 1 : ys);
        // (a scalar base -- field pointer plus the row's offset -- and the lane's 32-bit column offset: two scalar
        // registers per field where a buffer descriptor takes four)
        const size_t rowb = (size_t)((unsigned)Yr * (unsigned)g.nxh) * sizeof(T), wordb = (size_t)((unsigned)Yr * (unsigned)g.nw) * 8u;
        R.th = *(const T *)((const char *)job.theta + rowb + cc_colb);                 // (theta is the t0 plane unless FLY)
        if constexpr (FLY) {
            R.zz = *(const T *)((const char *)job.z + rowb + cc_colb);
            R.sg = *(const T *)((const char *)job.sigma + rowb + cc_colb);
        }
        R.lw = *(const uint32_t *)((const char *)job.clsbits + wordb + cc_clsb);
        R.lbit = rowok ? cc_lbit : 0u;
    };

    // running column totals of the table this wave sums along latitude (waves 5, 6: 64 bit; wave 7: the count)
    u64 carry = 0;

    // S1: the row's registers -> ring row (prefix along longitude only)
    auto stage = [&](StripRegs<T, FLY> &R, unsigned ent, int jp) __attribute__((always_inline)) {
        sb_cdp kt = (sb_cdp)sb_strip_k;
        asm volatile("" : "+s"(kt));                      // (opaque: with STRIP_KTAB the constants are loaded here, every time)
        // This block's loads were issued three steps ago, and the two steps since have each issued a block's loads of their
        // own (`issue` is unconditional) besides lists and stores: once at most two blocks' loads are outstanding -- the
        // counter is in order -- this block's have landed.  Said explicitly, tied to the registers: hipcc's own count of the
        // loads in flight has been seen to go wrong at this very place (see the step), and a wait that is missing here
        // goes unnoticed by every test, three steps being longer than the latency of memory.
        if constexpr (FLY) asm volatile("s_waitcnt vmcnt(8)" : "+v"(R.th), "+v"(R.zz), "+v"(R.sg), "+v"(R.lw) : : "memory");
        else asm volatile("s_waitcnt vmcnt(4)" : "+v"(R.th), "+v"(R.lw) : : "memory");
        // Predicates as one compare each, straight into a lane mask (an `a && b` of two of them costs two more vector
        // instructions to re-form the mask): the land-side bit is tested against a per-lane bit that is zero where the
        // cell does not exist.
        const u64 lm = __builtin_amdgcn_ballot_w64((R.lw & R.lbit) != 0u);
        T t0v = R.th;
        if constexpr (FLY) {
            // the sigmoid only where a lane of the wave stands above sea level (z == 0 -> t0 = theta exactly)   ref :166-167
            if (__builtin_amdgcn_ballot_w64(R.zz != T(0)) != 0) t0v = strip_t0(R.th, R.zz, R.sg, sd, rr, kt);
        }
        u64 qa = sb_to_fixed((double)t0v, kt);
        if (limited) { if (R.lbit == 0u) qa = 0ull; }    // (ghost-celled frames only: columns and rows beyond the frame)
        u64 ql = ((R.lw & R.lbit) != 0u) ? qa : 0ull;
        // (a row on one side of the coast needs one scan: its land-side sums are zero, or the sums over all cells)
        if (lm == 0ull) sb_scan1_u64(qa);                // wave-uniform; ql is zero everywhere
        else if (lm == ~0ull) { sb_scan1_u64(qa); ql = qa; }
        else sb_scan2_u64(qa, ql);
        const unsigned slot = (unsigned)(jp * C + wv) & RM;
        const unsigned o = __umul24(slot, P) + lane + 1;
        sA[o] = qa;
        sL[o] = ql;
        if (__builtin_expect(!cached, 0)) {              // (a stored plan knows every window's radius, count and class)
            // land-side cells up to and including the lane's: those of lanes 1 .. lane by mbcnt on the mask shifted down, lane 0's added
            const u64 lm1 = lm >> 1;
            const unsigned cnt = __builtin_amdgcn_mbcnt_hi((unsigned)(lm1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)lm1, (unsigned)(lm & 1ull)));
            sC[o] = (unsigned short)cnt;
            if (lane == 0) s_land[slot] = lm;
        }
        if (ent & SCH_RESTART) {
            // the tables start afresh: the row above the first one reads as zero, the running totals start at zero
            if (wv == NWV - 1) {
                const unsigned z = __umul24((unsigned)(jp * C - 1) & RM, P) + lane + 1;
                sA[z] = 0; sL[z] = 0; sC[z] = 0;
            }
            carry = 0;
        }
    };

    // S2, waves 5-7: prefix along latitude of the 16 rows of the block at ring position jp
    auto vertical = [&](int jp) __attribute__((always_inline)) {
        const unsigned r0 = (unsigned)(jp * C) & RM;     // a block never straddles the end of the ring (128 = 8 x 16)
        const unsigned o = __umul24(r0, P) + lane + 1;
        if (wv < 7) {
            u64 *tab = (wv == 5 ? sA : sL) + o;
#pragma unroll
            for (int h0 = 0; h0 < C; h0 += 8) {          // eight rows of reads in flight
                u64 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = tab[(h0 + i) * P];
#pragma unroll
                for (int i = 0; i < 8; ++i) { carry += v[i]; tab[(h0 + i) * P] = carry; }
            }
        } else if (!cached) {
            unsigned short *tab = sC + o;
            unsigned v[C];
#pragma unroll
            for (int i = 0; i < C; ++i) v[i] = tab[i * P];
            unsigned cc = (unsigned)carry;
#pragma unroll
            for (int i = 0; i < C; ++i) { cc += v[i]; tab[i * P] = (unsigned short)cc; }
            carry = cc;
        }
    };

    // The band bits of the two rows this wave lists in block jp of `strip`, as SCALAR loads (constant address space: the
    // plane is k_scan's, read-only here).  Scalar loads count in lgkmcnt, not in the in-order vmcnt queue of the
    // prefetched blocks: a vector load here would sit between them, and the wait for it would drain every load
    // issued before it.
    struct BandWords { u64 a0, b0, a1, b1, c0, c1; int sh; };   // c: land-side word of the last longitude (f2py rule)
    auto band_issue = [&](int strip, int jp) __attribute__((always_inline)) -> BandWords {
        const int k = max(wv - C / 2, 0);                // the listing waves are 8 .. 15: two rows each
        const int y0 = (jp - 1) * C + 2 * k;
        const int ya = min(max(y0, 0), g.ny - 1), yb = min(max(y0 + 1, 0), g.ny - 1);
        const int xa = strip * SW + g.h;                 // array column of the strip's first owned cell
        const int wlo = xa >> 6, whi = min(wlo + 1, g.nw - 1);
        cu64p bits = (cu64p)job.bandbits;
        BandWords w;
        w.a0 = bits[(size_t)(ya + g.h) * g.nw + wlo]; w.b0 = bits[(size_t)(ya + g.h) * g.nw + whi];
        w.a1 = bits[(size_t)(yb + g.h) * g.nw + wlo]; w.b1 = bits[(size_t)(yb + g.h) * g.nw + whi];
        w.sh = xa & 63;
        w.c0 = w.c1 = 0;
        if (g.bnd == BND_WRAPPER && strip == job.ntx - 1) {      // uniform; the strip that owns longitude nx
            cu64p cls = (cu64p)job.clsbits;
            const int wl = (g.nx - 1 + g.h) >> 6;
            w.c0 = cls[(size_t)(ya + g.h) * g.nw + wl]; w.c1 = cls[(size_t)(yb + g.h) * g.nw + wl];
        }
        return w;
    };

    // S1, waves 8-15: the band cells of two rows of the queried block -> the step's compact list.  A cell's code is
    // row << 5 | column, plus (f2py rule, last longitude only) bit 10 and in bit 9 its own land-side bit -- see `query`.
    // The waves reserve their entries with one LDS atomic each: the order of the list is of no consequence.
    const unsigned cell_code = (unsigned)((2 * max(wv - C / 2, 0) + (lane >> 5)) << 5 | (lane & (SW - 1)));   // row << 5 | column
    auto list_cells = [&](int strip, int jp, const BandWords &bwd, int buf) __attribute__((always_inline)) {
        // the wave's 64 band bits (lanes 0-31: first row, 32-63: second) by scalar funnel shifts of the four words
        const int y0 = (jp - 1) * C + 2 * max(wv - C / 2, 0);
        const int ncol = min(g.nx - strip * SW, SW);      // owned columns that exist (the last strip may be cut)
        const unsigned colmask = ncol >= 32 ? 0xffffffffu : (1u << ncol) - 1u;
        auto row_bits = [&](u64 a, u64 b2, int y) -> unsigned {
            const u64 f = bwd.sh ? (a >> bwd.sh) | (b2 << (64 - bwd.sh)) : a;
            return (y >= 0 && y < g.rows) ? (unsigned)f & colmask : 0u;
        };
        const u64 m = (u64)row_bits(bwd.a0, bwd.b0, y0) | (u64)row_bits(bwd.a1, bwd.b1, y0 + 1) << 32;
        if (m == 0) return;                              // wave-uniform
        unsigned code = cell_code;
        if (g.bnd == BND_WRAPPER && strip == job.ntx - 1) {      // uniform: the strip that owns the last longitude
            if (strip * SW + (int)(lane & (SW - 1)) == g.nx - 1) {
                const unsigned sl = (unsigned)((g.nx - 1 + g.h) & 63);
                code |= 1u << 10 | (unsigned)(((lane >> 5) ? (bwd.c1 >> sl) : (bwd.c0 >> sl)) & 1ull) << 9;
            }
        }
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_misc[1 + buf], __popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        const unsigned at = (unsigned)base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if ((m >> lane) & 1ull) s_cell[buf][at] = (unsigned short)code;
    };

    // S2, waves 0 ..: 64 entries of the list per wave: bisection for the radius, contrast, result
    auto query = [&](int qpos, int strip, int jp, int buf, unsigned qi) __attribute__((always_inline)) {
        // A cell's entry in the stored plan: bits 0-8 row << 5 | column, bit 9 its own class, bits 10-14 the radius of its
        // window (0: it outgrows the tables), bits 15-25 the land-side cells in it; all ones: no cell.  Radius, count
        // and class follow from the land-side plane, which stands as long as the plan does (k_scan watches it too).
        unsigned code;
        bool valid, found, own;
        int nn, nl;
        if (cached) {                                    // uniform: the list of the stored plan (loaded one step ahead)
            code = qc;
            valid = code != ~0u;
            if (__builtin_amdgcn_ballot_w64(valid) == 0ull) return;
            nn = (int)((code >> 10) & 31u);
            nl = (int)((code >> 15) & 2047u);
            own = ((code >> 9) & 1u) != 0u;
            found = valid && nn != 0;
            nn = max(nn, 1);                             // (reads in bounds; the result is not used)
        }
        if (__builtin_expect(!cached, 0)) {
            const int ncell = __builtin_amdgcn_readfirstlane(s_misc[1 + buf]);
            const int e = wv * SB_WAVE + lane;
            valid = e < ncell;
            code = s_cell[buf][valid ? e : 0];
            if (wv * SB_WAVE >= ncell) {                 // wave-uniform
                if (store_lists) plan_lists[qi * (unsigned)(SW * C) + (unsigned)e] = ~0u;
                return;
            }
        }
        const int lx = (int)(code & 31u), ly = (int)((code >> 5) & 15u);
        const int x = strip * SW + lx, y = (jp - 1) * C + ly;
        const unsigned o = (unsigned)y * (unsigned)g.nx + (unsigned)x;     // (fewer than 2^31 cells: check_dims)
        const unsigned rho = (unsigned)(jp * C + ly);    // ring row of the cell
        const unsigned cx = (unsigned)(lx + H + 1);      // its table column
        if (__builtin_expect(!cached, 0)) {
            int lim = H;
            if (limited) lim = min(lim, frame_reach(x, y));   // uniform branch
            const int limc = max(lim, 1);
            // land-side count of the square of radius rad: C(r1,a1) - C(r0,a1) - C(r1,a0) + C(r0,a0),
            // r0 = rho-rad-1, r1 = rho+rad, a0 = cx-rad-1, a1 = cx+rad
            auto count = [&](int rad) __attribute__((always_inline)) {
                const unsigned r1 = __umul24((rho + rad) & RM, P) + cx, r0 = __umul24((rho - rad - 1) & RM, P) + cx;
                return (int)(unsigned short)((unsigned)sC[r1 + rad] - (unsigned)sC[r0 + rad] - (unsigned)sC[r1 - rad - 1] + (unsigned)sC[r0 - rad - 1]);
            };
            // Two rounds of independent probes -- radii 4, 8, 12, 16, then the three radii below the smallest of those that
            // holds both classes: 28 reads in two LDS round trips, where a bisection makes 20 reads in five.  The march is
            // bound by the length of its dependent chains, not by LDS issue.
            int nl1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) nl1[k] = count(limited ? min(4 * (k + 1), limc) : 4 * (k + 1));
            int lo = 1, hi = limc;
            nl = 0;
            bool got = false;
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                const int rad = limited ? min(4 * (k + 1), limc) : 4 * (k + 1);
                const bool mixed = nl1[k] > 0 && nl1[k] < (2 * rad + 1) * (2 * rad + 1);
                if (mixed) { hi = rad; nl = nl1[k]; got = true; }
                else if (rad < hi) lo = max(lo, rad + 1);
            }
            found = valid && lim >= 1 && got;
            if (!got) lo = max(1, hi - 3);                   // (probes in bounds; their results are not used)
            int nl2[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) nl2[m] = count(min(lo + m, hi));
#pragma unroll
            for (int m = 2; m >= 0; --m) {
                const int rad = lo + m;
                if (rad < hi && nl2[m] > 0 && nl2[m] < (2 * rad + 1) * (2 * rad + 1)) { hi = rad; nl = nl2[m]; }
            }
            nn = hi;
            // the cell's own class: the table's centre, except that the f2py boundary rule maps the centre of the window at
            // the last longitude to column 1 (there the list entry carries the bit)   ref :182-186, seabreeze_diag_python.f90:202
            const u64 ownw = s_land[rho & RM];
            own = (code >> 10) & 1u ? ((code >> 9) & 1u) != 0u : ((ownw >> (lx + H)) & 1ull) != 0ull;
            if (store_lists)
                plan_lists[qi * (unsigned)(SW * C) + (unsigned)(wv * SB_WAVE + lane)] =
                    valid ? (code & 511u) | (own ? 1u << 9 : 0u) | (found ? (unsigned)nn << 10 : 0u) | (unsigned)nl << 15 : ~0u;
        }
        const int area = (2 * nn + 1) * (2 * nn + 1);
        const unsigned r1 = __umul24((rho + nn) & RM, P) + cx, r0 = __umul24((rho - nn - 1) & RM, P) + cx;
        const u64 l11 = sL[r1 + nn], l01 = sL[r0 + nn], l10 = sL[r1 - nn - 1], l00 = sL[r0 - nn - 1];
        const u64 q11 = sA[r1 + nn], q01 = sA[r0 + nn], q10 = sA[r1 - nn - 1], q00 = sA[r0 - nn - 1];
        // exact: the tables wrap, the window sum does not; every cell of the window carries the fixed-point bias
        const long long RL = (long long)((l11 - l01) - (l10 - l00) - (u64)nl * SB_FIX_BIAS);
        const long long RS = (long long)((q11 - q01) - (q10 - q00) - (u64)area * SB_FIX_BIAS) - RL;      // sea side
        // land mean - sea mean = (RL ns - RS nl) / (nl ns): one reciprocal of an exact small integer (v_rcp_f64 + two
        // Newton steps: within an ulp of the quotient)
        auto to_f64 = [](long long v) { return __builtin_fma((double)(int)(v >> 32), 0x1p32, (double)(unsigned)v); };
        const double dnl = (double)nl, dns = (double)(area - nl);
        const double num = to_f64(RL) * dns - to_f64(RS) * dnl;
        const T contrast = (T)(num * sb_inv(dnl * dns) * 0x1p-40);
        const T mul = own ? T(1) : T(-1);
        int nnmax = 0;
        if (found) { nnmax = nn; job.thc[o] = mul * contrast; }              // ref :216; k_wind applies :235-266
        // cells whose window outgrows the tables: marked, handled behind the march
        if (valid && !found) { job.thc[o] = strip_mark<T>(); s_misc[4] = 1; }
        // per-block largest radius (diagnostic, read by sb_last_counters); the flag k_scan raised is 1
        nnmax = sb_wave_max_to_last(nnmax);
        if (lane == SB_WAVE - 1 && nnmax > 1) atomicMax(&job.flags[qpos], nnmax);
    };

    // k_scan's shifted sums added up in k_prep's order and turned into the sigmoid scalars -- two divisions and a square
    // root in fp64, some 200 dependent instructions -- by the FIRST wave alone (the oldest wave of its SIMD has priority
    // at issue: it comes through the code in front of the march in half the time the last one takes, and everybody
    // waits for these scalars); all pick them up behind the barrier that opens the first run.  Workgroup 0 publishes
    // them (for the calls that reuse them: static sigma), with or without a share of the march.
    auto finish_stats = [&]() {
        if (wv == 0) {
            const Moments m = moments_of_shifted(shift_c, block_total_shifted_finish<NWV>(s_wpart));
            T st4[4];
            sigmoid_scalars<T>(m, st4);
            if (lane == 0) {
                s_sdr[0] = st4[0]; s_sdr[1] = st4[1];
                if (blockIdx.x == 0) { for (int i = 0; i < 4; ++i) job.stats_out[i] = st4[i]; }
            }
        }
    };
    if (fold_stats && r_begin >= r_end && blockIdx.x == 0) finish_stats();
    // ---- rounds: at most STRIP_ROUND active blocks each (one round on every grid the plane holds with >= 256 workgroups) ----
    for (int ra = r_begin; ra < r_end; ra += STRIP_ROUND) {
        if (ra > r_begin) {                              // (a further round: wave 0 plans it; the cell lists lay over the prefix array)
            if (wv == 0) { make_prefix(false); make_schedule(ra, min(ra + STRIP_ROUND, r_end)); }
            __syncthreads();
        }
        const int nst = __builtin_amdgcn_readfirstlane(s_misc[0]);
        if (nst == 0) break;
        // a step's entry travels in scalar registers from the step that issues its block's loads (three steps ahead)
        // to the step itself; behind the end of the schedule: idle steps
        auto entry = [&](int i, unsigned &e, unsigned &j) __attribute__((always_inline)) {
            const uint2 v = s_ent[i < nst ? i : nst - 1];
            e = i < nst ? (unsigned)__builtin_amdgcn_readfirstlane((int)v.x) : (SCH_DRAIN | SCH_IDLE);
            j = (unsigned)__builtin_amdgcn_readfirstlane((int)v.y);
        };
        // The first three steps of a round are its warm-up steps: all they do is issue the loads of steps 3, 4, 5 -- here,
        // ahead of the loop, so that the statistics below are finished while those loads travel.
        unsigned E0, J0, E1, J1, E2, J2;
        SB_T(28);
        entry(STRIP_DEPTH, E0, J0); entry(STRIP_DEPTH + 1, E1, J1); entry(STRIP_DEPTH + 2, E2, J2);
        StripRegs<T, FLY> R0, R1, R2;
        SB_T(29);
        issue(R0, J0);
        SB_T(30);
        issue(R1, J1); issue(R2, J2);
        SB_T(31);
        if (fold_stats && ra == r_begin) finish_stats();
        SB_T(4);                                         // march begins
        // A step: S1 of block i (and the list of the band cells to query), barrier, S2 (sums along latitude || queries
        // of the block two up); a drain step (behind the last block of a run) has no block and queries the block one
        // up.  The three register sets take turns -- as three copies of the step in the loop body, not as a switch, an
        // inner loop or a loop with an early exit: the compiler counts the loads in flight per program point, and only
        // straight-line rotation with the same loads on every path lets it wait for block i's loads alone while those
        // of blocks i + 1 and i + 2 stay in flight.  The set just consumed receives the loads of block i + 3.
        auto step = [&](StripRegs<T, FLY> &R, unsigned &E, unsigned &J, const unsigned &En, int i, int buf) __attribute__((always_inline)) {
            const unsigned ent = E, sj = J;
#if defined(SB_STAMPS) && !defined(SB_STAMPS_WIND)
            if (i < SB_NSTAMP - 5) SB_T(5 + i);          // step i begins (i >= 3)
#endif
            entry(i + STRIP_DEPTH, E, J);                // (consumed by `issue` below: the read travels under S1)
            const int pos = (int)(ent & 0xffffu);
            const int strip = (int)(sj >> 16), jp = (int)(sj & 0xffffu);
            const bool drain = (ent & SCH_DRAIN) != 0, idle = (ent & SCH_IDLE) != 0;
            const int qoff = drain ? 1 : 2;
            const bool qany = (ent & (drain ? SCH_Q1 : SCH_Q2)) != 0;
            if (!idle) {
                if (ent & SCH_RESTART) {
                    lds_barrier();                                // the queries of the run before have left the ring
                    if (FLY && (fold_stats || job.ngath > 0)) { sd = s_sdr[0]; rr = s_sdr[1]; }
                }
                BandWords bwd;
                const bool lister = qany && wv >= C / 2 && !cached;
                if (lister) bwd = band_issue(strip, jp - qoff);
                if (tid == STRIP_NT - 1) s_misc[1 + (buf == 2 ? 0 : buf + 1)] = 0;   // the next step's list starts empty
                if (!drain) stage(R, ent, jp);
                if (lister) list_cells(strip, jp - qoff, bwd, buf);
            }
            // The stored list of the NEXT step's query, loaded AHEAD of this step's block loads: the vector-memory counter is
            // in order, and a list loaded behind them could only be waited for together with them -- a query step then sat
            // out what was left of the latency of loads meant for three steps later (k_strip32: 1.4 us per query step).
            unsigned qn = ~0u;
            if (cached) qn = plan_lists[((En >> SCH_QI_SHIFT) & (SB_PLAN_NQ - 1)) * (unsigned)(SW * C) + qc_off];
            issue(R, J);                                          // (the one place of this copy of the step that loads blocks)
            if (!idle) {
                lds_barrier();
                if (qany && wv < C / 2) query(pos - qoff, strip, jp - qoff, buf, (ent >> SCH_QI_SHIFT) & (SB_PLAN_NQ - 1));
                if (!drain && wv >= 5 && wv < 8) vertical(jp);
            }
            // (Under the uniform condition, although every other load of the march is unconditional: with this load on
            // every path hipcc 7.2 emitted NO wait at all for the staged blocks' loads in the fp64 kernels -- the
            // results then hang on timing.  tools/check_waits.py (run by tests/test_abi_and_host.py) checks the waits of every variant in the built library.)
            if (cached) qc = qn;
        };
        for (int i = STRIP_DEPTH; i < nst; i += STRIP_DEPTH) {       // nst is a multiple of three
            step(R0, E0, J0, E1, i, 0);
            step(R1, E1, J1, E2, i + 1, 1);
            step(R2, E2, J2, E0, i + 2, 2);
        }
        __syncthreads();                                 // the schedule and the ring are free for the next round
    }

    SB_T(5);                                             // march done
    // ---- the marked cells (rare): every band cell of this workgroup's blocks that holds the mark takes the
    // global-memory search; everything it needs comes from the job's copy in device memory ----
    // the global-memory search for one marked cell
    auto slow_cell = [&](const DiagJob<T> &cj, int x, int y, int &nnmax) __attribute__((always_inline)) {
        const unsigned o = (unsigned)y * (unsigned)g.nx + (unsigned)x;
        int cap = g.nx + g.ny;
        if (limited) cap = min(cap, frame_reach(x, y));
        bool one_class;
        const T cg = contrast_global(cj, x, y, cap, sd, rr, nnmax, one_class);
        atomicAdd(&cj.counters[0], 1);
        if (one_class) atomicAdd(&cj.counters[1], 1);
        const T mulg = sb_bit(cj.clsbits, g.nw, x + g.h, y + g.h) ? T(1) : T(-1);
        job.thc[o] = mulg * cg;
    };
    if (s_misc[4] != 0 && cached) {                      // (uniform: read behind the round's last barrier)
        // A stored plan knows its marked cells: the entries of its lists whose radius field is zero -- no flags read,
        // no plane ranked.
        const DiagJob<T> &cj = *job.cold;
        int *s_qblk = (int *)&s_cell[0][0];              // strip << 16 | block of every query step's list
        __syncthreads();
        if (tid == 0) s_misc[9] = 0;
        __syncthreads();
        const int nstp = s_misc[0];
        for (int i = tid; i < nstp; i += STRIP_NT) {
            const uint2 v = s_ent[i];
            const bool dr = (v.x & SCH_DRAIN) != 0u;
            if (!(v.x & SCH_IDLE) && (v.x & (dr ? SCH_Q1 : SCH_Q2)) != 0u) {
                const int qi = (int)((v.x >> SCH_QI_SHIFT) & (SB_PLAN_NQ - 1));
                s_qblk[qi] = (int)((v.y & 0xffff0000u) | ((v.y & 0xffffu) - (dr ? 1u : 2u)));
                atomicMax(&s_misc[9], qi + 1);
            }
        }
        __syncthreads();
        const int nq = s_misc[9];
        for (int q = 0; q < nq; ++q) {
            const unsigned code = tid < SW * C ? plan_lists[(unsigned)q * (unsigned)(SW * C) + (unsigned)tid] : ~0u;
            if (code != ~0u && ((code >> 10) & 31u) == 0u) {
                const int blk = s_qblk[q];
                const int strip = blk >> 16, jp = blk & 0xffff;
                int nnmax = 0;
                slow_cell(cj, strip * SW + (int)(code & 31u), (jp - 1) * C + (int)((code >> 5) & 15u), nnmax);
                if (nnmax > 1) atomicMax(&job.flags[strip * npad + jp], nnmax);
            }
        }
        __syncthreads();                                 // (the update below lays its own table over the cell lists)
    } else if (s_misc[4] != 0) {
        const DiagJob<T> &cj = *job.cold;
        if (wv == 0) make_prefix(false);                 // the prefix array again (the cell lists lay over it)
        __syncthreads();
        tot_packed = s_misc[7];
        for (int r = r_begin; r < r_end; ++r) {
            int n;
            const int kw = find_word(r, false, n);
            if (kw < 0) break;
            const int pos = kw * 64 + nth_bit(sb_uniform64(s_bits[kw]), n);
            const int strip = pos / npad, jp = pos - strip * npad;
            int nnmax = 0;
            if (tid < SW * C) {
                const int x = strip * SW + (tid & (SW - 1)), y = (jp - 1) * C + (tid >> 5);
                if (x < g.nx && y >= 0 && y < g.rows && sb_bit(job.bandbits, g.nw, x + g.h, y + g.h)) {
                    const unsigned o = (unsigned)y * (unsigned)g.nx + (unsigned)x;
                    if (strip_is_mark(job.thc[o])) slow_cell(cj, x, y, nnmax);
                }
            }
            if (nnmax > 1) atomicMax(&job.flags[pos], nnmax);
        }
    }
    SB_T(6);                                             // marked cells done
    if (job.update) {
        // ---- a band step: k_wind ran ahead of the ghost rows and left this call's winds in scratch planes; thresholds,
        // scaling and state update (ref :235-266) of every band cell this workgroup queried, now that thc holds its
        // contrast.  Behind the march, not inside it: a cell's winds and state loaded in a step would drain the blocks the
        // march keeps in flight.  The cells come from the plan's lists (stored, or written by this very launch); four
        // list rows per wave in flight.  thc is read past the L1 (other waves of this workgroup wrote it). ----
        __syncthreads();
        const DiagJob<T> &cj = *job.cold;
        auto apply = [&](unsigned o) __attribute__((always_inline)) {
            const T n_thc = __hip_atomic_load(&job.thc[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sb_trigger_update<T, false>(cj, (size_t)o, n_thc, sb_trigger_load<T>(cj, (size_t)o));
        };
        if (cached || store_lists) {
            int *s_qblk = (int *)&s_cell[0][0];          // strip << 16 | block of every query step's list
            if (tid == 0) s_misc[9] = 0;
            __syncthreads();
            const int nstp = s_misc[0];
            for (int i = tid; i < nstp; i += STRIP_NT) {
                const uint2 v = s_ent[i];
                const bool dr = (v.x & SCH_DRAIN) != 0u;
                if (!(v.x & SCH_IDLE) && (v.x & (dr ? SCH_Q1 : SCH_Q2)) != 0u) {
                    const int qi = (int)((v.x >> SCH_QI_SHIFT) & (SB_PLAN_NQ - 1));
                    s_qblk[qi] = (int)((v.y & 0xffff0000u) | ((v.y & 0xffffu) - (dr ? 1u : 2u)));
                    atomicMax(&s_misc[9], qi + 1);
                }
            }
            __syncthreads();
            const int nrows = s_misc[9] * (C / 2);        // eight rows of 64 entries per list
            for (int r0 = wv; r0 < nrows; r0 += 4 * NWV) {
                unsigned code[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = r0 + k * NWV;
                    code[k] = ~0u;
                    if (r < nrows) code[k] = __hip_atomic_load(&plan_lists[(unsigned)(r >> 3) * (unsigned)(SW * C) + (unsigned)((r & 7) * SB_WAVE + lane)],
                                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (code[k] == ~0u) continue;
                    const int blk = s_qblk[(r0 + k * NWV) >> 3];
                    const int x = (blk >> 16) * SW + (int)(code[k] & 31u), y = ((blk & 0xffff) - 1) * C + (int)((code[k] >> 5) & 15u);
                    apply((unsigned)y * (unsigned)g.nx + (unsigned)x);
                }
            }
        } else {
            // (no lists: a share of several rounds, or of more query steps than a plan holds -- block by block)
            if (wv == 0) make_prefix(false);
            __syncthreads();
            tot_packed = s_misc[7];
            for (int r = r_begin; r < r_end; ++r) {
                int n;
                const int kw = find_word(r, false, n);
                if (kw < 0) break;
                const int pos = kw * 64 + nth_bit(sb_uniform64(s_bits[kw]), n);
                const int strip = pos / npad, jp = pos - strip * npad;
                if (tid < SW * C) {
                    const int x = strip * SW + (tid & (SW - 1)), y = (jp - 1) * C + (tid >> 5);
                    if (x < g.nx && y >= 0 && y < g.rows && sb_bit(job.bandbits, g.nw, x + g.h, y + g.h))
                        apply((unsigned)y * (unsigned)g.nx + (unsigned)x);
                }
            }
        }
    }
    if (job.fold && !(cached && job.lists_stand)) {
        // ---- k_wind's segment lists: compacted when the plan is made, and again only when it is made again (they follow
        // from the band plane, as the plan does) (k_prep's work on single-domain host-model calls): sub-list `part` holds the
        // segments with band cells of its contiguous range of the band plane, in ascending order ----
        const unsigned nseg = (unsigned)g.nyh * (unsigned)g.nw;
        const unsigned cap = (unsigned)job.seg_cap;
        for (int part = G - 1 - (int)blockIdx.x; part < SB_SEG_PARTS; part += G) {
            if (part < 0) break;
            const unsigned s0 = (unsigned)part * cap, s1 = min(s0 + cap, nseg);
            const unsigned per = (cap + STRIP_NT - 1) / STRIP_NT;
            const unsigned a0 = min(s0 + (unsigned)tid * per, s1), a1 = min(a0 + per, s1);
            int cnt = 0;
            for (unsigned sg = a0; sg < a1; ++sg) cnt += job.bandbits[sg] != 0 ? 1 : 0;
            int total;
            int at = thc_block_excl_scan<STRIP_NT>(cnt, s_scan, total);
            SbSegEntry *list = job.seg_list + (size_t)part * cap;
            for (unsigned sg = a0; sg < a1; ++sg) {
                const u64 w = job.bandbits[sg];
                if (w) { SbSegEntry e; e.word = w; e.seg = sg; e.pad = 0; list[at++] = e; }
            }
            if (tid == 0) job.seg_count[part] = total;
        }
    }
    SB_T(7);                                             // end
}

// the hot part of the job, by value; everything else the kernel reads -- rarely -- from the copy of the whole job that
// k_scan leaves in device memory (job.self)
template <typename T>
static StripJob<T> strip_job(const DiagJob<T> &job) {
    StripJob<T> s;
    s.g = job.g;
    s.theta = job.t0_fly ? job.theta : job.t0; s.z = job.z; s.sigma = job.sigma;
    s.clsbits = job.clsbits; s.bandbits = job.bandbits;
    s.thc = job.thc;
    s.flags = job.tile_nnmax;
    s.ntx = job.thc_ntx; s.nty = job.thc_nty;
    s.fold = job.fold; s.fold_nparts = job.fold_nparts; s.ngath = job.ngath; s.seg_cap = job.seg_cap;
    s.lists_stand = job.lists_stand;
    s.stats = job.stats; s.stats_out = job.stats_out;
    s.fold_partials = job.fold_partials; s.gath = job.gath;
    s.seg_list = job.seg_list; s.seg_count = job.seg_count;
    s.cold = job.self;
    s.plan = job.plan; s.plan_gen = job.plan_gen; s.call_id = job.call_id; s.plan_use = job.plan_use;
    s.update = job.strip_update;
    s.stamps = job.stamps;
    return s;
}

template <typename T>
hipError_t sb_launch_strip(const DiagJob<T> &job, int ncu, hipStream_t st) {
    const dim3 gr(ncu), bl(STRIP_NT);                   // one persistent workgroup per CU
    const StripJob<T> sj = strip_job<T>(job);
    if (!job.wind_final) return hipErrorInvalidValue;   // (the update is k_wind's: sb_launch_diag sees to it)
    if (job.t0_fly) hipLaunchKernelGGL((k_strip<T, true>), gr, bl, 0, st, sj.plan, sj.plan_gen, sj.fold_partials, ncu, sj);
    else hipLaunchKernelGGL((k_strip<T, false>), gr, bl, 0, st, sj.plan, sj.plan_gen, sj.fold_partials, ncu, sj);
    return hipGetLastError();
}
template hipError_t sb_launch_strip<float>(const DiagJob<float> &, int, hipStream_t);
template hipError_t sb_launch_strip<double>(const DiagJob<double> &, int, hipStream_t);

// the strip kernel's block grid for a domain of nx x rows interior cells; false: the position plane cannot hold it
bool sb_strip_shape(int nx, int rows, int *ntx, int *nty) {
    *ntx = (nx + STRIP_SW - 1) / STRIP_SW;
    *nty = (rows + STRIP_C - 1) / STRIP_C;
    return (long long)*ntx * (*nty + 2) < (long long)STRIP_MAXW * 64;
}
