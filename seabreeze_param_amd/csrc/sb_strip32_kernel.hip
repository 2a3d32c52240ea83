// sb_strip32_kernel.hip -- thermal heating contrast on gfx950 for search radii up to 31 cells in SINGLE precision:
// marching strips with 96-column rows (the N2560 grid of BASELINE configs[3]: its distance field is made with a window
// of 30 cells, so a land/sea window reaches at most 31).
// ref: generic/sea_breeze_diag.f90:166-167 (t0), :188-216 (window search and contrast),
//      python_wrapper/seabreezediag/seabreeze_diag_python.f90:187-221
//
// The march is k_strip's (sb_strip_kernel.hip; read that header first): strips of 32 owned longitudes, blocks of 16 rows,
// summed-area tables in a ring in LDS, every row staged once per run, a stored plan.  What a halo of 32 cells changes:
//
// * A staged row is 96 columns -- 32 owned, 32 either side -- i.e. one and a half waves.  Every wave stages TWO pieces
//   of its row: segment A (columns 0..63, lane = column) and segment B (columns 64..95, lanes 0..31), each with a
//   prefix sum of its own: no carry crosses from one wave's registers to another's.  A window that straddles column 64
//   is the A part up to column 63 plus the B part: six table reads per corner pair instead of four.
// * The tables must hold 2 x 31 + 1 rows round a cell plus the block being staged; what 160 KB of LDS hold is a ring of
//   SIX blocks (96 rows) of 97 entries of 16 bytes -- and that only because an entry is two words, not three: single
//   precision leaves room to PACK.  t0 goes to fixed point with 24 fractional bits (exact for every fp32 value >= 0.5 K
//   in magnitude; 2^-25 K otherwise), |t0| < 2048 K, a window holds < 2^13 cells, so a window sum needs 48 bits: the
//   land-side table holds (sum << 16 | count) in one 64-bit word, sums and counts wrap independently, and the four-
//   (six-)corner combination gives the window's sum and its land-side count at once -- there is no count table.
//   (Double precision needs all 64 bits for the sum alone: radii beyond 16 stay with the tile kernel there.)
// * With six blocks in the ring the block being staged may not be written while the block three up is queried (its
//   window's first row lies five blocks back): a step has TWO barriers, stage | sums along latitude + queries.
//   A window of radius 32 would reach one row further back still: radii beyond 31 are marked and take the global path.
// * An active block needs two staged blocks either side; the flags carry two virtual blocks above and below a strip;
//   a block is queried when the block three positions down has been staged.
#include "sb_strip_common.hpp"

#define S32_HB 2                  // halo of the tables in blocks
#define S32_HMAX 31               // largest radius answered from LDS
#define S32_W 96                  // staged columns
#define S32_SW 32                 // owned columns
#define S32_C 16                  // rows per block = waves per workgroup
#define S32_NT 1024
#define S32_NBLK 6                // blocks in the ring
#define S32_RING (S32_NBLK * S32_C)
#define S32_P 97                  // table pitch: 64 entries of segment A, 32 of segment B, one of padding (odd: no bank conflicts down a column)
#define S32_MAXW 768              // 64-bit words of the position plane a workgroup can hold (49,151 positions)
#define S32_SCHED 384             // steps of one round of a workgroup
#define S32_ROUND 60              // active blocks of one round (at most 5 x 60 staged blocks + 60 drain + 3 warm-up + 2 padding steps)
#define S32_FB 24                 // fractional bits of the fixed-point t0
#define S32_DEPTH 3               // blocks of inputs in flight per wave
#define S32_QOFF (S32_HB + 1)     // a block is queried in the step that stages the block this many positions down

#if defined(SB_STAMPS) && !defined(SB_STAMPS_WIND)
#define SB_T(i) do { if (lane == 0) job.stamps[(size_t)(blockIdx.x * (S32_NT / SB_WAVE) + wv) * SB_NSTAMP + (i)] = wall_clock64(); } while (0)
#else
#define SB_T(i) do { } while (0)
#endif

// t0 (K) -> fixed point with 24 fractional bits, BIASED (as k_strip's sb_to_fixed: the bits of fma(x, 2^24, 1.5 * 2^52))
#define S32_FIX_BIAS 0x4338000000000000ull
__device__ __forceinline__ u64 s32_to_fixed(double x) {
    return (u64)__double_as_longlong(__builtin_fma(x, 0x1p24, 0x1.8p52));
}

// what a lane holds of one piece (segment A or B) of a staged row between the issue of its loads and S1
template <bool FLY>
struct S32Regs {
    float th;                      // theta (FLY) or t0
    float zz, sg;                  // z, sigma (FLY only)
    uint32_t lw;                   // the 32-bit half of the land-side word that holds the cell
    uint32_t lbit;                 // the cell's bit in lw; 0: no such cell
};

// The LDS of one workgroup.
struct S32Lds {
    u64 sA[S32_RING * S32_P];      // prefix sums of t0 (fixed point, biased), every cell
    u64 sL[S32_RING * S32_P];      // land-side cells: (sum << 16) | count
    u64 s_land[S32_RING];          // land-side bits of segment A of every ring row
    u64 s_bits[S32_MAXW];          // the active blocks as a bit plane
    uint2 s_ent[S32_SCHED];        // steps of the round: x = position | flags, y = strip << 16 | block within the padded strip
    Moments s_wpart[S32_NT / SB_WAVE];
    float s_sdr[2];
    int s_scan[S32_NT / SB_WAVE];
    int s_misc[12];
    unsigned short s_cell[3][S32_SW * S32_C];
};
static_assert(sizeof(S32Lds) <= 160 * 1024, "k_strip32: LDS of one workgroup");

// FLY: t0 from theta, z, sigma while staging.  The contrast goes to thc; thresholds and state update are k_wind's, or --
// a band step -- applied behind the march (job.update), as in k_strip.
template <bool FLY>
__global__ __launch_bounds__(S32_NT) void k_strip32(char *plan, const int *plan_gen, const Moments *fold_partials, int G, StripJob<float> job) {
    typedef float T;
    constexpr int HB = S32_HB, SW = S32_SW, C = S32_C, P = S32_P, NWV = S32_NT / SB_WAVE, RING = S32_RING;
    static_assert(NWV == C && SW == 32, "one staged row per wave, 32 owned columns");
    static_assert(S32_SCHED == SB_PLAN_SCHED, "a stored plan holds one round's steps");
    __shared__ S32Lds L;
    u64 (&sA)[S32_RING * S32_P] = L.sA;
    u64 (&sL)[S32_RING * S32_P] = L.sL;
    u64 (&s_land)[S32_RING] = L.s_land;
    u64 (&s_bits)[S32_MAXW] = L.s_bits;
    uint2 (&s_ent)[S32_SCHED] = L.s_ent;
    unsigned short (&s_cell)[3][S32_SW * S32_C] = L.s_cell;
    Moments (&s_wpart)[S32_NT / SB_WAVE] = L.s_wpart;
    int (&s_scan)[S32_NT / SB_WAVE] = L.s_scan;
    int (&s_misc)[12] = L.s_misc;                      // [0] steps of the round, [1..3] entries of the three cell lists, [4] a cell was marked,
                                                       // [5], [6] the share (ranks of active blocks), [7] totals of the plane,
                                                       // [8] the plan of this call is stored (incl. its cell lists),
                                                       // [9] query steps of the plan
    float (&s_sdr)[2] = L.s_sdr;

    const Geo g = job.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int npad = job.nty + 2 * HB;                 // blocks of a strip incl. the virtual ones above and below
    const int npos = job.ntx * npad;
    const unsigned npad_magic = 0xffffffffu / (unsigned)npad + 1u;       // floor(p / npad) = umulhi(p, magic) for p < 2^16

    SB_T(0);                                             // start
    // ---- prologue 1: statistics; the flags k_scan raised, 1024 at a time, as a bit plane ----
    T sd = T(0), rr = T(0);
    Moments pm = moments_empty();
    double shift_c = 0.0;
    const bool fold_stats = job.fold && job.fold_nparts > 0;
    if (fold_stats) {
        if (tid < job.fold_nparts) pm = fold_partials[tid];
        shift_c = (double)job.sigma[(size_t)g.h * g.nxh + g.h];
    } else if (FLY && job.ngath > 0) {
        // band step: the first wave merges the moments gathered from all ranks in rank order; workgroup 0 publishes them
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
    const int nch = (npos + S32_NT - 1) / S32_NT;            // <= S32_MAXW / 16 (host)
    const int nwords = nch * NWV;
    // ---- the plan (see k_strip): stored in device memory, valid while k_scan finds both planes unchanged ----
    typedef const __attribute__((address_space(4))) int *cintp;
    char *const plan_wg = plan + (size_t)blockIdx.x * SB_PLAN_STRIDE;
    unsigned *const plan_lists = (unsigned *)(plan_wg + SB_PLAN_LIST_OFF);
    // (the stored plan's steps travel WITH its header -- one round trip, not two; used only if the plan stands)
    const uint2 plan_ent = ((const uint2 *)(plan_wg + SB_PLAN_ENT_OFF))[tid < S32_SCHED ? tid : 0];
    const int plan_stored = ((cintp)plan_wg)[0], plan_nst = min(((cintp)plan_wg)[1], S32_SCHED);
    const int plan_rb = ((cintp)plan_wg)[2], plan_re = ((cintp)plan_wg)[3];
    const bool cached = job.plan_use != 0 && plan_stored != 0 && *(cintp)plan_gen <= plan_stored;      // uniform
    auto load_plane = [&]() {
        u64 mine = 0;
        for (int base = 0; base < nch; base += 8) {              // 8 loads in flight (clamped, so none is conditional)
            int f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = (base + j) * S32_NT + tid;
                f[j] = job.flags[i < npos ? i : npos - 1];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = (base + j) * S32_NT + tid;
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
        if (tid < plan_nst) s_ent[tid] = plan_ent;      // (at most S32_SCHED < 1024 steps)
        if (tid == 0) { s_misc[0] = plan_nst; s_misc[5] = plan_rb; s_misc[6] = plan_re; s_misc[8] = 0; }
    }
    if (tid == 0) s_misc[4] = 0;
    if (fold_stats) wave_total_shifted_store(pm, s_wpart);
    SB_T(1);                                             // first barrier reached
    __syncthreads();
    SB_T(2);                                             // ... passed
    // ---- prologue 2, WAVE 0 ALONE: this workgroup's share (equal in cost: 4 per staged block, 2 per active block, 2 per
    // run, as k_strip) and the schedule of its first round ----
    unsigned *s_pre = (unsigned *)&s_cell[0][0];
    unsigned *s_cost = (unsigned *)&sA[0];
    constexpr int CW_STAGED = 4, CW_ACTIVE = 2, CW_RUN = 2;
    static_assert(sizeof(L.s_cell) >= sizeof(unsigned) * S32_MAXW, "the prefix array fits where the cell lists lie");
    auto wave_sync = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };      // this wave's LDS writes have landed
    auto stage_word = [&](int k) -> u64 {                // staged = active, or within two positions of an active block
        const u64 a = s_bits[k], pv = k > 0 ? s_bits[k - 1] : 0ull, nx = k + 1 < nwords ? s_bits[k + 1] : 0ull;
        return a | (a << 1) | (pv >> 63) | (a << 2) | (pv >> 62) | (a >> 1) | (nx << 63) | (a >> 2) | (nx << 62);
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
    // The schedule of the round that holds the active blocks of ranks [ra, rb): the staged positions in ascending order
    // with their flags, a drain step behind every run, three warm-up steps in front, padded to a multiple of three.
    auto make_schedule = [&](int ra, int rb) {           // one wave; leaves the number of steps in s_misc[0]
        int n0, n1;
        const int k0w = find_word(ra, false, n0), k1w = find_word(rb - 1, false, n1);
        const int p0 = k0w < 0 ? -1 : k0w * 64 + nth_bit(sb_uniform64(s_bits[k0w < 0 ? 0 : k0w]), n0);
        const int p1 = k1w < 0 ? -1 : k1w * 64 + nth_bit(sb_uniform64(s_bits[k1w < 0 ? 0 : k1w]), n1);
        wave_sync();                                     // (the prefix array may be overwritten from here on)
        if (p0 < HB || p1 < p0) { if (lane == 0) s_misc[0] = 0; return; }     // (cannot happen: the first HB positions are virtual)
        if (lane < S32_DEPTH) s_ent[lane] = make_uint2(SCH_DRAIN | SCH_IDLE, 0u);
        if (lane < 3) s_misc[1 + lane] = 0;
        int n_out = S32_DEPTH;
        for (int c = p0 - HB; c <= p1 + HB; c += SB_WAVE) {
            const int pp = c + lane;
            // active blocks (of this round) at positions pp - 3 .. pp + 3: bits 0 .. 6
            unsigned win = 0;
#pragma unroll
            for (int d = 0; d < 2 * HB + 3; ++d) {
                const int q = pp + d - (HB + 1);
                const bool in = q >= p0 && q <= p1;
                const u64 w = s_bits[in ? q >> 6 : 0];
                win |= (in && ((w >> (q & 63)) & 1ull)) ? 1u << d : 0u;
            }
            const bool st = pp <= p1 + HB && (win & 0x3eu) != 0u;                 // pp - 2 .. pp + 2
            // (a run never crosses from one strip into the next: the last virtual block of a strip ends it, the first one
            // of the next strip starts afresh -- the blocks of a run are queried by their position within ONE strip)
            const int sp = (int)__umulhi((unsigned)(pp < 0 ? 0 : pp), npad_magic), jpp = pp - sp * npad;
            const bool st_prev = (win & 0x1fu) != 0u && jpp != 0, st_next = (win & 0x7cu) != 0u && jpp != npad - 1;
            const bool en = st && !st_next;
            const u64 ms = __builtin_amdgcn_ballot_w64(st), me = __builtin_amdgcn_ballot_w64(en);
            const u64 below = (1ull << lane) - 1ull;
            const int at = n_out + __popcll(ms & below) + __popcll(me & below);
            n_out += __popcll(ms) + __popcll(me);
            if (st) {
                const unsigned sjv = ((unsigned)sp << 16) | (unsigned)jpp;
                const unsigned e = (unsigned)pp | ((win & 1u) ? SCH_Q2 : 0u) | (st_prev ? 0u : SCH_RESTART);
                if (at < S32_SCHED) s_ent[at] = make_uint2(e, sjv);
                // (the run ends at pp: the block two up is the run's last active one)
                if (en && at + 1 < S32_SCHED) s_ent[at + 1] = make_uint2((unsigned)pp | SCH_DRAIN | ((win & 2u) ? SCH_Q1 : 0u), sjv);
            }
        }
        n_out = min(n_out, S32_SCHED - 2);
        const int n_pad = (n_out + S32_DEPTH - 1) / S32_DEPTH * S32_DEPTH;
        wave_sync();
        const unsigned last_sj = s_ent[n_out - 1].y;         // (n_out >= S32_DEPTH + 1 here)
        if (lane < n_pad - n_out) s_ent[n_out + lane] = make_uint2(SCH_DRAIN | SCH_IDLE, last_sj);
        if (lane == 0) s_misc[0] = n_pad;
    };
    if (__builtin_expect(!cached, 0)) {
        if (wv == 0) {
            make_prefix(true);
            const int rb0 = act_before_cost((int)(((long long)blockIdx.x * tot_cost) / G));
            const int re0 = blockIdx.x + 1 == (unsigned)G ? (tot_packed & 0xffff) : act_before_cost((int)(((long long)(blockIdx.x + 1) * tot_cost) / G));
            if (lane == 0) { s_misc[5] = rb0; s_misc[6] = re0; s_misc[0] = 0; }
            if (rb0 < re0) make_schedule(rb0, min(rb0 + S32_ROUND, re0));
            wave_sync();
            // the plan goes to device memory: the steps, each query step with the number of its cell list
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
            const bool ok = re0 - rb0 <= S32_ROUND && nq <= SB_PLAN_NQ;
            if (lane == 0) {
                int *h = (int *)plan_wg;
                h[1] = nstv; h[2] = rb0; h[3] = re0;
                h[0] = ok ? job.call_id : 0;
                s_misc[8] = ok ? 1 : 0;
            }
        }
        __syncthreads();
        SB_T(3);                                         // planned
    }
    const bool store_lists = __builtin_amdgcn_readfirstlane(s_misc[8]) != 0;
    unsigned qc = ~0u;                                   // the cell list entry of the next step's query, from the stored plan
    const unsigned qc_off = (unsigned)(min(wv, C / 2 - 1) * SB_WAVE + lane);
    const int r_begin = __builtin_amdgcn_readfirstlane(s_misc[5]), r_end = __builtin_amdgcn_readfirstlane(s_misc[6]);

    const bool fastx = g.nx > S32_W + 2;               // one conditional add wraps every column of a staged row
    const bool limited = g.bnd == BND_HALO;
    // how far a window round (x, y) may reach inside a ghost-celled frame: the ghost width beyond the interior -- except
    // in the directions in which a band's frame is not an edge at all (round the circle; beyond a pole)
    const int big_reach = 1 << 20;
    auto frame_reach = [&](int x, int y) __attribute__((always_inline)) -> int {
        const int rx = (g.band & GEO_BAND_EW) ? big_reach : min(x + g.h, g.nx - 1 - x + g.h);
        const int rs = (g.band & GEO_BAND_SOUTH) ? big_reach : y + g.h, rn = (g.band & GEO_BAND_NORTH) ? big_reach : g.ny - 1 - y + g.h;
        return min(rx, min(rs, rn));
    };

    // the lane's columns of the strip the loads are issued for (segment A: column lane; segment B: column 64 + lane, lanes
    // 0 .. 31): byte offsets in a field row and in a row of the land-side plane, bit in the 32-bit word (0: no such cell)
    int cc_strip = -1;
    unsigned cc_colA = 0, cc_clsA = 0, cc_bitA = 0, cc_colB = 0, cc_clsB = 0, cc_bitB = 0;
    auto column_of = [&](int xs, bool live, unsigned &colb, unsigned &clsb, unsigned &lbit) __attribute__((always_inline)) {
        bool ok = live;
        int Xc = 0;
        if (g.bnd == BND_HALO) {
            int xw = xs;
            if (g.band & GEO_BAND_EW) xw = xs < 0 ? xs + g.nx : (xs >= g.nx ? xs - g.nx : xs);   // (a band holds whole circles; nx > 96 + 2)
            Xc = xw + g.h; ok = ok && Xc >= 0 && Xc < g.nxh;
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
        const unsigned xc = ok ? (unsigned)Xc : 0u;      // every load is unconditional, from a clamped address
        colb = xc * (unsigned)sizeof(T);
        clsb = (xc >> 5) * 4u;
        lbit = ok ? 1u << (xc & 31u) : 0u;
    };

    // loads of row wv of block jp of `strip`: both pieces.  Always the same loads, also behind the end of the schedule and
    // for a drain step, from clamped addresses (see k_strip: the compiler counts the loads in flight per program point).
    auto issue = [&](S32Regs<FLY> &RA, S32Regs<FLY> &RB, unsigned sj) __attribute__((always_inline)) {
        const int strip = (int)(sj >> 16), jp = (int)(sj & 0xffffu);
        if (strip != cc_strip) {                         // wave-uniform
            cc_strip = strip;
            column_of(strip * SW - 32 + lane, true, cc_colA, cc_clsA, cc_bitA);
            column_of(strip * SW + 32 + (lane & 31), lane < 32, cc_colB, cc_clsB, cc_bitB);
        }
        const int ys = (jp - HB) * C + wv;              // interior row (may lie outside the grid: clamped or absent)
        int Yr;
        bool rowok = true;
        if (g.bnd == BND_HALO) {
            int yw = ys;
            if ((g.band & GEO_BAND_SOUTH) && yw < 0) yw = 0;          // beyond a pole: the edge row again (the latitude clamp)
            if ((g.band & GEO_BAND_NORTH) && yw >= g.ny) yw = g.ny - 1;
            Yr = yw + g.h; rowok = Yr >= 0 && Yr < g.nyh; Yr = rowok ? Yr : 0;
        }
        else Yr = ys < 0 ? 0 : (ys >= g.ny ? g.ny - 1 : ys);
        const size_t rowb = (size_t)((unsigned)Yr * (unsigned)g.nxh) * sizeof(T), wordb = (size_t)((unsigned)Yr * (unsigned)g.nw) * 8u;
        RA.th = *(const T *)((const char *)job.theta + rowb + cc_colA);                 // (theta is the t0 plane unless FLY)
        RB.th = *(const T *)((const char *)job.theta + rowb + cc_colB);
        if constexpr (FLY) {
            RA.zz = *(const T *)((const char *)job.z + rowb + cc_colA);
            RB.zz = *(const T *)((const char *)job.z + rowb + cc_colB);
            RA.sg = *(const T *)((const char *)job.sigma + rowb + cc_colA);
            RB.sg = *(const T *)((const char *)job.sigma + rowb + cc_colB);
        }
        RA.lw = *(const uint32_t *)((const char *)job.clsbits + wordb + cc_clsA);
        RB.lw = *(const uint32_t *)((const char *)job.clsbits + wordb + cc_clsB);
        RA.lbit = rowok ? cc_bitA : 0u;
        RB.lbit = rowok ? cc_bitB : 0u;
    };

    // running column totals of the table this wave sums along latitude (waves 8, 9, 10)
    u64 carry = 0;

    // ring row of row r of the block at padded position jp (blocks never straddle the end of the ring)
    auto slot_of = [&](int jp, int r) __attribute__((always_inline)) -> unsigned {
        return (unsigned)((jp % S32_NBLK) * C + r);
    };

    // S1: one piece of the row -> its ring row (prefix along longitude only).  SEG 0: columns 0 .. 63; 1: columns 64 .. 95
    // (lanes 0 .. 31; the upper lanes carry zeros through the scan and write nothing).
    auto stage = [&](S32Regs<FLY> &R, unsigned slot, int seg) __attribute__((always_inline)) {
        // (the block's loads have landed once at most two blocks' loads -- two pieces each -- are outstanding: see k_strip)
        if constexpr (FLY) asm volatile("s_waitcnt vmcnt(16)" : "+v"(R.th), "+v"(R.zz), "+v"(R.sg), "+v"(R.lw) : : "memory");
        else asm volatile("s_waitcnt vmcnt(8)" : "+v"(R.th), "+v"(R.lw) : : "memory");
        const bool live = seg == 0 || lane < 32;
        const bool land = live && (R.lw & R.lbit) != 0u;
        const u64 lm = __builtin_amdgcn_ballot_w64(land);
        T t0v = R.th;
        if constexpr (FLY) {
            // the sigmoid only where a lane of the wave stands above sea level (z == 0 -> t0 = theta exactly)   ref :166-167
            if (__builtin_amdgcn_ballot_w64(R.zz != T(0)) != 0) t0v = sb_t0<T>(R.th, R.zz, R.sg, sd, rr);
        }
        u64 qa = s32_to_fixed((double)t0v);
        if (!live || (limited && R.lbit == 0u)) qa = 0ull;   // (the upper half of a B wave; ghost-celled frames: cells beyond the frame)
        u64 ql = land ? (qa << 16) | 1ull : 0ull;            // (the bias leaves by the shift: its low 48 bits are zero)
        const u64 all = seg == 0 ? ~0ull : 0xffffffffull;
        if (lm == 0ull) sb_scan1_u64(qa);                // wave-uniform; ql is zero everywhere
        else if (lm == all) { sb_scan1_u64(qa); ql = (qa << 16) + (u64)(lane + 1); }      // every cell land side: sums and counts follow
        else sb_scan2_u64(qa, ql);
        const unsigned o = __umul24(slot, P) + (seg == 0 ? 0u : 64u) + (unsigned)lane;
        if (live) { sA[o] = qa; sL[o] = ql; }
        if (seg == 0 && !cached && lane == 0) s_land[slot] = lm;
    };

    // S2, waves 8 .. 10: prefix along latitude of the 16 rows of the block at padded position jp: wave 8 the all-cells table's
    // segment A, wave 9 the land-side table's, wave 10 both tables' segment B (lanes 0 .. 31 | 32 .. 63)
    auto vertical = [&](int jp) __attribute__((always_inline)) {
        const unsigned r0 = slot_of(jp, 0);
        u64 *tab;
        if (wv == 8) tab = sA + __umul24(r0, P) + lane;
        else if (wv == 9) tab = sL + __umul24(r0, P) + lane;
        else tab = (lane < 32 ? sA : sL) + __umul24(r0, P) + 64 + (lane & 31);
#pragma unroll
        for (int h0 = 0; h0 < C; h0 += 8) {              // eight rows of reads in flight
            u64 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = tab[(h0 + i) * P];
#pragma unroll
            for (int i = 0; i < 8; ++i) { carry += v[i]; tab[(h0 + i) * P] = carry; }
        }
    };

    // The band bits of the two rows this wave lists in block jp of `strip`, as SCALAR loads (see k_strip)
    struct BandWords { u64 a0, b0, a1, b1, c0, c1; int sh; };   // c: land-side word of the last longitude (f2py rule)
    auto band_issue = [&](int strip, int jp) __attribute__((always_inline)) -> BandWords {
        const int k = max(wv - C / 2, 0);                // the listing waves are 8 .. 15: two rows each
        const int y0 = (jp - HB) * C + 2 * k;
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

    // S1, waves 8-15: the band cells of two rows of the queried block -> the step's compact list (as k_strip)
    const unsigned cell_code = (unsigned)((2 * max(wv - C / 2, 0) + (lane >> 5)) << 5 | (lane & (SW - 1)));   // row << 5 | column
    auto list_cells = [&](int strip, int jp, const BandWords &bwd, int buf) __attribute__((always_inline)) {
        const int y0 = (jp - HB) * C + 2 * max(wv - C / 2, 0);
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

    // S2, waves 0 .. 7: 64 entries of the list per wave: radius, contrast, result
    // A cell's entry in the stored plan: bits 0-8 row << 5 | column, bit 9 its own class, bits 10-14 the radius of its window
    // (0: it outgrows the tables), bits 15-27 the land-side cells in it; all ones: no cell.
    auto query = [&](int qpos, int strip, int jp, int buf, unsigned qi) __attribute__((always_inline)) {
        unsigned code;
        bool valid, found, own;
        int nn, nl;
        if (cached) {                                    // uniform: the list of the stored plan (loaded one step ahead)
            code = qc;
            valid = code != ~0u;
            if (__builtin_amdgcn_ballot_w64(valid) == 0ull) return;
            nn = (int)((code >> 10) & 31u);
            nl = (int)((code >> 15) & 8191u);
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
        const int x = strip * SW + lx, y = (jp - HB) * C + ly;
        const unsigned o = (unsigned)y * (unsigned)g.nx + (unsigned)x;     // (fewer than 2^31 cells: check_dims)
        const int rho = (int)slot_of(jp, ly);            // ring row of the cell
        const int cx = lx + 32;                          // its staged column (segment A)
        // the six entries of a window of radius rad: rows r1 = rho + rad and r0 = rho - rad - 1 (ring), columns
        // left = cx - rad - 1 (segment A: >= 0 for rad <= 31), min(cx + rad, 63) and -- where the window crosses into
        // segment B -- cx + rad itself
        auto corners = [&](int rad, unsigned &i11, unsigned &i10, unsigned &i01, unsigned &i00, unsigned &b1, unsigned &b0, bool &hasb) __attribute__((always_inline)) {
            int r1 = rho + rad, r0 = rho - rad - 1;
            r1 -= r1 >= RING ? RING : 0;
            r0 += r0 < 0 ? RING : 0;
            const int right = cx + rad, left = cx - rad - 1;
            hasb = right >= 64;
            const unsigned p1 = __umul24((unsigned)r1, P), p0 = __umul24((unsigned)r0, P);
            const unsigned rc = (unsigned)min(right, 63), rb = (unsigned)max(right, 64);
            i11 = p1 + rc; i10 = p1 + (unsigned)left; i01 = p0 + rc; i00 = p0 + (unsigned)left;
            b1 = p1 + rb; b0 = p0 + rb;
        };
        if (__builtin_expect(!cached, 0)) {
            int lim = S32_HMAX;
            if (limited) lim = min(lim, frame_reach(x, y));   // uniform branch
            const int limc = max(lim, 1);
            // land-side count of the square of radius rad: the low 16 bits of the land-side table's six-entry combination
            const unsigned short *c16 = (const unsigned short *)sL;
            auto count = [&](int rad) __attribute__((always_inline)) {
                unsigned i11, i10, i01, i00, b1, b0;
                bool hasb;
                corners(rad, i11, i10, i01, i00, b1, b0, hasb);
                const unsigned a = (unsigned)c16[4 * i11] - (unsigned)c16[4 * i01] - (unsigned)c16[4 * i10] + (unsigned)c16[4 * i00];
                const unsigned b = (unsigned)c16[4 * b1] - (unsigned)c16[4 * b0];
                return (int)(unsigned short)(a + (hasb ? b : 0u));
            };
            // One round of independent probes -- radii 8, 16, 24, 31 -- brackets the answer ("holds both classes" is
            // monotone in the radius); three dependent probes bisect the bracket of at most eight radii.  (This is the
            // planning call's path: a stored plan knows every radius.  Seven independent probes instead of the bisection
            // cost the kernel 13 spilled vector registers.)
            int nl1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) nl1[k] = count(min(k == 3 ? S32_HMAX : 8 * (k + 1), limc));
            int lo = 1, hi = limc;
            nl = 0;
            bool got = false;
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                const int rad = min(k == 3 ? S32_HMAX : 8 * (k + 1), limc);
                const bool mixed = nl1[k] > 0 && nl1[k] < (2 * rad + 1) * (2 * rad + 1);
                if (mixed) { hi = rad; nl = nl1[k]; got = true; }
                else if (rad < hi) lo = max(lo, rad + 1);
            }
            found = valid && lim >= 1 && got;
            if (!got) lo = hi;                               // (no probe needed; the result is not used)
#pragma unroll 1
            for (int it = 0; it < 3; ++it) {                 // lo <= answer <= hi, hi holds both classes; hi - lo < 8
                const int mid = (lo + hi) >> 1;              // (mid < hi unless lo == hi)
                const int c = count(mid);
                const bool mixed = c > 0 && c < (2 * mid + 1) * (2 * mid + 1);
                if (lo < hi) {
                    if (mixed) { hi = mid; nl = c; }
                    else lo = mid + 1;
                }
            }
            nn = hi;
            // the cell's own class: the table's centre, except at the last longitude under the f2py boundary rule (see k_strip)
            const u64 ownw = s_land[rho];
            own = (code >> 10) & 1u ? ((code >> 9) & 1u) != 0u : ((ownw >> cx) & 1ull) != 0ull;
            if (store_lists)
                plan_lists[qi * (unsigned)(SW * C) + (unsigned)(wv * SB_WAVE + lane)] =
                    valid ? (code & 511u) | (own ? 1u << 9 : 0u) | (found ? (unsigned)nn << 10 : 0u) | (unsigned)nl << 15 : ~0u;
        }
        const int area = (2 * nn + 1) * (2 * nn + 1);
        unsigned i11, i10, i01, i00, b1, b0;
        bool hasb;
        corners(nn, i11, i10, i01, i00, b1, b0, hasb);
        const u64 l11 = sL[i11], l01 = sL[i01], l10 = sL[i10], l00 = sL[i00], lb1 = sL[b1], lb0 = sL[b0];
        const u64 q11 = sA[i11], q01 = sA[i01], q10 = sA[i10], q00 = sA[i00], qb1 = sA[b1], qb0 = sA[b0];
        // exact: the tables wrap, the window's sums do not.  Land side: (sum << 16) + count; all cells: sum + area x bias.
        const u64 PL = (l11 - l01) - (l10 - l00) + (hasb ? lb1 - lb0 : 0ull);
        const long long RL = (long long)(PL - (u64)nl) >> 16;
        const long long RS = (long long)((q11 - q01) - (q10 - q00) + (hasb ? qb1 - qb0 : 0ull) - (u64)area * S32_FIX_BIAS) - RL;      // sea side
        auto to_f64 = [](long long v) { return __builtin_fma((double)(int)(v >> 32), 0x1p32, (double)(unsigned)v); };
        const double dnl = (double)nl, dns = (double)(area - nl);
        const double num = to_f64(RL) * dns - to_f64(RS) * dnl;
        const T contrast = (T)(num * sb_inv(dnl * dns) * 0x1p-24);
        const T mul = own ? T(1) : T(-1);
        int nnmax = 0;
        if (found) { nnmax = nn; job.thc[o] = mul * contrast; }              // ref :216; k_wind applies :235-266
        // cells whose window outgrows the tables: marked, handled behind the march
        if (valid && !found) { job.thc[o] = strip_mark<T>(); s_misc[4] = 1; }
        // per-block largest radius (diagnostic, read by sb_last_counters); the flag k_scan raised is 1
        nnmax = sb_wave_max_to_last(nnmax);
        if (lane == SB_WAVE - 1 && nnmax > 1) atomicMax(&job.flags[qpos], nnmax);
    };

    // k_scan's shifted sums added up and turned into the sigmoid scalars by the FIRST wave alone (see k_strip)
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
    // ---- rounds: at most S32_ROUND active blocks each ----
    for (int ra = r_begin; ra < r_end; ra += S32_ROUND) {
        if (ra > r_begin) {                              // (a further round: wave 0 plans it; the cell lists lay over the prefix array)
            if (wv == 0) { make_prefix(false); make_schedule(ra, min(ra + S32_ROUND, r_end)); }
            __syncthreads();
        }
        const int nst = __builtin_amdgcn_readfirstlane(s_misc[0]);
        if (nst == 0) break;
        auto entry = [&](int i, unsigned &e, unsigned &j) __attribute__((always_inline)) {
            const uint2 v = s_ent[i < nst ? i : nst - 1];
            e = i < nst ? (unsigned)__builtin_amdgcn_readfirstlane((int)v.x) : (SCH_DRAIN | SCH_IDLE);
            j = (unsigned)__builtin_amdgcn_readfirstlane((int)v.y);
        };
        unsigned E0, J0, E1, J1, E2, J2;
        entry(S32_DEPTH, E0, J0); entry(S32_DEPTH + 1, E1, J1); entry(S32_DEPTH + 2, E2, J2);
        S32Regs<FLY> A0, B0, A1, B1, A2, B2;
        issue(A0, B0, J0); issue(A1, B1, J1); issue(A2, B2, J2);
        if (fold_stats && ra == r_begin) finish_stats();
        SB_T(4);                                         // march begins
        // A step: S1 -- both pieces of row wv of block i (and, without a stored plan, the list of the band cells to query) --
        // barrier, S2 -- sums along latitude (waves 8 .. 10) || queries of the block three up (waves 0 .. 7) -- barrier.
        // The three register sets take turns as three copies of the step, as in k_strip.
        auto step = [&](S32Regs<FLY> &RA, S32Regs<FLY> &RB, unsigned &E, unsigned &J, const unsigned &En, int i, int buf) __attribute__((always_inline)) {
            const unsigned ent = E, sj = J;
#if defined(SB_STAMPS) && !defined(SB_STAMPS_WIND)
            if (i < SB_NSTAMP - 10) SB_T(5 + i);         // step i begins (i >= 3)
#define SB_TS(k) do { if (i == 12) SB_T(27 + (k)); } while (0)      // the inside of one step (the tenth of the march)
#else
#define SB_TS(k) do { } while (0)
#endif
            entry(i + S32_DEPTH, E, J);                  // (consumed by `issue` below: the read travels under S1)
            const int pos = (int)(ent & 0xffffu);
            const int strip = (int)(sj >> 16), jp = (int)(sj & 0xffffu);
            const bool drain = (ent & SCH_DRAIN) != 0, idle = (ent & SCH_IDLE) != 0;
            const int qoff = drain ? HB : S32_QOFF;
            const bool qany = (ent & (drain ? SCH_Q1 : SCH_Q2)) != 0;
            if (!idle) {
                if (ent & SCH_RESTART) {
                    carry = 0;                                    // the tables start afresh (no window reaches above a run's first row)
                    if (FLY && (fold_stats || job.ngath > 0)) { sd = s_sdr[0]; rr = s_sdr[1]; }
                }
                BandWords bwd;
                const bool lister = qany && wv >= C / 2 && !cached;
                if (lister) bwd = band_issue(strip, jp - qoff);
                if (tid == S32_NT - 1) s_misc[1 + (buf == 2 ? 0 : buf + 1)] = 0;   // the next step's list starts empty
                if (!drain) {
                    const unsigned slot = slot_of(jp, wv);
                    stage(RA, slot, 0);
                    stage(RB, slot, 1);
                }
                if (lister) list_cells(strip, jp - qoff, bwd, buf);
            }
            // The stored list of the NEXT step's query, loaded AHEAD of this step's block loads: the vector-memory counter is
            // in order, and a list loaded behind them could only be waited for together with them -- every query step then
            // sat out what was left of the latency of loads meant for three steps later (round 4: 1.4 us per query step).
            unsigned qn = ~0u;
            if (cached) qn = plan_lists[((En >> SCH_QI_SHIFT) & (SB_PLAN_NQ - 1)) * (unsigned)(SW * C) + qc_off];
            SB_TS(0);                                             // staged
            issue(RA, RB, J);                                     // (the one place of this copy of the step that loads blocks)
            if (!idle) {
                SB_TS(1);                                         // loads issued, first barrier reached
                lds_barrier();
                SB_TS(2);                                         // ... passed
                if (qany && wv < C / 2) query(pos - qoff, strip, jp - qoff, buf, (ent >> SCH_QI_SHIFT) & (SB_PLAN_NQ - 1));
                if (!drain && wv >= 8 && wv <= 10) vertical(jp);
                SB_TS(3);                                         // queries / sums along latitude done
                lds_barrier();                                    // (the next step writes the ring block this step's queries read)
                SB_TS(4);                                         // second barrier passed
            }
            if (cached) qc = qn;
        };
        // (the statistics are in LDS before the first step reads them: the first real step begins with a restart)
        lds_barrier();
        for (int i = S32_DEPTH; i < nst; i += S32_DEPTH) {       // nst is a multiple of three
            step(A0, B0, E0, J0, E1, i, 0);
            step(A1, B1, E1, J1, E2, i + 1, 1);
            step(A2, B2, E2, J2, E0, i + 2, 2);
        }
        __syncthreads();                                 // the schedule and the ring are free for the next round
    }

    SB_T(5);                                             // march done
    // ---- the marked cells (rare): the global-memory search ----
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
    // strip << 16 | block of every query step's list of the plan in LDS -> s_qblk; returns the number of lists
    auto list_blocks = [&](int *s_qblk) -> int {
        __syncthreads();
        if (tid == 0) s_misc[9] = 0;
        __syncthreads();
        const int nstp = s_misc[0];
        for (int i = tid; i < nstp; i += S32_NT) {
            const uint2 v = s_ent[i];
            const bool dr = (v.x & SCH_DRAIN) != 0u;
            if (!(v.x & SCH_IDLE) && (v.x & (dr ? SCH_Q1 : SCH_Q2)) != 0u) {
                const int qi = (int)((v.x >> SCH_QI_SHIFT) & (SB_PLAN_NQ - 1));
                s_qblk[qi] = (int)((v.y & 0xffff0000u) | ((v.y & 0xffffu) - (dr ? (unsigned)HB : (unsigned)S32_QOFF)));
                atomicMax(&s_misc[9], qi + 1);
            }
        }
        __syncthreads();
        return s_misc[9];
    };
    if (s_misc[4] != 0 && cached) {                      // (uniform) a stored plan knows its marked cells: radius field zero
        const DiagJob<T> &cj = *job.cold;
        int *s_qblk = (int *)&s_cell[0][0];
        const int nq = list_blocks(s_qblk);
        for (int q = 0; q < nq; ++q) {
            const unsigned code = tid < SW * C ? plan_lists[(unsigned)q * (unsigned)(SW * C) + (unsigned)tid] : ~0u;
            if (code != ~0u && ((code >> 10) & 31u) == 0u) {
                const int blk = s_qblk[q];
                const int strip = blk >> 16, jp = blk & 0xffff;
                int nnmax = 0;
                slow_cell(cj, strip * SW + (int)(code & 31u), (jp - HB) * C + (int)((code >> 5) & 15u), nnmax);
                if (nnmax > 1) atomicMax(&job.flags[strip * npad + jp], nnmax);
            }
        }
        __syncthreads();
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
                const int x = strip * SW + (tid & (SW - 1)), y = (jp - HB) * C + (tid >> 5);
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
        // ---- a band step: thresholds, scaling and state update (ref :235-266) of every band cell this workgroup queried,
        // behind the march (see k_strip) ----
        __syncthreads();
        const DiagJob<T> &cj = *job.cold;
        auto apply = [&](unsigned o) __attribute__((always_inline)) {
            const T n_thc = __hip_atomic_load(&job.thc[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sb_trigger_update<T, false>(cj, (size_t)o, n_thc, sb_trigger_load<T>(cj, (size_t)o));
        };
        if (cached || store_lists) {
            int *s_qblk = (int *)&s_cell[0][0];          // strip << 16 | block of every query step's list
            const int nrows = list_blocks(s_qblk) * (C / 2);        // eight rows of 64 entries per list
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
                    const int x = (blk >> 16) * SW + (int)(code[k] & 31u), y = ((blk & 0xffff) - HB) * C + (int)((code[k] >> 5) & 15u);
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
                    const int x = strip * SW + (tid & (SW - 1)), y = (jp - HB) * C + (tid >> 5);
                    if (x < g.nx && y >= 0 && y < g.rows && sb_bit(job.bandbits, g.nw, x + g.h, y + g.h))
                        apply((unsigned)y * (unsigned)g.nx + (unsigned)x);
                }
            }
        }
    }
    if (job.fold && !(cached && job.lists_stand)) {
        // ---- k_wind's segment lists (see k_strip) ----
        const unsigned nseg = (unsigned)g.nyh * (unsigned)g.nw;
        const unsigned cap = (unsigned)job.seg_cap;
        for (int part = G - 1 - (int)blockIdx.x; part < SB_SEG_PARTS; part += G) {
            if (part < 0) break;
            const unsigned s0 = (unsigned)part * cap, s1 = min(s0 + cap, nseg);
            const unsigned per = (cap + S32_NT - 1) / S32_NT;
            const unsigned a0 = min(s0 + (unsigned)tid * per, s1), a1 = min(a0 + per, s1);
            int cnt = 0;
            for (unsigned sg = a0; sg < a1; ++sg) cnt += job.bandbits[sg] != 0 ? 1 : 0;
            int total;
            int at = thc_block_excl_scan<S32_NT>(cnt, s_scan, total);
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

static StripJob<float> strip32_job(const DiagJob<float> &job) {
    StripJob<float> s;
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

// single precision only: double precision keeps the tile kernel for radii beyond 16 (see the header of this file)
template <>
hipError_t sb_launch_strip32<float>(const DiagJob<float> &job, int ncu, hipStream_t st) {
    const dim3 gr(ncu), bl(S32_NT);                     // one persistent workgroup per CU
    const StripJob<float> sj = strip32_job(job);
    if (!job.wind_final) return hipErrorInvalidValue;   // (the update is k_wind's, or applied behind the march: sb_launch_diag sees to it)
    if (job.t0_fly) hipLaunchKernelGGL((k_strip32<true>), gr, bl, 0, st, sj.plan, sj.plan_gen, sj.fold_partials, ncu, sj);
    else hipLaunchKernelGGL((k_strip32<false>), gr, bl, 0, st, sj.plan, sj.plan_gen, sj.fold_partials, ncu, sj);
    return hipGetLastError();
}
template <>
hipError_t sb_launch_strip32<double>(const DiagJob<double> &, int, hipStream_t) { return hipErrorInvalidValue; }

// the block grid for a domain of nx x rows interior cells; false: the position plane cannot hold it
bool sb_strip32_shape(int nx, int rows, int *ntx, int *nty) {
    *ntx = (nx + S32_SW - 1) / S32_SW;
    *nty = (rows + S32_C - 1) / S32_C;
    return (long long)*ntx * (*nty + 2 * S32_HB) < (long long)S32_MAXW * 64 && *nty + 2 * S32_HB < 0xffff;
}
