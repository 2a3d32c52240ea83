// sb_device.hpp -- shared device-side definitions for the sea-breeze HIP kernels (gfx950).
//
// Everything here is written for CDNA4 directly: 64-lane wavefronts (ballots are 64 bit,
// one ballot word == one 64-cell longitude segment of a row), LDS-resident summed-area
// tiles, no CUDA compatibility layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SB_WAVE 64
#define SB_NSTAMP 32                 // diagnostic build: clock sums per k_thc3 workgroup

enum { SB_FLAVOUR_GENERIC = 0, SB_FLAVOUR_WRAPPER = 1 };
enum { BND_WRAPPER = 0, BND_GLOBAL = 1, BND_HALO = 2 };

// Running moments of a sample, merged pairwise (Chan et al.): the parallel replacement
// for the reference's two sequential passes (mean, then sum of squared deviations,
// ref: generic/sea_breeze_diag.f90:466-477).
struct Moments {
    double n, mean, m2, mn, mx;
};

__host__ __device__ inline Moments moments_empty() {
    Moments m;
    m.n = 0.0; m.mean = 0.0; m.m2 = 0.0;
    m.mn = 1.0e308; m.mx = -1.0e308;
    return m;
}

__host__ __device__ inline Moments moments_merge(const Moments &a, const Moments &b) {
    if (b.n == 0.0) return a;
    if (a.n == 0.0) return b;
    Moments r;
    r.n = a.n + b.n;
    const double d = b.mean - a.mean;
    const double f = b.n / r.n;
    r.mean = a.mean + d * f;
    r.m2 = a.m2 + b.m2 + d * d * a.n * f;
    r.mn = a.mn < b.mn ? a.mn : b.mn;
    r.mx = a.mx > b.mx ? a.mx : b.mx;
    return r;
}

#ifdef __HIPCC__
// Inclusive prefix sum over the 64 lanes of a wave with DPP row shifts and row broadcasts (gfx9
// family): six VALU steps, where a shuffle-based scan pays six LDS-crossbar round trips.
__device__ __forceinline__ int sb_wave_scan_add(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    return v;
}

// The same scan for doubles: the two 32-bit halves travel by DPP moves, the add is a plain v_add_f64 (DPP does
// not apply to 64-bit VALU operations).  Lanes without a source receive +0.0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double sb_dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sb_wave_scan_add_f64(double v) {
    v += sb_dpp_f64<0x111, 0xf>(v);      // row_shr:1
    v += sb_dpp_f64<0x112, 0xf>(v);      // row_shr:2
    v += sb_dpp_f64<0x114, 0xf>(v);      // row_shr:4
    v += sb_dpp_f64<0x118, 0xf>(v);      // row_shr:8
    v += sb_dpp_f64<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v += sb_dpp_f64<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3
    return v;
}
// value of lane `l` (compile-time or wave-uniform) of a double, in every lane
__device__ __forceinline__ double sb_readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

#define SB_STATS_NT 1024             // threads of every workgroup that merges moments

// k_scan's partial results are SHIFTED SUMS about a shift c common to the whole call (the first interior value of
// sigma): count, sum (x - c), sum (x - c)^2, min, max -- kept in a Moments (n, mean <- s1, m2 <- s2, mn, mx).
// Merging them is addition, where merging (n, mean, M2) triples costs a division per pair: the merge sits on
// the critical path of every k_thc3 workgroup (or of k_prep).  One conversion at the end (moments_from_shifted).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double sb_dpp_f64_self(double v) {    // lanes without a source keep their own value
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sb_wave_scan_min_f64(double v) {
    v = fmin(v, sb_dpp_f64_self<0x111, 0xf>(v));
    v = fmin(v, sb_dpp_f64_self<0x112, 0xf>(v));
    v = fmin(v, sb_dpp_f64_self<0x114, 0xf>(v));
    v = fmin(v, sb_dpp_f64_self<0x118, 0xf>(v));
    v = fmin(v, sb_dpp_f64_self<0x142, 0xa>(v));
    v = fmin(v, sb_dpp_f64_self<0x143, 0xc>(v));
    return v;
}
// totals of the NW waves' shifted sums, in every thread: wave totals to LDS, one barrier (the caller's, if it has other
// business with it), then the wave totals added in wave order -- so that workgroups of different sizes agree bit for
// bit as long as the waves beyond the data hold the empty set
__device__ __forceinline__ void wave_total_shifted_store(Moments v, Moments *wpart) {
    v.n = sb_wave_scan_add_f64(v.n);
    v.mean = sb_wave_scan_add_f64(v.mean);
    v.m2 = sb_wave_scan_add_f64(v.m2);
    v.mn = sb_wave_scan_min_f64(v.mn);
    v.mx = -sb_wave_scan_min_f64(-v.mx);
    if ((threadIdx.x & 63) == 63) wpart[threadIdx.x >> 6] = v;
}
template <int NW>
__device__ __forceinline__ Moments block_total_shifted_finish(const Moments *wpart) {
    Moments r = wpart[0];
#pragma unroll 4                                 // (all NW x 5 values at once would cost up to 160 registers)
    for (int w = 1; w < NW; ++w) {
        const Moments o = wpart[w];
        r.n += o.n; r.mean += o.mean; r.m2 += o.m2;
        r.mn = fmin(r.mn, o.mn); r.mx = fmax(r.mx, o.mx);
    }
    return r;
}
template <int NW>
__device__ __forceinline__ Moments block_total_shifted(Moments v, Moments *wpart) {
    wave_total_shifted_store(v, wpart);
    __syncthreads();
    return block_total_shifted_finish<NW>(wpart);
}
// shifted sums -> (n, mean, M2, min, max)
__host__ __device__ inline Moments moments_of_shifted(double c, const Moments &t) {
    Moments acc = t;
    if (t.n > 0.0) {
        acc.mean = c + t.mean / t.n;
        acc.m2 = t.m2 - t.mean * t.mean / t.n;
    } else {
        acc.mean = 0.0; acc.m2 = 0.0;
    }
    return acc;
}

__device__ __forceinline__ Moments wave_merge(Moments m) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Moments o;
        o.n = __shfl_down(m.n, off);
        o.mean = __shfl_down(m.mean, off);
        o.m2 = __shfl_down(m.m2, off);
        o.mn = __shfl_down(m.mn, off);
        o.mx = __shfl_down(m.mx, off);
        m = moments_merge(m, o);
    }
    return m;
}

// merge across the NW waves of a workgroup; the result is valid in thread 0.  The tree is the same for every NW
// as long as the waves beyond the data hold empty moments (merging an empty set is the identity, bit for bit).
template <int NW>
__device__ __forceinline__ Moments block_merge_w(Moments m, Moments *wpart) {
    m = wave_merge(m);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();                             // wpart may still be read from a previous use
    if (lane == 0) wpart[wv] = m;
    __syncthreads();
    Moments r = moments_empty();
    if (threadIdx.x < SB_WAVE) {
        if (threadIdx.x < NW) r = wpart[threadIdx.x];
        r = wave_merge(r);
    }
    return r;
}
// ... of a 1024-thread workgroup
__device__ __forceinline__ Moments block_merge(Moments m, Moments *wpart) {
    return block_merge_w<SB_STATS_NT / SB_WAVE>(m, wpart);
}

// std = 2/sqrt(var/N), r = (max-min)/4 in the working precision
// ref: generic/sea_breeze_diag.f90:478-479 (N = nlons*nlats, the merged sample count)
template <typename T>
__device__ __forceinline__ void sigmoid_scalars(const Moments &m, T *__restrict__ stats) {
    const T var = (T)m.m2;
    const T cnt = (T)m.n;
    stats[0] = T(2) / sqrt(var / cnt);
    stats[1] = ((T)m.mx - (T)m.mn) / T(4);
    stats[2] = (T)m.mean;
    stats[3] = var;
}

#endif  // __HIPCC__

// Grid geometry shared by every kernel of one call.
struct Geo {
    int nx, ny;      // interior cells (lon, lat)
    int h;           // ghost-cell width around the 2-D input fields (0 unless bnd == BND_HALO)
    int nxh, nyh;    // nx + 2h, ny + 2h
    int nw;          // 64-bit words per row of the bit planes (ceil(nxh / 64))
    int bnd;         // BND_*
    int rows;        // rows processed: ny (generic) or ny-1 (wrapper, ref: seabreeze_diag_python.f90:165)
    int band;        // BND_HALO only, a latitude band of a multi-GPU run (GEO_BAND_*): which ghost cells of the per-step field
                     // theta the kernels do NOT read because they follow from its interior -- the east-west ghost columns
                     // (the band holds full longitude circles: the periodic wrap is index arithmetic) and the ghost rows
                     // beyond a pole (the reference's latitude clamp: the edge row again).  0: every ghost cell is read as
                     // the caller filled it (the UM layout, a band run that fills them: sb_swap_bounds / sb_fill_ghosts).
};
enum { GEO_BAND_EW = 1, GEO_BAND_SOUTH = 2, GEO_BAND_NORTH = 4 };

// Map an interior 0-based (xs, ys), possibly outside [0,nx) x [0,ny), to coordinates in
// the (nxh, nyh) arrays.  Returns false when the cell does not exist (BND_HALO only).
__device__ __forceinline__ bool sb_map_cell(const Geo &g, int xs, int ys, int &X, int &Y) {
    if (g.bnd == BND_HALO) {
        if (g.band & GEO_BAND_EW) {
            int m = xs % g.nx;
            xs = m < 0 ? m + g.nx : m;
        }
        if ((g.band & GEO_BAND_SOUTH) && ys < 0) ys = 0;
        if ((g.band & GEO_BAND_NORTH) && ys >= g.ny) ys = g.ny - 1;
        X = xs + g.h;
        Y = ys + g.h;
        return X >= 0 && X < g.nxh && Y >= 0 && Y < g.nyh;
    }
    int yy = ys < 0 ? 0 : (ys >= g.ny ? g.ny - 1 : ys);    // lat index is bounded
    Y = yy;
    if (g.bnd == BND_WRAPPER) {
        // kj = max(1, modulo(jj, nlons)) with jj = xs+1 (1-based): 0 and nlons both -> 1
        int m = (xs + 1) % g.nx;
        if (m < 0) m += g.nx;
        X = (m < 1 ? 1 : m) - 1;
    } else {
        int m = xs % g.nx;
        if (m < 0) m += g.nx;
        X = m;
    }
    return true;
}

// One 64-cell longitude segment that holds band cells, as k_prep lists it for k_wind: the index of the
// segment in the bit planes (row * nw + word) and its band word.
struct SbSegEntry {
    uint64_t word;
    uint32_t seg, pad;
};
// the strip kernel's plan, per workgroup: 16 ints (call that stored it or 0, steps, first and last rank of its share),
// the steps (8 bytes each), the cell lists of up to SB_PLAN_NQ query steps (8 x 64 entries of 32 bits, all ones: none)
#define SB_PLAN_SCHED 384
#define SB_PLAN_NQ 64
#define SB_PLAN_ENT_OFF 64
#define SB_PLAN_LIST_OFF (SB_PLAN_ENT_OFF + 8 * SB_PLAN_SCHED)
#define SB_PLAN_STRIDE (SB_PLAN_LIST_OFF + 2048 * SB_PLAN_NQ)
#define SB_SEG_PARTS 64              // k_prep workgroups that compact the segment list, one sub-list each

template <typename T>
struct DiagJob {
    Geo g;
    int nz;                 // levels of p/u/v (generic) or nps (wrapper)
    int flavour;            // SB_FLAVOUR_*
    int tn;                 // timestep_number
    int refresh;            // modulo(real(tn)*timestep, target_time) < 0.0001, evaluated on the host in T
    T target_plev, thr_wind, thr_dir, thr_ch, thr_thc, maxdist, fill;
    // inputs
    const T *p, *u, *v;             // generic: (nx,ny,nz); wrapper: p(nps), u/v (nx,ny,nps)
    const T *theta, *mask, *z, *sigma;   // (nxh, nyh); mask may sit in a wider frame (UM layout): row pitch mask_ld,
    int mask_ld;                    //   first cell of the (nxh, nyh) frame at mask[mask_off]
    unsigned mask_off;
    int level_rule;                 // 0: first minimum of |p - target| over all levels (generic :223);
                                    // 1: the UM copy's upward walk that stops at the first increase (UM :265-274)
    // state / outputs, (nx, ny)
    T *ws, *wd, *thc, *sb_con;
    T *out;                         // wrapper: (nx,ny,4) packed output, else nullptr
    // workspace
    T *nws, *nwd;                   // (nx, ny): this call's wind speed / direction at band cells (k_wind -> k_thc3, band steps only)
    int *next_flags;                // the other tile-flag buffer: k_wind clears it for the next call
    int next_flags_n;
    int wind_final;                 // 1: k_wind applies thresholds + state update itself (k_thc3 ran before it and
                                    //    left thc); 0: k_wind leaves nws/nwd and the contrast kernel applies them
    int t0_fly;                     // 1: k_thc3 derives t0 from theta,z,sigma while staging; 0: reads the t0 workspace
    T *t0;                          // (nxh, nyh)
    uint64_t *bandbits;             // nyh * nw words: interior cells with |mask| <= maxdist
    uint64_t *clsbits;              // nyh * nw words: mask >= 0 ("land side")
    const T *stats;                 // [0]=std  [1]=r  (sigmoid scalars)
    int thc_ty, thc_ntx, thc_nty;   // contrast-kernel tile rows and tile-grid shape
    int thc_txs;                    // log2 of the tile width (5: 32 longitudes)
    int tile_sx, tile_sy, tile_off; // flag of tile (column tx, row ty) = tile_nnmax[tx * tile_sx + ty * tile_sy + tile_off]:
                                    //   row-major for the tile kernel; strip-major with a virtual block above and below
                                    //   every strip for the strip kernel (sb_strip_kernel.hip)
    int strip;                      // 1: the strip kernel runs the contrast (LDS halo <= 16); 2: the 96-column strip kernel (radii
                                    //    up to 31, single precision: sb_strip32_kernel.hip)
    // the strip kernel's plan (its share of the blocks, the order of its steps, the band cells of every block it
    // queries) depends on the band plane alone; it is kept from call to call and remade when k_scan finds that the
    // plane changed -- see sb_strip_kernel.hip
    char *plan;                     // per workgroup: header, steps, cell lists (SB_PLAN_STRIDE bytes)
    int *plan_gen;                  // number of the last call whose band plane differed from that of the call before
    int call_id;                    // number of this call (> 0, ascending)
    int plan_use;                   // 1: a plan stored by a call no older than *plan_gen may be used
    Moments *moments_out;           // band step: k_scan's last workgroup merges the partials and leaves this band's moments
    int *stats_ticket;              //   here (for the all-gather); a device word that is zero between launches
    int strip_update;               // band step on the strip kernel: it applies thresholds and state update behind its march
    int seg_trust;                  // 1: k_wind's segment lists are those the strip kernel of the call before compacted;
                                    //    they stand unless k_scan found the planes changed in this call (*plan_gen == call_id)
    int *tile_nnmax;                // per contrast tile: 0 = no band cell; k_scan raises 1, k_thc3 leaves the largest radius
    int *ticket;                    // spare device word (zeroed by k_scan)
    // lists k_prep compacts between k_scan and the kernels that consume them
    int *tile_list;                 // [0] = number of active tiles, [1..] their indices in row-major order, then
    int tile_pad;                   //   tile_pad entries of -1 (two per k_thc3 workgroup)
    SbSegEntry *seg_list;           // SB_SEG_PARTS sub-lists of seg_cap entries: the segments that hold band cells
    int *seg_count;                 // entries in each sub-list
    int seg_cap;
    // single-domain host-model calls fold k_prep's work into k_thc3 (job.fold): k_scan's partial moments are merged
    // and the active tiles picked by every workgroup for itself, the segment lists compacted in its last workgroups
    const Moments *gath;            // band step: the moments gathered from every rank (ngath of them), merged by the first wave
    int ngath;                      //   of every k_thc3 workgroup in k_merge_moments' tree; 0: the scalars in stats stand
    int fold;
    int lists_stand;                // the segment lists of an earlier call are in place and belong to the planes the stored plan
                                    // belongs to: a march by that plan does not compact them again
    const Moments *fold_partials;   // k_scan's per-workgroup moments (fold_nparts of them; 0: the scalars in stats stand)
    int fold_nparts;
    T *stats_out;                   // where workgroup 0 publishes the sigmoid scalars
    DiagJob<T> *self;               // where k_scan leaves a copy of this job in device memory: kernels that take only the
                                    // hot part of it by value (k_strip) read the rest from there, rarely
    int *counters;                  // [0] cells on the global-memory path, [1] one-class cells
    long long *stamps;              // diagnostic build (-DSB_STAMPS) only: SB_NSTAMP clock sums per k_thc3 workgroup
};

// What the strip contrast kernel needs on every step, passed by value (about 50 scalar registers; the whole DiagJob
// by value costs the kernel some 100 spilled scalar registers).
template <typename T>
struct StripJob {
    Geo g;
    const T *theta, *z, *sigma;     // theta: the t0 plane unless t0 is derived while staging
    const uint64_t *clsbits, *bandbits;
    T *thc;
    int *flags;                     // strip-major block flags (DiagJob::tile_nnmax)
    int ntx, nty;                   // strips, blocks per strip
    int fold, fold_nparts, ngath, seg_cap;
    int lists_stand;                // as DiagJob
    int update;                     // band step: thresholds and state update of this workgroup's band cells behind the march
    const T *stats;
    T *stats_out;
    const Moments *fold_partials, *gath;
    SbSegEntry *seg_list;
    int *seg_count;
    const DiagJob<T> *cold;         // the whole job in device memory (k_scan wrote it): slow path, band-step update
    char *plan;                     // as DiagJob
    const int *plan_gen;
    int call_id, plan_use;
    long long *stamps;              // diagnostic build only
};

__device__ __forceinline__ int sb_bit(const uint64_t *bits, int nw, int X, int Y) {
    return (int)((bits[(size_t)Y * nw + (X >> 6)] >> (X & 63)) & 1ull);
}

// Fortran MODULO(a, p) for reals as flang evaluates it: fmod, then fold into [0,p) for p > 0.
template <typename T>
__host__ __device__ inline T sb_modulo(T a, T p) {
    // fmod is exact, and so are these two cases of it: |a| < p leaves a, and p <= a < 2p is a - p
    // (Sterbenz); the wind-direction difference of ref :245 always lands in one of them
    T r;
    if (p > T(0) && fabs(a) < p) r = a;
    else if (p > T(0) && a >= p && a < p + p) r = a - p;
    else r = fmod(a, p);
    if ((a < T(0)) != (p < T(0))) {
        if (r == T(0)) r = -r; else r += p;
    }
    return r;
}

// 1/(1+exp(y)) in the working precision.  For doubles: exp by argument reduction to |r| <= ln2/2 (two-part ln2,
// fused), a degree-11 Taylor polynomial and v_ldexp_f64; the reciprocal by v_rcp_f64 and two Newton steps.  About 30
// instructions against about 90 for libm's exp plus an IEEE division, and within a few ulp of them.
// ONE evaluation order for every kernel (the strip kernel's copy, strip_logistic_of_neg, differs only in where its
// constants come from): t0 is the same to the last bit whichever kernel forms it.
template <typename T>
__device__ __forceinline__ T sb_logistic_of_neg(T y) {        // returns 1 / (1 + exp(y))
    return T(1) / (T(1) + exp(y));
}
template <>
__device__ __forceinline__ double sb_logistic_of_neg<double>(double y) {
    y = fmin(fmax(y, -708.0), 700.0);                         // beyond: 1 or 0 to the last bit
    const double n = __builtin_rint(y * 1.4426950408889634);
    double r = __builtin_fma(-n, 0.6931471805599453, y);
    r = __builtin_fma(-n, 2.3190468138462996e-17, r);         // |r| <= ln2/2
    // exp(r) by its Taylor polynomial of degree 11 (truncation r^12/12! < 7e-15 relative) as E(r^2) + r O(r^2): two
    // independent Horner chains of five fused multiply-adds (six deep in all, where one chain would be eleven)
    const double r2 = r * r;
    double e = __builtin_fma(r2, 2.755731922398589e-07, 2.48015873015873e-05);               // 1/10!, 1/8!
    double o = __builtin_fma(r2, 2.505210838544172e-08, 2.7557319223985893e-06);             // 1/11!, 1/9!
    e = __builtin_fma(e, r2, 1.388888888888889e-03);   o = __builtin_fma(o, r2, 1.984126984126984e-04);      // 1/6!, 1/7!
    e = __builtin_fma(e, r2, 4.1666666666666664e-02);  o = __builtin_fma(o, r2, 8.333333333333333e-03);      // 1/4!, 1/5!
    e = __builtin_fma(e, r2, 0.5);                     o = __builtin_fma(o, r2, 1.6666666666666666e-01);      // 1/2!, 1/3!
    e = __builtin_fma(e, r2, 1.0);                     o = __builtin_fma(o, r2, 1.0);
    const double p = __builtin_fma(o, r, e);
    const double x = 1.0 + ldexp(p, (int)n);
    // v_rcp_f64 is good to about 2^-24 relative (2^29 ulp); a Newton step squares the error: 4e-15 after one, rounding
    // level after two
    double q = __builtin_amdgcn_rcp(x);
    q = __builtin_fma(q, __builtin_fma(-x, q, 1.0), q);
    q = __builtin_fma(q, __builtin_fma(-x, q, 1.0), q);
    return q;
}

// t0 = theta - (gmma*z)*sigmoid(sigma)   ref: generic/sea_breeze_diag.f90:166-167,478-480
// At sea level (z == 0) the product is a signed zero whatever the sigmoid, so theta comes
// back bit for bit and the exp is skipped -- most cells of a coastal tile are ocean.
template <typename T>
__device__ __forceinline__ T sb_t0(T theta, T z, T sigma, T sd, T r) {
    if (z == T(0)) return theta;
    return theta - ((T(-0.0060956) * z) * sb_logistic_of_neg<T>(-sd * (sigma - r)));
}

// Thresholds, scaling and state update of one band cell at linear index o, given this
// call's contrast n_thc and the wind speed/direction k_wind left in the workspace.
// ref: generic/sea_breeze_diag.f90:235-266, seabreeze_diag_python.f90:236-280
template <typename T>
struct SbCellState {
    T n_ws, n_wd, ws, wd;           // this call's wind (from k_wind) and the carried state
};

// the loads of sb_trigger_update, separable so a kernel can issue them early
template <typename T>
__device__ __forceinline__ SbCellState<T> sb_trigger_load(const DiagJob<T> &job, size_t o) {
    SbCellState<T> s;
    s.n_ws = job.nws[o]; s.n_wd = job.nwd[o]; s.ws = job.ws[o]; s.wd = job.wd[o];
    return s;
}

template <typename T, bool STORE_THC = true>      // STORE_THC = false: thc already holds n_thc (k_thc3 wrote it)
__device__ __forceinline__ void sb_trigger_update(const DiagJob<T> &job, size_t o, T n_thc,
                                                  const SbCellState<T> &st) {
    const T n_ws = st.n_ws, n_wd = st.n_wd;
    T ws_old = st.ws, wd_old = st.wd;
    if (job.tn < 2) { ws_old = n_ws; wd_old = n_wd; }            // ref :235-239
    const T thc_abs = fabs(n_thc);
    const T mws = (ws_old + n_ws) / T(2);
    const T dws = fabs(ws_old - n_ws);
    const T dwd = fabs(sb_modulo<T>((wd_old - n_wd) + T(180), T(360)) - T(180));
    T sb = T(0);
    if (dwd < job.thr_dir && dws < job.thr_ch && mws < job.thr_wind && thc_abs > job.thr_thc) {
        const T scale_wind = (job.thr_wind - mws) / (mws > T(1) ? mws : T(1));
        const T scale_thc = (thc_abs - job.thr_thc) / n_thc;
        sb = scale_thc * scale_wind;
    }
    if constexpr (STORE_THC) job.thc[o] = n_thc;                 // ref :262
    if (job.flavour == SB_FLAVOUR_GENERIC) {
        job.sb_con[o] = sb;
        job.ws[o] = n_ws;                                        // ref :261 (every call)
        if (job.refresh) job.wd[o] = n_wd;                       // ref :264-266
        else if (job.tn < 2) job.wd[o] = wd_old;
    } else {
        const size_t pl = (size_t)job.g.nx * job.g.ny;
        T ws_new = ws_old, wd_new = wd_old;                      // ref: seabreeze_diag_python.f90:268-280
        if (job.refresh) { ws_new = n_ws; wd_new = n_wd; }
        if (job.refresh || job.tn < 2) { job.ws[o] = ws_new; job.wd[o] = wd_new; }
        job.out[o] = sb;
        job.out[2 * pl + o] = ws_new;
        job.out[3 * pl + o] = wd_new;
    }
}
