// sb_diag_kernels.hip -- the seabreeze_diag hot path as CDNA4 (gfx950) kernels.
//
// One call of seabreeze_diag / diag (ref: generic/sea_breeze_diag.f90:55-271,
// python_wrapper/seabreezediag/seabreeze_diag_python.f90:49-285) is four launches on one
// stream (five for the f2py flavour):
//
//   k_scan   one pass over sigma and mask: per-workgroup moments of sigma, the band and
//            land-side bit planes, tile flags, and the fill value for every cell outside
//            the coastal band; k_moments_final merges the moments -> std, r   [HBM stream]
//   k_wind   per band cell (dense lanes): level nearest target_plev in the p column, wind
//            speed / direction                                               [HBM gather]
//   k_t0     f2py flavour only: the t0 plane is an output there               [HBM stream]
//   k_thc    (sb_thc_kernel.hip) per active 64 x TY tile: t0 and its summed-area tables in
//            LDS, bisection for the window radius -> thc, then thresholds, scaling and
//            state update -> sb_con                                           [LDS]
//
// Memory-bound integer/fp64 work: no MFMA anywhere.
#include "sb_device.hpp"
#include "sb_launch.hpp"
#include <cstdlib>

// ------------------------------------------------------------------------------------
// moments helpers
// ------------------------------------------------------------------------------------
#define STATS_NT SB_STATS_NT

// Shifted sums of one thread -> (n, mean, M2)
__device__ __forceinline__ Moments moments_from_shifted(double c, double s1, double s2, double mn, double mx,
                                                         int cnt) {
    Moments acc = moments_empty();
    if (cnt > 0) {
        acc.n = (double)cnt;
        acc.mean = c + s1 / acc.n;
        acc.m2 = s2 - s1 * s1 / acc.n;
        acc.mn = mn;
        acc.mx = mx;
    }
    return acc;
}

// Workgroup partial -> global partials.  The cross-workgroup merge is a kernel of its own
// (k_moments_final): an in-kernel ticket would need an agent-scope release per workgroup,
// and in a kernel that also writes tens of MB (k_scan's fill) every such release drains
// the XCD's dirty L2 -- measured 20 us for 512 workgroups, against ~2 us for the boundary.
__device__ __forceinline__ void store_partial(Moments acc, Moments *wpart, Moments *__restrict__ partials) {
    acc = block_merge(acc, wpart);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// One workgroup merges the partials in index order (a fixed tree: the result does not
// depend on arrival order) and publishes either the moments themselves (band-local, for
// the multi-GPU gather) or the sigmoid scalars.
template <typename T>
__global__ __launch_bounds__(STATS_NT) void k_moments_final(const Moments *__restrict__ partials, int nparts,
                                                            T *__restrict__ stats,
                                                            Moments *__restrict__ moments_out) {
    __shared__ Moments wpart[STATS_NT / SB_WAVE];
    Moments m = moments_empty();
    for (int b = threadIdx.x; b < nparts; b += STATS_NT) m = moments_merge(m, partials[b]);
    m = block_merge(m, wpart);
    if (threadIdx.x == 0) {
        if (moments_out) *moments_out = m;
        else sigmoid_scalars<T>(m, stats);
    }
}

// ------------------------------------------------------------------------------------
// k_stats: statistics of a plain 2-D field (stand-alone sigmoid; band-local moments for
// the multi-GPU gather).  One workgroup per CU, 8 independent loads per thread in flight;
// sums about a common shift c (first element) -> (n, mean, M2) once per thread, merged
// pairwise (Chan) from there on.  Replaces the reference's two sequential passes
// (ref: generic/sea_breeze_diag.f90:466-477).
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(STATS_NT) void k_stats(const T *__restrict__ ary, int nx, int ny, int ld,
                                                    size_t off0, Moments *__restrict__ partials) {
    const unsigned n = (unsigned)nx * (unsigned)ny;
    const unsigned stride = gridDim.x * STATS_NT;
    const bool flat = (ld == nx);
    const double c = (double)ary[off0];
    double s1 = 0.0, s2 = 0.0, mn = 1.0e308, mx = -1.0e308;
    int cnt = 0;
    for (unsigned base = blockIdx.x * STATS_NT + threadIdx.x; base < n; base += 8 * stride) {
        double x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const unsigned i = base + q * stride;
            x[q] = c;
            if (i < n) {
                const size_t idx = flat ? (size_t)i : (size_t)(i / (unsigned)nx) * ld + (i % (unsigned)nx);
                x[q] = (double)ary[off0 + idx];
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (base + q * stride < n) {
                const double d = x[q] - c;
                s1 += d;
                s2 += d * d;
                mn = x[q] < mn ? x[q] : mn;
                mx = x[q] > mx ? x[q] : mx;
                ++cnt;
            }
        }
    }
    __shared__ Moments wpart[STATS_NT / SB_WAVE];
    store_partial(moments_from_shifted(c, s1, s2, mn, mx, cnt), wpart, partials);
}

// Merge the moments gathered from every latitude band (one entry per rank) and derive the
// sigmoid scalars; a single wave.  The merge order is rank order on every rank, so all
// ranks hold bit-identical scalars.
template <typename T>
__global__ __launch_bounds__(SB_WAVE) void k_merge_moments(const Moments *__restrict__ parts, int nparts,
                                                           T *__restrict__ stats) {
    Moments m = moments_empty();
    for (int b = threadIdx.x; b < nparts; b += SB_WAVE) m = moments_merge(m, parts[b]);
    m = wave_merge(m);
    if (threadIdx.x == 0) sigmoid_scalars<T>(m, stats);
}

// sm = 1/(1+exp(-std*(ary-r)))   ref: generic/sea_breeze_diag.f90:480
template <typename T>
__global__ __launch_bounds__(256) void k_sigmoid_apply(const T *__restrict__ ary, T *__restrict__ sm,
                                                       size_t n, const T *__restrict__ stats) {
    const T sd = stats[0], r = stats[1];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        sm[i] = T(1) / (T(1) + exp(-sd * (ary[i] - r)));
}

// ------------------------------------------------------------------------------------
// k_scan: one persistent 1024-thread workgroup per CU streams sigma and mask.  A wave
// owns aligned 64-cell longitude segments (four per trip, eight loads in flight), so its
// ballots are exactly the words of the bit planes.
//   * sigma  -> shifted sums -> one (n, mean, M2, min, max) partial per workgroup
//   * mask   -> land-side bit  mask >= 0              ref: generic/sea_breeze_diag.f90:182,200
//            -> band bit  !(|mask| > maxdist)         ref :174
//            -> flag of the k_thc tile(s) the segment's band cells fall in
//            -> fill value outside the band           ref :176 / seabreeze_diag_python.f90:173,279-280
// ------------------------------------------------------------------------------------
template <typename T, int SPT, bool WR, bool ST>      // WR: f2py flavour; ST: accumulate sigma's moments
__global__ __launch_bounds__(STATS_NT) void k_scan(DiagJob<T> job, Moments *__restrict__ partials) {
    constexpr bool wrapper = WR, do_stats = ST;
    const Geo g = job.g;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int NWV = STATS_NT / SB_WAVE;              // waves per workgroup; SPT segments per trip
    const unsigned nseg = (unsigned)g.nyh * (unsigned)g.nw;
    // wave w of the grid takes segments w, w + W, w + 2W, ... (W = waves in the grid): every
    // wave gets floor or ceil of nseg/W segments, and neighbouring waves read neighbouring memory
    const unsigned nwaves = gridDim.x * NWV;
    const size_t pl = (size_t)g.nx * g.ny;
    if (blockIdx.x == 0 && threadIdx.x == 0 && job.ticket) *job.ticket = 0;   // k_thc2's tile dealing starts over
    const double c = do_stats ? (double)job.sigma[(size_t)g.h * g.nxh + g.h] : 0.0;
    double s1 = 0.0, s2 = 0.0, mn = 1.0e308, mx = -1.0e308;
    int cnt = 0;

    // (row, word) of the wave's first segment and of the stride, kept wave-uniform: the
    // segment walk then needs no division (this kernel is issue-bound, not byte-bound)
    const unsigned w0 = __builtin_amdgcn_readfirstlane(blockIdx.x * NWV + wv);
    const unsigned unw = (unsigned)g.nw;
    unsigned Yc = w0 / unw, Xc = w0 - Yc * unw;
    const unsigned dY = nwaves / unw, dX = nwaves - dY * unw;
    const unsigned nxh = (unsigned)g.nxh, unx = (unsigned)g.nx;

    // One trip = SPT segments.  The loads of the NEXT trip are issued before this trip's values are
    // used, unconditionally and from clamped addresses: a load under a branch is waited for inside the
    // branch (one round trip per segment), and a load issued behind this trip's stores would make the
    // next wait sit out those stores as well (loads and stores share the in-order vmcnt counter).
    struct Trip {
        T sg[SPT], mk[SPT], ws[SPT], wd[SPT];
        unsigned Y[SPT], W[SPT];
    };
    auto issue = [&](unsigned s0, Trip &t) {
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const unsigned seg = s0 + q * nwaves;
            t.Y[q] = Yc;
            t.W[q] = Xc;
            const int X = (int)(Xc * 64u) + lane;
            const int xi = X - g.h, yi = (int)Yc - g.h;
            const bool in = seg < nseg && X < g.nxh;
            const bool interior = in && xi >= 0 && xi < g.nx && yi >= 0 && yi < g.ny;
            const unsigned idx = in ? Yc * nxh + (unsigned)X : 0u;
            t.mk[q] = job.mask[idx];
            t.sg[q] = T(0); t.ws[q] = T(0); t.wd[q] = T(0);
            if (do_stats) t.sg[q] = job.sigma[idx];                      // wave-uniform condition
            if (wrapper) {                                               // wave-uniform condition
                const unsigned o = (interior && yi < g.rows) ? (unsigned)yi * unx + (unsigned)xi : 0u;
                t.ws[q] = job.ws[o];
                t.wd[q] = job.wd[o];
            }
            Yc += dY; Xc += dX;
            if (Xc >= unw) { Xc -= unw; Yc += 1; }
        }
    };
    auto process = [&](unsigned s0, const Trip &t) {
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const unsigned seg = s0 + q * nwaves;
            if (seg >= nseg) break;                              // wave-uniform
            const int X = (int)(t.W[q] * 64u) + lane, xi = X - g.h, yi = (int)t.Y[q] - g.h;
            const bool in = X < g.nxh;
            const bool interior = in && xi >= 0 && xi < g.nx && yi >= 0 && yi < g.ny;
            if (do_stats && interior) {
                const double x = (double)t.sg[q], d = x - c;
                s1 += d;
                s2 = __builtin_fma(d, d, s2);
                mn = fmin(mn, x);
                mx = fmax(mx, x);
                ++cnt;
            }
            const bool cls = in && (t.mk[q] >= T(0));
            const bool band = interior && yi < g.rows && !(fabs(t.mk[q]) > job.maxdist);
            const uint64_t wc = __ballot(cls);
            const uint64_t wb = __ballot(band);
            if (lane == 0) {
                job.clsbits[seg] = wc;
                job.bandbits[seg] = wb;
            }
            if (wb) {                                            // wave-uniform
                // the tile columns the segment's band cells fall in (two of 32 cells, three when the ghost
                // width is not a multiple of the tile width): lane j looks at the bits of column tA + j
                // in the ballot and raises that tile's flag -- one exec-masked store, no further ballots
                const int txs = job.thc_txs, tw = 1 << txs;
                const int xi0 = (int)(t.W[q] * 64u) - g.h;       // interior longitude of lane 0 (may be negative)
                const int tA = xi0 >> txs;
                int lo = ((tA + lane) << txs) - xi0, hi = lo + tw;
                lo = lo < 0 ? 0 : lo;
                hi = hi > 64 ? 64 : hi;
                if (lane <= (64 >> txs) && lo < hi) {
                    const uint64_t m = (hi - lo == 64 ? ~0ull : ((1ull << (hi - lo)) - 1ull)) << lo;
                    if (wb & m) job.tile_nnmax[(yi / job.thc_ty) * job.thc_ntx + tA + lane] = 1;   // benign duplicates
                }
            }
            if (interior && yi < g.rows && !band) {
                const unsigned o = (unsigned)yi * unx + (unsigned)xi;
                if (!wrapper) job.sb_con[o] = job.fill;
                else {
                    job.out[o] = job.fill;
                    job.out[2 * pl + o] = t.ws[q];
                    job.out[3 * pl + o] = t.wd[q];
                }
            }
        }
    };
    // two register sets, alternating: the loads of trip n+1 are issued before trip n is used
    const unsigned step = SPT * nwaves;
    Trip ta, tb;
    unsigned s0 = w0;
    if (s0 < nseg) issue(s0, ta);
    while (s0 < nseg) {
        if (s0 + step < nseg) issue(s0 + step, tb);              // wave-uniform
        process(s0, ta);
        s0 += step;
        if (s0 >= nseg) break;
        if (s0 + step < nseg) issue(s0 + step, ta);
        process(s0, tb);
        s0 += step;
    }
    if (!do_stats) return;
    __shared__ Moments wpart[STATS_NT / SB_WAVE];
    store_partial(moments_from_shifted(c, s1, s2, mn, mx, cnt), wpart, partials);
}

// ------------------------------------------------------------------------------------
// k_t0: t0 = theta - (gmma*z)*sigmoid(sigma) for every cell   ref: generic/...:167
// Used by the f2py flavour, whose t0 plane is an output (seabreeze_diag_python.f90:278);
// the host-model flavour derives t0 inside k_thc while staging instead.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_t0(DiagJob<T> job) {
    const Geo g = job.g;
    const int X = blockIdx.x * 256 + threadIdx.x, Y = blockIdx.y;
    if (X >= g.nxh) return;
    const size_t idx = (size_t)Y * g.nxh + X;
    const T sd = job.stats[0], r = job.stats[1];
    const T t0v = sb_t0<T>(job.theta[idx], job.z[idx], job.sigma[idx], sd, r);
    job.t0[idx] = t0v;
    const int xi = X - g.h, yi = Y - g.h;
    if (job.out && xi >= 0 && xi < g.nx && yi >= 0 && yi < g.rows)
        job.out[(size_t)g.nx * g.ny + (size_t)yi * g.nx + xi] = t0v;
}

// ------------------------------------------------------------------------------------
// Dense band-cell enumeration inside a workgroup: the workgroup covers 256 consecutive
// longitudes of one latitude row; the band cells among them are compacted (ballot +
// popcount prefix) so that thread i owns the i-th band cell.  Waves are full however
// ragged the band is, surplus waves leave, and consecutive lanes still touch consecutive
// longitudes inside a run.  Returns the number of band cells; s_x[i] is their longitude.
// ------------------------------------------------------------------------------------
#define ROW_NT 256

// streaming load: non-temporal for data read once per call (the p columns; plain loads
// measured 17 % slower for k_wind)
template <typename T, bool NTL>
__device__ __forceinline__ T sb_ld(const T *p) {
    if constexpr (NTL) return __builtin_nontemporal_load(p);
    else return *p;
}

template <int CPW>
__device__ __forceinline__ int block_band_cells(const uint64_t *__restrict__ bandbits, const Geo &g, int y,
                                                unsigned short *s_x, int *s_wcnt) {
    // the workgroup covers CPW * ROW_NT consecutive longitudes: CPW chunks of ROW_NT, a wave on 64 of them
    constexpr int NW = ROW_NT / SB_WAVE;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    bool band[CPW];
    uint64_t bm[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int x = (blockIdx.x * CPW + c) * ROW_NT + tid;
        band[c] = x < g.nx && sb_bit(bandbits, g.nw, (x < g.nx ? x : 0) + g.h, y + g.h) != 0;
        bm[c] = __ballot(band[c]);
        if (lane == 0) s_wcnt[c * NW + wv] = __popcll(bm[c]);
    }
    __syncthreads();
    int total = 0, before[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        before[c] = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int cw = s_wcnt[c * NW + w];
            before[c] += total * (w == 0 ? 1 : 0);               // entries of the chunks before this one
            before[c] += (w < wv) ? cw : 0;
            total += cw;
        }
    }
    if (total == 0) return 0;
#pragma unroll
    for (int c = 0; c < CPW; ++c)
        if (band[c]) s_x[before[c] + __popcll(bm[c] & ((1ull << lane) - 1ull))] = (unsigned short)(c * ROW_NT + tid);
    __syncthreads();
    return total;
}

// ------------------------------------------------------------------------------------
// k_wind: level nearest the target pressure + wind speed/direction for every band cell.
// Band cells walk their p column (nz planes, stride nx*ny) with UN independent
// non-temporal loads in flight and keep the first minimum of |p - target|
// ref: generic/sea_breeze_diag.f90:223-227, seabreeze_diag_python.f90:228-233 (1-D p: one
// level for all cells).  A workgroup covers CPW * 256 longitudes of one row: most row blocks hold
// no band cell, and the launch of their waves is a measurable share of the kernel (14 us for the
// empty grid at CPW = 1 on the 2560 x 1920 grid), so fewer, wider workgroups are launched.
// ------------------------------------------------------------------------------------
#ifndef SB_WIND_MINW
#define SB_WIND_MINW 8               // waves per SIMD asked of the compiler for k_wind: 30 registers and 61 scalar
#endif                               // registers per wave, so that eight 256-thread workgroups fit a CU
#if SB_WIND_MINW > 0
#define SB_WIND_BOUNDS __launch_bounds__(ROW_NT, SB_WIND_MINW)
#else
#define SB_WIND_BOUNDS __launch_bounds__(ROW_NT)
#endif
template <typename T, int UN, bool NTL, int CPW = 1>
__global__ SB_WIND_BOUNDS void k_wind(DiagJob<T> job, int ystride, int early) {
    __shared__ unsigned short s_x[ROW_NT * CPW];
    __shared__ int s_wcnt[ROW_NT / SB_WAVE * CPW];
    const Geo g = job.g;
    // clear the other tile-flag buffer for the next call (this call's is read by k_thc)
    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    for (int i = bid * ROW_NT + threadIdx.x; i < job.next_flags_n; i += gridDim.x * gridDim.y * ROW_NT)
        job.next_flags[i] = 0;
    // row of this workgroup: with ystride S, consecutive blockIdx.y are rows/S apart (experiment)
    int y = blockIdx.y;
    if (ystride > 1) {
        const int per = (g.rows + ystride - 1) / ystride;
        y = ((int)blockIdx.y % ystride) * per + (int)blockIdx.y / ystride;
        if (y >= g.rows) return;
    }
    if (early == 1) {
        // no band cell among the longitudes: leave before any LDS traffic or barrier.  Every wave
        // looks at the same words of the band plane, so the four waves agree.
        const int X0 = blockIdx.x * ROW_NT * CPW + g.h, w0 = X0 >> 6;
        const int lane = threadIdx.x & 63;
        const int nwd = min(g.nw - w0, ROW_NT * CPW / 64 + ((X0 & 63) ? 1 : 0));
        const uint64_t w = job.bandbits[(size_t)(y + g.h) * g.nw + w0 + (lane < nwd ? lane : 0)];
        if (__ballot(lane < nwd && w != 0) == 0) return;
    }
    const int total = block_band_cells<CPW>(job.bandbits, g, y, s_x, s_wcnt);
    if (early == 2) return;                                      // diagnostic: cost of the empty grid
    const size_t pl = (size_t)g.nx * g.ny;
    const int nz = job.nz;
    // one band cell per thread when the workgroup covers 256 longitudes: no loop.  With a loop the compiler
    // keeps the constants of the fp64 atan2 / update below in registers across it: 93 registers per lane
    // and 5 workgroups per CU instead of 8, which cost the kernel 5-10 us
    for (int i = threadIdx.x; i < total; i += (CPW == 1 ? (1 << 30) : ROW_NT)) {
        const int x = blockIdx.x * ROW_NT * CPW + s_x[i];
        const size_t o = (size_t)y * g.nx + x;
        // the final update's inputs: issued ahead of the column walk
        T n_thc = T(0), ws_old = T(0), wd_old = T(0);
        if (job.wind_final) { n_thc = job.thc[o]; ws_old = job.ws[o]; wd_old = job.wd[o]; }
        int lev = 0;
        if (job.flavour == SB_FLAVOUR_GENERIC) {
            // whole batches of UN levels in flight; the last batch is padded by re-reading level
            // nz-1 (never a new minimum: the comparison is strict), so no serial tail of single
            // loads follows the batches
            const T *pc = job.p + o;
            T best = T(0);
            for (int k0 = 0; k0 < nz; k0 += UN) {
                T d[UN];
#pragma unroll
                for (int q = 0; q < UN; ++q) {
                    const int k = k0 + q < nz ? k0 + q : nz - 1;
                    d[q] = sb_ld<T, NTL>(pc + (size_t)k * pl);
                }
#pragma unroll
                for (int q = 0; q < UN; ++q) {
                    const int k = k0 + q;
                    const T a = fabs(d[q] - job.target_plev);
                    if (k == 0) best = a;                        // the search starts at level 1 (ref :223)
                    else if (k < nz && a < best) { best = a; lev = k; }
                }
            }
        } else {
            T best = fabs(job.p[0] - job.target_plev);
            for (int k = 1; k < nz; ++k) {
                const T a = fabs(job.p[k] - job.target_plev);
                if (a < best) { best = a; lev = k; }
            }
        }
        const T uu = job.u[(size_t)lev * pl + o];
        const T vv = job.v[(size_t)lev * pl + o];
        const T n_ws = sqrt(uu * uu + vv * vv);                  // ref :225
        const T n_wd = atan2(-uu, -vv) * T(57.2957);             // ref :227, rad2deg (sic) :128
        if (job.wind_final) {
            // k_thc2 ran first and left this call's contrast in thc: thresholds, scaling and state
            // update happen here, under the HBM latency of the column walk   ref :235-266
            sb_trigger_update<T>(job, o, n_thc, SbCellState<T>{n_ws, n_wd, ws_old, wd_old});
        } else {
            job.nws[o] = n_ws;
            job.nwd[o] = n_wd;
        }
    }
}

// ------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------
template <typename T>
hipError_t sb_launch_stats(const T *ary, int nx, int ny, int ld, size_t off0, Moments *partials, T *stats,
                           Moments *moments_out, hipStream_t st) {
    // one workgroup per CU, fewer when the field is small (8 elements per thread per trip)
    const size_t n = (size_t)nx * ny;
    int nblk = (int)((n + (size_t)STATS_NT * 8 - 1) / ((size_t)STATS_NT * 8));
    if (nblk < 1) nblk = 1;
    if (nblk > 256) nblk = 256;
    hipLaunchKernelGGL(k_stats<T>, dim3(nblk), dim3(STATS_NT), 0, st, ary, nx, ny, ld, off0, partials);
    hipLaunchKernelGGL(k_moments_final<T>, dim3(1), dim3(STATS_NT), 0, st, partials, nblk, stats, moments_out);
    return hipGetLastError();
}

template <typename T>
hipError_t sb_launch_sigmoid_apply(const T *ary, T *sm, size_t n, const T *stats, hipStream_t st) {
    int nblk = (int)((n + 255) / 256);
    if (nblk > 4096) nblk = 4096;
    hipLaunchKernelGGL(k_sigmoid_apply<T>, dim3(nblk), dim3(256), 0, st, ary, sm, n, stats);
    return hipGetLastError();
}

// k_scan is instantiated per flavour and with/without the statistics, so that neither is a branch in its
// (issue-bound) segment loop
template <typename T>
static void launch_scan(const DiagJob<T> &job, int nblk, Moments *partials, bool stats, hipStream_t st) {
    const bool wr = job.flavour == SB_FLAVOUR_WRAPPER;
    if (wr && stats) hipLaunchKernelGGL((k_scan<T, 2, true, true>), dim3(nblk), dim3(STATS_NT), 0, st, job, partials);
    else if (wr) hipLaunchKernelGGL((k_scan<T, 2, true, false>), dim3(nblk), dim3(STATS_NT), 0, st, job, partials);
    else if (stats) hipLaunchKernelGGL((k_scan<T, 2, false, true>), dim3(nblk), dim3(STATS_NT), 0, st, job, partials);
    else hipLaunchKernelGGL((k_scan<T, 2, false, false>), dim3(nblk), dim3(STATS_NT), 0, st, job, partials);
}

template <typename T>
hipError_t sb_launch_diag(const DiagJob<T> &job, int H, const SbLaunchCtx &lc) {
    const Geo &g = job.g;
    hipStream_t st = lc.stream;
    hipEvent_t *ev = lc.prof;
    static const int un = getenv("SB_WIND_UN") ? atoi(getenv("SB_WIND_UN")) : 8;   // tuning knob (diagnostic)
    const bool gathered = lc.gathered != nullptr;
    const bool ph1 = (lc.phases & 1) != 0, ph2 = (lc.phases & 2) != 0;
    // k_thc2 merges k_scan's moments itself; the f2py flavour needs the scalars earlier, for k_t0
    const bool merge_in_thc2 = job.t0_fly && !gathered;
    hipError_t e = hipSuccess;
    const unsigned nseg = (unsigned)g.nyh * (unsigned)g.nw;
    int nblk = (int)((nseg + 79) / 80);                          // 16 waves x 5 segments per trip
    if (nblk < 1) nblk = 1;
    static const int scan_wgs = getenv("SB_SCAN_WGS") ? atoi(getenv("SB_SCAN_WGS")) : 1;   // tuning knob (diagnostic)
    if (nblk > scan_wgs * lc.ncu) nblk = scan_wgs * lc.ncu;      // one 1024-thread workgroup per CU, two trips of loads in flight
    // Single-domain calls on the k_thc2 path run the contrast first and let k_wind apply the
    // thresholds and the state update (job.wind_final); a band step must run k_scan + k_wind
    // before its ghost rows arrive, so there k_thc2 applies them.
    if (job.wind_final && ph2) {
        if (ev) { (void)hipEventRecord(ev[0], st); }
        launch_scan<T>(job, nblk, lc.partials, true, st);
        if (!merge_in_thc2)
            hipLaunchKernelGGL(k_moments_final<T>, dim3(1), dim3(STATS_NT), 0, st, lc.partials, nblk, (T *)lc.stats,
                               (Moments *)nullptr);
        if (ev) { (void)hipEventRecord(ev[1], st); (void)hipEventRecord(ev[4], st); }
        if (!job.t0_fly) hipLaunchKernelGGL(k_t0<T>, dim3((g.nxh + 255) / 256, g.nyh), dim3(256), 0, st, job);
        if (ev) { (void)hipEventRecord(ev[5], st); }
        if ((e = sb_launch_thc2<T>(job, H, lc.ncu, merge_in_thc2 ? lc.partials : nullptr, merge_in_thc2 ? nblk : 0,
                                   (T *)lc.stats, st)) != hipSuccess) return e;
        if (ev) { (void)hipEventRecord(ev[6], st); (void)hipEventRecord(ev[7], st); (void)hipEventRecord(ev[2], st); }
        static const int ystr = getenv("SB_WIND_YSTRIDE") ? atoi(getenv("SB_WIND_YSTRIDE")) : 1;   // tuning knob (diagnostic)
        static const int early = getenv("SB_WIND_EARLY") ? atoi(getenv("SB_WIND_EARLY")) : 0;       // tuning knob (diagnostic)
        const int gy = ystr > 1 ? ystr * ((g.rows + ystr - 1) / ystr) : g.rows;
        const dim3 wg((g.nx + ROW_NT - 1) / ROW_NT, gy), wb(ROW_NT);
        if (un <= 4) hipLaunchKernelGGL((k_wind<T, 4, true>), wg, wb, 0, st, job, ystr, early);
        else if (un <= 8) hipLaunchKernelGGL((k_wind<T, 8, true>), wg, wb, 0, st, job, ystr, early);
        else hipLaunchKernelGGL((k_wind<T, 14, true>), wg, wb, 0, st, job, ystr, early);
        if (ev) { (void)hipEventRecord(ev[3], st); }
        return hipGetLastError();
    }
    // ---- phase 1: needs neither theta's ghost cells nor the statistics -----------------------
    if (ph1) {
        // k_scan (+ merge of the statistics when they are this domain's own and k_thc2 does not do it)
        if (ev) (void)hipEventRecord(ev[0], st);
        // a band step takes this band's own sigma moments from the same pass (lc.moments_out), publishes
        // them for the all-gather and signals the communication stream
        launch_scan<T>(job, nblk, lc.partials, !gathered || lc.moments_out != nullptr, st);
        if (gathered && lc.moments_out) {
            hipLaunchKernelGGL(k_moments_final<T>, dim3(1), dim3(STATS_NT), 0, st, lc.partials, nblk, (T *)lc.stats,
                               lc.moments_out);
            if (lc.moments_event && (e = hipEventRecord(lc.moments_event, st)) != hipSuccess) return e;
        }
        if (!gathered && !merge_in_thc2)
            hipLaunchKernelGGL(k_moments_final<T>, dim3(1), dim3(STATS_NT), 0, st, lc.partials, nblk, (T *)lc.stats,
                               (Moments *)nullptr);
        if (ev) (void)hipEventRecord(ev[1], st);
        // k_wind
        if (ev) (void)hipEventRecord(ev[2], st);
        const dim3 wg((g.nx + ROW_NT - 1) / ROW_NT, g.rows), wb(ROW_NT);
        static const bool plain = getenv("SB_WIND_PLAIN") != nullptr;   // tuning knob (diagnostic)
        if (plain) hipLaunchKernelGGL((k_wind<T, 8, false>), wg, wb, 0, st, job, 1, 0);
        else if (un <= 4) hipLaunchKernelGGL((k_wind<T, 4, true>), wg, wb, 0, st, job, 1, 0);
        else if (un <= 8) hipLaunchKernelGGL((k_wind<T, 8, true>), wg, wb, 0, st, job, 1, 0);
        else hipLaunchKernelGGL((k_wind<T, 14, true>), wg, wb, 0, st, job, 1, 0);
        if (ev) (void)hipEventRecord(ev[3], st);
    }
    // ---- phase 2: statistics of all bands, theta with its ghost cells -------------------------
    if (ph2) {
        // the bands' moments: merged by k_thc2 itself (host-model flavour, up to 64 bands: the same tree as
        // k_merge_moments) or by a launch of their own (the f2py flavour's k_t0 needs the scalars first)
        const bool gath_in_thc2 = gathered && job.t0_fly && lc.ngathered <= SB_WAVE;
        if (gathered && !gath_in_thc2)
            hipLaunchKernelGGL(k_merge_moments<T>, dim3(1), dim3(SB_WAVE), 0, st, lc.gathered, lc.ngathered, (T *)lc.stats);
        if (ev) (void)hipEventRecord(ev[4], st);
        // k_t0 (f2py flavour: the t0 plane is an output)
        if (!job.t0_fly) hipLaunchKernelGGL(k_t0<T>, dim3((g.nxh + 255) / 256, g.nyh), dim3(256), 0, st, job);
        if (ev) (void)hipEventRecord(ev[5], st);
        // contrast, thresholds, state update
        const Moments *mp = merge_in_thc2 ? lc.partials : gath_in_thc2 ? lc.gathered : nullptr;
        const int mn = merge_in_thc2 ? nblk : gath_in_thc2 ? lc.ngathered : 0;
        e = sb_launch_thc2<T>(job, H, lc.ncu, mp, mn, (T *)lc.stats, st);
        if (e != hipSuccess) return e;
        if (ev) (void)hipEventRecord(ev[6], st);
        if (ev) (void)hipEventRecord(ev[7], st);
    }
    return hipGetLastError();
}

template hipError_t sb_launch_stats<float>(const float *, int, int, int, size_t, Moments *, float *, Moments *,
                                           hipStream_t);
template hipError_t sb_launch_stats<double>(const double *, int, int, int, size_t, Moments *, double *, Moments *,
                                            hipStream_t);
template hipError_t sb_launch_sigmoid_apply<float>(const float *, float *, size_t, const float *, hipStream_t);
template hipError_t sb_launch_sigmoid_apply<double>(const double *, double *, size_t, const double *, hipStream_t);
template hipError_t sb_launch_diag<float>(const DiagJob<float> &, int, const SbLaunchCtx &);
template hipError_t sb_launch_diag<double>(const DiagJob<double> &, int, const SbLaunchCtx &);
