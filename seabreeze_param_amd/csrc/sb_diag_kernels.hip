// sb_diag_kernels.hip -- the seabreeze_diag hot path as CDNA4 (gfx950) kernels.
//
// One call of seabreeze_diag / diag (ref: generic/sea_breeze_diag.f90:55-271,
// python_wrapper/seabreezediag/seabreeze_diag_python.f90:49-285) is three launches on one
// stream for a host-model call on a single domain, four or five otherwise:
//
//   k_scan   one pass over sigma and mask: per-workgroup shifted sums of sigma, the band and
//            land-side bit planes (compared word by word with those of the call before: the strip
//            kernel's stored plan stands while they do), block flags, and the fill value for every
//            cell outside the coastal band                                    [HBM stream]
//   k_prep   (f2py flavour, search radii beyond 16, the first step of a band run, sb_set_fold(ctx, 0);
//            otherwise the strip kernel does this work itself) 2 + SB_SEG_PARTS workgroups: the sums
//            merged into the sigmoid scalars, the tile flags compacted into the list of active tiles
//            (for k_thc3), the band plane into the lists of 64-cell segments that hold band cells
//            (for k_wind)                                                      [tiny]
//   k_t0     f2py flavour only: the t0 plane is an output there               [HBM stream]
//   k_strip  (sb_strip_kernel.hip; radii up to 16) t0 and its summed-area tables marched through a
//            ring in LDS, strip by strip -> thc; k_thc3 (sb_thc_kernel.hip) per tile for radii
//            up to 32                                                          [VALU + LDS]
//   k_wind   per listed segment: level nearest target_plev in the p column of every band
//            cell, wind speed / direction, thresholds, scaling, state update   [HBM gather]
//
// A band step of a multi-GPU run launches k_wind before its ghost rows arrive (instance <0>: the winds go
// to scratch planes); the contrast kernel applies thresholds and update behind its march.
// Memory-bound integer/fp64 work: no MFMA anywhere.
#include "sb_device.hpp"
#include "sb_launch.hpp"
#include "sb_scan_body.hpp"

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

// Workgroup partial -> global partials.  For k_scan the cross-workgroup merge is the next kernel's business (the strip
// kernel's prologue, or k_prep): an in-kernel ticket with an agent-scope release per workgroup drains the XCD's dirty
// L2 every time, and k_scan writes tens of MB (the fill) -- measured 20 us for 512 workgroups, against ~2 us for a
// kernel boundary.  k_stats, which writes nothing else, hands its partials over with write-through stores instead
// (no release, nothing to drain) and merges them itself -- see there.
__device__ __forceinline__ void store_partial(Moments acc, Moments *wpart, Moments *__restrict__ partials) {
    acc = block_merge(acc, wpart);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// ------------------------------------------------------------------------------------
// k_stats: statistics of a plain 2-D field (stand-alone sigmoid; band-local moments for
// the multi-GPU gather).  One workgroup per CU, 8 independent loads per thread in flight;
// sums about a common shift c (first element) -> (n, mean, M2) once per thread, merged
// pairwise (Chan) from there on.  Replaces the reference's two sequential passes
// (ref: generic/sea_breeze_diag.f90:466-477).
// ------------------------------------------------------------------------------------
// The workgroup that arrives last merges the partials -- in index order, a fixed tree: the result does not depend on who is
// last -- and publishes the moments (band-local, for the multi-GPU gather) or the sigmoid scalars: one launch, where a
// merge kernel of its own used to follow (a band step's communication stream is a chain of such small launches).
// Hand-over across workgroups (the per-XCD L2s are not coherent with each other; k_scan may be filling them with dirty
// lines on another stream meanwhile, so no L2 write-back here): every partial is stored write-through (agent-scope
// atomic stores) and drained before its workgroup draws a ticket; the last arriver reads them past its own caches
// (agent-scope atomic loads).
template <typename T>
__global__ __launch_bounds__(STATS_NT) void k_stats(const T *__restrict__ ary, int nx, int ny, int ld,
                                                    size_t off0, Moments *__restrict__ partials, int *__restrict__ ticket,
                                                    T *__restrict__ stats, Moments *__restrict__ moments_out) {
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
    __shared__ int s_last;
    const Moments mine = block_merge(moments_from_shifted(c, s1, s2, mn, mx, cnt), wpart);
    if (threadIdx.x == 0) {
        double *dst = (double *)&partials[blockIdx.x];
        const double v[5] = {mine.n, mine.mean, mine.m2, mine.mn, mine.mx};
        static_assert(sizeof(Moments) == 5 * sizeof(double), "Moments is five doubles");
        for (int i = 0; i < 5; ++i) __hip_atomic_store(dst + i, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == (int)gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    Moments m = moments_empty();
    for (int b = threadIdx.x; b < (int)gridDim.x; b += STATS_NT) {
        const double *src = (const double *)&partials[b];
        Moments o;
        o.n = __hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        o.mean = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        o.m2 = __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        o.mn = __hip_atomic_load(src + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        o.mx = __hip_atomic_load(src + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        m = moments_merge(m, o);
    }
    __syncthreads();                                     // (wpart is used again)
    m = block_merge(m, wpart);
    if (threadIdx.x == 0) {
        if (moments_out) *moments_out = m;
        else sigmoid_scalars<T>(m, stats);
        __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // (zeroed at creation; every launch leaves it zero)
    }
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
        sm[i] = sb_logistic_of_neg<T>(-sd * (ary[i] - r));
}

// ------------------------------------------------------------------------------------
// k_scan: one persistent 1024-thread workgroup per CU streams sigma and mask.  A wave
// owns aligned 64-cell longitude segments (four per trip, eight loads in flight), so its
// ballots are exactly the words of the bit planes.
//   * sigma  -> shifted sums about its first interior value -> one (count, s1, s2, min, max) partial per workgroup
//   * mask   -> land-side bit  mask >= 0              ref: generic/sea_breeze_diag.f90:182,200
//            -> band bit  !(|mask| > maxdist)         ref :174
//            -> flag of the k_thc tile(s) the segment's band cells fall in
//            -> fill value outside the band           ref :176 / seabreeze_diag_python.f90:173,279-280
// ------------------------------------------------------------------------------------
template <typename T, int SPT, bool WR, bool ST>      // WR: f2py flavour; ST: accumulate sigma's moments
__global__ __launch_bounds__(STATS_NT) void k_scan(DiagJob<T> job, Moments *__restrict__ partials) {
    constexpr bool do_stats = ST;
    Moments t;
    const bool plane_changed = sb_scan_pass<T, SPT, WR, ST>(job, t);
    const Geo g = job.g;
    const double c = do_stats ? (double)job.sigma[(size_t)g.h * g.nxh + g.h] : 0.0;
    if (plane_changed && job.plan_gen) atomicMax(job.plan_gen, job.call_id);
    if (!do_stats) return;
    // the workgroup's shifted sums (sb_device.hpp: merged by addition downstream, converted once at the end)
    __shared__ Moments wpart[STATS_NT / SB_WAVE];
    t = block_total_shifted<STATS_NT / SB_WAVE>(t, wpart);
    if (!job.moments_out) {                              // the next kernel adds them up (strip kernel's prologue, k_prep)
        if (threadIdx.x == 0) partials[blockIdx.x] = t;
        return;
    }
    // A band step: the moments of this band go to the all-gather as soon as they exist -- the last workgroup to arrive
    // adds the partials up (k_prep's order: same bits) and publishes them.  Hand-over as in k_stats: write-through
    // stores, drained, then the ticket; read past the caches.  (No agent-scope release: it would write back the L2's
    // dirty lines, and this kernel has just written the fill values of most of its cells.)
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        double *dst = (double *)&partials[blockIdx.x];
        const double v[5] = {t.n, t.mean, t.m2, t.mn, t.mx};
        for (int i = 0; i < 5; ++i) __hip_atomic_store(dst + i, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int tk = __hip_atomic_fetch_add(job.stats_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = tk == (int)gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    Moments v = moments_empty();
    for (int b = threadIdx.x; b < (int)gridDim.x; b += STATS_NT) {
        const double *src = (const double *)&partials[b];
        v.n += __hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.mean += __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.m2 += __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.mn = fmin(v.mn, __hip_atomic_load(src + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        v.mx = fmax(v.mx, __hip_atomic_load(src + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    __syncthreads();                                     // (wpart is used again)
    const Moments m = moments_of_shifted(c, block_total_shifted<STATS_NT / SB_WAVE>(v, wpart));
    if (threadIdx.x == 0) {
        *job.moments_out = m;
        __hip_atomic_store(job.stats_ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
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

// theta <- t0 in place over the whole ghost-celled frame: what the UM copy does before its point loop
// (ref: UM/vn10.7/sea_breeze_diag.F90:210-211); optional, after the diagnostic has read theta.
template <typename T>
__global__ __launch_bounds__(256) void k_theta_to_t0(T *__restrict__ theta, const T *__restrict__ z,
                                                     const T *__restrict__ sigma, size_t n, const T *__restrict__ stats) {
    const T sd = stats[0], r = stats[1];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        theta[i] = sb_t0<T>(theta[i], z[i], sigma[i], sd, r);
}

template <typename T>
hipError_t sb_launch_theta_to_t0(T *theta, const T *z, const T *sigma, size_t n, const T *stats, hipStream_t st) {
    int nblk = (int)((n + 255) / 256);
    if (nblk > 4096) nblk = 4096;
    hipLaunchKernelGGL(k_theta_to_t0<T>, dim3(nblk), dim3(256), 0, st, theta, z, sigma, n, stats);
    return hipGetLastError();
}
template hipError_t sb_launch_theta_to_t0<float>(float *, const float *, const float *, size_t, const float *, hipStream_t);
template hipError_t sb_launch_theta_to_t0<double>(double *, const double *, const double *, size_t, const double *, hipStream_t);

// ------------------------------------------------------------------------------------
// k_prep: the small jobs between k_scan and the kernels that consume its flags, one role per
// workgroup (2 + SB_SEG_PARTS workgroups), so that neither the 256 persistent workgroups of k_thc3 nor
// the waves of k_wind repeat them:
//   block 0               k_scan's per-workgroup shifted sums added up in index order (a fixed tree) into the
//                         sigmoid scalars
//   block 1               tile flags -> row-major list of active tiles: tile_list[0] = count, then the tiles, then
//                         tile_pad entries of -1
//   blocks 2 .. 2+PARTS-1 band plane -> the 64-cell segments that hold band cells, each part a
//                         contiguous range of segments with its own sub-list (ascending order)
// ------------------------------------------------------------------------------------
#define PREP_NT 1024

// exclusive prefix of one int per thread over the workgroup; total in `total`.  Two barriers.
__device__ __forceinline__ int block_excl_scan(int v, int *s_w, int &total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int incl = sb_wave_scan_add(v);
    __syncthreads();
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    const int wt = lane < PREP_NT / SB_WAVE ? s_w[lane] : 0;
    const int wincl = sb_wave_scan_add(wt);
    total = __shfl(wincl, PREP_NT / SB_WAVE - 1);
    return incl - v + __shfl(wincl, wv) - __shfl(wt, wv);
}

// Ordered compaction by one PREP_NT-thread workgroup: emit(rank, i) for every i in [0, n) with pred(i), rank = number
// of earlier such i.  Items are taken PREP_NT at a time (coalesced, independent loads), a ballot per chunk and wave
// gives the per-(chunk, wave) counts, one workgroup scan their prefix.  n <= 64 * PREP_NT; returns the total.
#define PREP_MAXCH 64
template <typename Pred, typename Emit>
__device__ __forceinline__ int block_compact(int n, int *s_cnt, int *s_w, Pred pred, Emit emit) {
    constexpr int NW = PREP_NT / SB_WAVE;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nch = (n + PREP_NT - 1) / PREP_NT;
    uint64_t mine = 0;
    for (int c0 = 0; c0 < nch; c0 += 4) {                        // four independent (clamped) loads in flight
        bool on[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = (c0 + j) * PREP_NT + tid;
            on[j] = pred(i < n ? i : n - 1) && i < n;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j;
            const uint64_t b = __ballot(on[j]);
            mine |= on[j] ? 1ull << c : 0ull;
            if (lane == 0 && c < nch) s_cnt[c * NW + wv] = __popcll(b);
        }
    }
    __syncthreads();
    const int v = tid < nch * NW ? s_cnt[tid] : 0;
    int total;
    const int excl = block_excl_scan(v, s_w, total);     // its barriers stand between the reads above and the writes below
    if (tid < nch * NW) s_cnt[tid] = excl;
    __syncthreads();
    const uint64_t lt = (1ull << lane) - 1ull;
    for (int c = 0; c < nch; ++c) {
        const bool on = (mine >> c) & 1ull;
        const uint64_t b = __ballot(on);
        if (on) emit(s_cnt[c * NW + wv] + __popcll(b & lt), c * PREP_NT + tid);
    }
    return total;
}

template <typename T>
__global__ __launch_bounds__(PREP_NT) void k_prep(DiagJob<T> job, const Moments *__restrict__ partials, int nparts,
                                                  T *__restrict__ stats, Moments *__restrict__ moments_out) {
    __shared__ Moments wpart[PREP_NT / SB_WAVE];
    __shared__ int s_w[PREP_NT / SB_WAVE];
    __shared__ int s_cnt[PREP_MAXCH * (PREP_NT / SB_WAVE)];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        if (nparts <= 0) return;
        // k_scan's shifted sums (about sigma's first interior value) added up in index order, converted once
        const double c = (double)job.sigma[(size_t)job.g.h * job.g.nxh + job.g.h];
        Moments v = moments_empty();
        for (int b = tid; b < nparts; b += PREP_NT) {
            const Moments o = partials[b];
            v.n += o.n; v.mean += o.mean; v.m2 += o.m2;
            v.mn = fmin(v.mn, o.mn); v.mx = fmax(v.mx, o.mx);
        }
        const Moments m = moments_of_shifted(c, block_total_shifted<PREP_NT / SB_WAVE>(v, wpart));
        if (tid == 0) {
            if (moments_out) *moments_out = m;
            else sigmoid_scalars<T>(m, stats);
        }
        return;
    }
    if (blockIdx.x == 1) {
        if (job.strip) return;                       // the strip kernel ranks the flags itself
        const int ntiles = job.thc_ntx * job.thc_nty;
        int total;
        if (ntiles <= PREP_MAXCH * PREP_NT) {
            total = block_compact(ntiles, s_cnt, s_w, [&](int t) { return job.tile_nnmax[t] != 0; },
                                  [&](int rank, int t) { job.tile_list[1 + rank] = t; });
        } else {                                     // more tiles than the ballot form holds: a serial run per thread
            const int per = (ntiles + PREP_NT - 1) / PREP_NT;
            const int t0 = tid * per, t1 = min(t0 + per, ntiles);
            int cnt = 0;
            for (int t = t0; t < t1; ++t) cnt += job.tile_nnmax[t] != 0 ? 1 : 0;
            int at = block_excl_scan(cnt, s_w, total);
            for (int t = t0; t < t1; ++t)
                if (job.tile_nnmax[t] != 0) job.tile_list[1 + at++] = t;
        }
        if (tid == 0) job.tile_list[0] = total;
        // the list ends in -1 entries, two per k_thc3 workgroup: a workgroup reads its first two positions before
        // it knows the count, and finds its end without it
        for (int i = tid; i < job.tile_pad; i += PREP_NT) job.tile_list[1 + total + i] = -1;
        return;
    }
    const int part = (int)blockIdx.x - 2;
    const unsigned nseg = (unsigned)job.g.nyh * (unsigned)job.g.nw;
    const unsigned cap = (unsigned)job.seg_cap;
    const unsigned s0 = min((unsigned)part * cap, nseg), s1 = min(s0 + cap, nseg);
    SbSegEntry *list = job.seg_list + (size_t)part * cap;
    int total;
    if (cap <= (unsigned)(PREP_MAXCH * PREP_NT)) {
        total = block_compact((int)(s1 - s0), s_cnt, s_w, [&](int i) { return job.bandbits[s0 + (unsigned)i] != 0; },
                              [&](int rank, int i) {
                                  SbSegEntry e;
                                  e.word = job.bandbits[s0 + (unsigned)i]; e.seg = s0 + (unsigned)i; e.pad = 0;
                                  list[rank] = e;
                              });
    } else {
        const unsigned per = (cap + PREP_NT - 1) / PREP_NT;
        const unsigned a0 = min(s0 + (unsigned)tid * per, s1), a1 = min(a0 + per, s1);
        int cnt = 0;
        for (unsigned sg = a0; sg < a1; ++sg) cnt += job.bandbits[sg] != 0 ? 1 : 0;
        int at = block_excl_scan(cnt, s_w, total);
        for (unsigned sg = a0; sg < a1; ++sg) {
            const uint64_t w = job.bandbits[sg];
            if (w) { SbSegEntry e; e.word = w; e.seg = sg; e.pad = 0; list[at++] = e; }
        }
    }
    if (tid == 0) job.seg_count[part] = total;
}

// ------------------------------------------------------------------------------------
// k_wind: level nearest the target pressure + wind speed/direction for every band cell, and -- on a
// single domain, where k_thc3 has already left this call's contrast in thc -- thresholds, scaling and
// state update.   ref: generic/sea_breeze_diag.f90:223-227,235-266; seabreeze_diag_python.f90:228-233
// (1-D p: one level for all cells).
//
// Persistent waves work through k_prep's list of segments that hold band cells; a wave owns one aligned
// 64-cell segment (four 128-byte lines of every level plane) at a time and its band lanes walk their p
// column (nz planes, stride nx*ny) in whole batches of UN non-temporal loads.  No workgroup is launched
// for the half of the 256-longitude row blocks that hold no band cell, no lane compaction, no barrier;
// every line of p that holds a band cell is requested by exactly one wave.
// ------------------------------------------------------------------------------------
#define WIND_NT 256

template <typename T, bool NTL>
__device__ __forceinline__ T sb_ld(const T *p) {
    if constexpr (NTL) return __builtin_nontemporal_load(p);
    else return *p;
}

// MODE 0: the winds go to scratch planes (a band step, ahead of the contrast kernel, which applies the update);
// 1: the update is applied here (single domain: the contrast kernel ran first)
template <typename T> struct WindCfg { static constexpr int UN = SB_WIND_UN, WGS = SB_WIND_WGS_PER_CU; };
template <> struct WindCfg<float> { static constexpr int UN = SB_WIND_UN_F32, WGS = SB_WIND_WGS_PER_CU_F32; };

// diagnostic build (make stamps EXTRA=-DSB_STAMPS_WIND, tools/stamp_wind.py): the wall clock of every wave of k_wind at
// the marks below (the strip kernels' marks are off in that build: one buffer)
#ifdef SB_STAMPS_WIND
#define SB_WT(i) do { if (stl_ && job.stamps) job.stamps[(size_t)gw_ * SB_NSTAMP + (i)] = wall_clock64(); } while (0)
#else
#define SB_WT(i) do { } while (0)
#endif

template <typename T, int UN, int MODE>
__global__ __launch_bounds__(WIND_NT, WindCfg<T>::WGS) void k_wind(DiagJob<T> job) {
    constexpr bool FINAL = MODE != 0;
    const Geo g = job.g;
    const int lane = threadIdx.x & 63;
#ifdef SB_STAMPS_WIND
    const int gw_ = blockIdx.x * (WIND_NT / SB_WAVE) + (threadIdx.x >> 6);
    int nsegs_ = 0;
    bool stl_ = lane == 0;
    if (gw_ < 4096 && lane < SB_NSTAMP && job.stamps) job.stamps[(size_t)gw_ * SB_NSTAMP + lane] = 0;
#endif
    SB_WT(0);                                            // start
    // clear the other tile-flag buffer for the next call (this call's was read by k_prep)
    for (int i = blockIdx.x * WIND_NT + threadIdx.x; i < job.next_flags_n; i += gridDim.x * WIND_NT) job.next_flags[i] = 0;
    // the sub-lists' sizes -> position of an entry: lane k holds (count, inclusive prefix) of sub-list k
    const int cnt = lane < SB_SEG_PARTS ? job.seg_count[lane] : 0;
    const int incl = sb_wave_scan_add(cnt);
    const int total = __builtin_amdgcn_readlane(incl, SB_SEG_PARTS - 1);
    const int nwaves = gridDim.x * (WIND_NT / SB_WAVE);
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * (WIND_NT / SB_WAVE) + (threadIdx.x >> 6));
    auto entry = [&](int e) {
        const uint64_t hit = __ballot(lane < SB_SEG_PARTS && e < incl);
        const int part = hit ? __ffsll((unsigned long long)hit) - 1 : 0;
        const int base = __shfl(incl, part) - __shfl(cnt, part);
        return job.seg_list[(size_t)part * job.seg_cap + (e < total ? e - base : 0)];
    };
    const size_t pl = (size_t)g.nx * g.ny;
    const int nz = job.nz;
    // 1-D p (f2py flavour): one level for the whole grid
    int lev1 = 0;
    if (job.flavour != SB_FLAVOUR_GENERIC) {
        T best = fabs(job.p[0] - job.target_plev);
        for (int k = 1; k < nz; ++k) {
            const T a = fabs(job.p[k] - job.target_plev);
            if (a < best) { best = a; lev1 = k; }
        }
    }
    // one aligned 64-cell segment: its band lanes walk their column
    auto segment = [&](const SbSegEntry cur) __attribute__((always_inline)) {
#ifdef SB_STAMPS_WIND
        const int sbase_ = 3 + 4 * (nsegs_ < 6 ? nsegs_ : 6);
        ++nsegs_;
        stl_ = lane == __ffsll((unsigned long long)cur.word) - 1;     // the segment's first band lane keeps the clock
        SB_WT(sbase_);                                   // segment begins (its entry has arrived)
#endif
        const unsigned Y = cur.seg / (unsigned)g.nw, Xw = cur.seg - Y * (unsigned)g.nw;
        const int x = (int)(Xw * 64u) + lane - g.h, y = (int)Y - g.h;
        if (!((cur.word >> lane) & 1ull)) return;               // band bits are set for interior cells of processed rows only
        const size_t o = (size_t)y * g.nx + x;
        // the final update's inputs: issued ahead of the column walk
        T n_thc = T(0), ws_old = T(0), wd_old = T(0);
        if constexpr (FINAL) { n_thc = job.thc[o]; ws_old = job.ws[o]; wd_old = job.wd[o]; }
        int lev = lev1;
        if (job.flavour == SB_FLAVOUR_GENERIC && job.level_rule == 0) {
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
                    d[q] = sb_ld<T, true>(pc + (size_t)k * pl);
                }
#pragma unroll
                for (int q = 0; q < UN; ++q) {
                    const int k = k0 + q;
                    const T a = fabs(d[q] - job.target_plev);
                    if (k == 0) best = a;                        // the search starts at level 1 (ref :223)
                    else if (k < nz && a < best) { best = a; lev = k; }
                }
            }
        } else if (job.flavour == SB_FLAVOUR_GENERIC) {
            // the UM copy's walk: upwards while |p - target| does not grow (ties move on), starting from a
            // difference of 1e6; it stops at the first increase   ref: UM/vn10.7/sea_breeze_diag.F90:265-274.
            // (Where even the first level is further than 1e6 from the target the UM copy leaves the level
            // undefined; level 1 is used here.)
            const T *pc = job.p + o;
            T diff = T(1000000.);
            bool done = false;
            for (int k0 = 0; k0 < nz && !done; k0 += UN) {
                T d[UN];
#pragma unroll
                for (int q = 0; q < UN; ++q) {
                    const int k = k0 + q < nz ? k0 + q : nz - 1;
                    d[q] = sb_ld<T, true>(pc + (size_t)k * pl);
                }
#pragma unroll
                for (int q = 0; q < UN; ++q) {
                    const int k = k0 + q;
                    const T a = fabs(d[q] - job.target_plev);
                    if (!done && k < nz) {
                        if (a <= diff) { lev = k; diff = a; }
                        else done = true;
                    }
                }
            }
        }
#ifdef SB_STAMPS_WIND
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SB_WT(sbase_ + 1);                               // column walked
#endif
        const T uu = job.u[(size_t)lev * pl + o];
        const T vv = job.v[(size_t)lev * pl + o];
#ifdef SB_STAMPS_WIND
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SB_WT(sbase_ + 2);                               // u, v there
#endif
        const T n_ws = sqrt(uu * uu + vv * vv);                  // ref :225
        const T n_wd = atan2(-uu, -vv) * T(57.2957);             // ref :227, rad2deg (sic) :128
        if constexpr (FINAL) {
            // k_thc3 ran first and left this call's contrast in thc: thresholds, scaling and state
            // update happen here, under the HBM latency of the column walk   ref :235-266
            sb_trigger_update<T, false>(job, o, n_thc, SbCellState<T>{n_ws, n_wd, ws_old, wd_old});
        } else {
            job.nws[o] = n_ws;
            job.nwd[o] = n_wd;
        }
#ifdef SB_STAMPS_WIND
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SB_WT(sbase_ + 3);                               // updated, stores drained
#endif
    };
    // The lists of the call before (a band step whose planning of the march stands: no k_prep in this call) hold as long
    // as k_scan found both planes unchanged; if it did not -- once per change of the coast -- the waves take the
    // segments of the plane itself in turn: unbalanced, correct.
    if (job.seg_trust && *job.plan_gen >= job.call_id) {
        const unsigned nseg = (unsigned)g.nyh * (unsigned)g.nw;
        for (unsigned sg = (unsigned)gw; sg < nseg; sg += (unsigned)nwaves) {
            SbSegEntry cur;
            cur.word = job.bandbits[sg]; cur.seg = sg; cur.pad = 0;
            if (cur.word) segment(cur);                           // wave-uniform
        }
        return;
    }
    SB_WT(1);                                            // sub-lists' sizes there
    if (gw >= total) return;
    SbSegEntry ent = entry(gw);
    for (int e = gw; e < total; e += nwaves) {
        const SbSegEntry cur = ent;
        if (e + nwaves < total) ent = entry(e + nwaves);          // the next entry travels under this one's walk
        segment(cur);
    }
#ifdef SB_STAMPS_WIND
    stl_ = lane == 0;
#endif
    SB_WT(2);                                            // end
}

// ------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------
template <typename T>
hipError_t sb_launch_stats(const T *ary, int nx, int ny, int ld, size_t off0, Moments *partials, T *stats,
                           Moments *moments_out, int *ticket, hipStream_t st) {
    // one workgroup per CU, fewer when the field is small (8 elements per thread per trip)
    const size_t n = (size_t)nx * ny;
    int nblk = (int)((n + (size_t)STATS_NT * 8 - 1) / ((size_t)STATS_NT * 8));
    if (nblk < 1) nblk = 1;
    if (nblk > 256) nblk = 256;
    hipLaunchKernelGGL(k_stats<T>, dim3(nblk), dim3(STATS_NT), 0, st, ary, nx, ny, ld, off0, partials, ticket, stats, moments_out);
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
// segments per trip of a wave of k_scan (two trips in flight): 2 in double precision; SB_SCAN_SPT_F32 in single precision,
// where a segment is half the bytes
#ifndef SB_SCAN_SPT_F32
#define SB_SCAN_SPT_F32 4              // (measured: 5120x3840 fp32 k_scan 73 -> 61 us against 2; the same at 2560x1920)
#endif
template <typename T> struct SCAN_SPT { static constexpr int value = 2; };
template <> struct SCAN_SPT<float> { static constexpr int value = SB_SCAN_SPT_F32; };

template <typename T>
static void launch_scan(const DiagJob<T> &job, int nblk, Moments *partials, bool stats, hipStream_t st) {
    const bool wr = job.flavour == SB_FLAVOUR_WRAPPER;
    if (wr && stats) hipLaunchKernelGGL((k_scan<T, SCAN_SPT<T>::value, true, true>), dim3(nblk), dim3(STATS_NT), 0, st, job, partials);
    else if (wr) hipLaunchKernelGGL((k_scan<T, SCAN_SPT<T>::value, true, false>), dim3(nblk), dim3(STATS_NT), 0, st, job, partials);
    else if (stats) hipLaunchKernelGGL((k_scan<T, SCAN_SPT<T>::value, false, true>), dim3(nblk), dim3(STATS_NT), 0, st, job, partials);
    else hipLaunchKernelGGL((k_scan<T, SCAN_SPT<T>::value, false, false>), dim3(nblk), dim3(STATS_NT), 0, st, job, partials);
}

template <typename T>
static void launch_wind(const DiagJob<T> &job, int ncu, hipStream_t st) {
    const dim3 wg(ncu * WindCfg<T>::WGS), wb(WIND_NT);
    if (job.wind_final) hipLaunchKernelGGL((k_wind<T, WindCfg<T>::UN, 1>), wg, wb, 0, st, job);
    else hipLaunchKernelGGL((k_wind<T, WindCfg<T>::UN, 0>), wg, wb, 0, st, job);
}

// event pair k of a profiled call brackets kernel k (SB_PROF_*); pairs of kernels a call does not launch stay unrecorded
#define SB_EV_BEGIN(k) do { if (ev) { (void)hipEventRecord(ev[2 * (k)], st); *lc.prof_mask |= 1u << (k); } } while (0)
#define SB_EV_END(k)   do { if (ev) (void)hipEventRecord(ev[2 * (k) + 1], st); } while (0)

// the contrast kernel for this job: marching strips (LDS halo 16; 32 in single precision) or LDS tiles (halos of 24 and 32 cells)
template <typename T>
static hipError_t launch_contrast(const DiagJob<T> &job, int H, int ncu, hipStream_t st) {
    if (job.strip == 2) return sb_launch_strip32<T>(job, ncu, st);
    return job.strip ? sb_launch_strip<T>(job, ncu, st) : sb_launch_thc<T>(job, H, ncu, st);
}

template <typename T>
hipError_t sb_launch_diag(const DiagJob<T> &job, int H, const SbLaunchCtx &lc) {
    const Geo &g = job.g;
    hipStream_t st = lc.stream;
    hipEvent_t *ev = lc.prof;
    const bool gathered = lc.gathered != nullptr;
    const bool ph1 = (lc.phases & 1) != 0, ph2 = (lc.phases & 2) != 0;
    const bool reuse = lc.reuse_stats;
    hipError_t e = hipSuccess;
    int nl = 0;                                                  // kernels enqueued
    const unsigned nseg = (unsigned)g.nyh * (unsigned)g.nw;
    int nblk = (int)((nseg + 31) / 32);                          // 16 waves x one trip of 2 segments: a small domain (a band of
                                                                 // a multi-GPU run) is spread over all CUs, its waves make few
                                                                 // dependent trips -- k_scan is latency, not bytes, there
    if (nblk < 1) nblk = 1;
    if (nblk > lc.ncu) nblk = lc.ncu;                            // one 1024-thread workgroup per CU, two trips of loads in flight
                                                                 // (two workgroups per CU, 8 waves per SIMD, measured slower in round 4:
                                                                 // k_scan 25.0 -> 28.7 us at 2560x1920 fp64, 60.1 -> 65.0 at 5120x3840 fp32)
    const dim3 pg(2 + SB_SEG_PARTS), pb(PREP_NT);
    // Single-domain calls run the contrast first and let k_wind apply the thresholds and the state update
    // (job.wind_final); a band step must run k_scan + k_wind before its ghost rows arrive, so there the contrast
    // kernel applies them.
    if (job.wind_final && ph1 && ph2 && !gathered) {
        SB_EV_BEGIN(SB_PROF_SCAN);
        launch_scan<T>(job, nblk, lc.partials, !reuse, st);
        SB_EV_END(SB_PROF_SCAN);
        // host-model flavour: the strip kernel does k_prep's work itself (one dependent launch less on the critical path)
        if (job.strip && job.t0_fly && !lc.no_fold && nblk <= 1024) {
            DiagJob<T> fj = job;
            fj.fold = 1;
            fj.fold_partials = lc.partials;
            fj.fold_nparts = reuse ? 0 : nblk;
            fj.stats_out = (T *)lc.stats;
            SB_EV_BEGIN(SB_PROF_THC);
            if ((e = launch_contrast<T>(fj, H, lc.ncu, st)) != hipSuccess) return e;
            SB_EV_END(SB_PROF_THC);
            SB_EV_BEGIN(SB_PROF_WIND);
            launch_wind<T>(job, lc.ncu, st);
            SB_EV_END(SB_PROF_WIND);
            if (lc.launches) *lc.launches += 3;
            return hipGetLastError();
        }
        SB_EV_BEGIN(SB_PROF_PREP);
        hipLaunchKernelGGL(k_prep<T>, pg, pb, 0, st, job, (const Moments *)lc.partials, reuse ? 0 : nblk, (T *)lc.stats, (Moments *)nullptr);
        SB_EV_END(SB_PROF_PREP);
        nl += 2;
        if (!job.t0_fly) {
            SB_EV_BEGIN(SB_PROF_T0);
            hipLaunchKernelGGL(k_t0<T>, dim3((g.nxh + 255) / 256, g.nyh), dim3(256), 0, st, job);
            SB_EV_END(SB_PROF_T0);
            ++nl;
        }
        SB_EV_BEGIN(SB_PROF_THC);
        if ((e = launch_contrast<T>(job, H, lc.ncu, st)) != hipSuccess) return e;
        SB_EV_END(SB_PROF_THC);
        SB_EV_BEGIN(SB_PROF_WIND);
        launch_wind<T>(job, lc.ncu, st);
        SB_EV_END(SB_PROF_WIND);
        nl += 2;
        if (lc.launches) *lc.launches += nl;
        return hipGetLastError();
    }
    // ---- a band step on the strip kernel: k_scan ahead of the join; behind it the contrast -- it merges the moments
    // gathered from all ranks and compacts k_wind's segment lists itself -- and k_wind with the update: the three
    // kernels of a single-domain call
    if (job.wind_final) {
        if (ph1) {
            const bool publish = gathered && lc.moments_out != nullptr && !reuse;     // this band's moments, for the all-gather
            DiagJob<T> sj = job;
            if (publish) { sj.moments_out = lc.moments_out; sj.stats_ticket = lc.stats_ticket; }
            launch_scan<T>(sj, nblk, lc.partials, publish, st);
            if (publish && lc.moments_event && (e = hipEventRecord(lc.moments_event, st)) != hipSuccess) return e;
            ++nl;
        }
        if (ph2) {
            DiagJob<T> fj = job;
            fj.fold = 1;
            fj.fold_partials = nullptr;
            fj.fold_nparts = 0;
            if (gathered && !reuse) { fj.gath = lc.gathered; fj.ngath = lc.ngathered; }
            fj.stats_out = (T *)lc.stats;
            if ((e = launch_contrast<T>(fj, H, lc.ncu, st)) != hipSuccess) return e;
            launch_wind<T>(job, lc.ncu, st);
            nl += 2;
        }
        if (lc.launches) *lc.launches += nl;
        return hipGetLastError();
    }
    // ---- phase 1: needs neither theta's ghost cells nor the statistics of the other bands ---------
    if (ph1) {
        // a band step takes this band's own sigma moments from the same pass (lc.moments_out), publishes
        // them for the all-gather and signals the communication stream
        const bool own_stats = (!gathered || lc.moments_out != nullptr) && !reuse;
        // (a band step: k_scan's last workgroup merges and publishes the moments itself -- the all-gather can start
        // behind this one kernel)
        const bool scan_publishes = own_stats && gathered && lc.moments_out != nullptr;
        DiagJob<T> sj = job;
        if (scan_publishes) { sj.moments_out = lc.moments_out; sj.stats_ticket = lc.stats_ticket; }
        SB_EV_BEGIN(SB_PROF_SCAN);
        launch_scan<T>(sj, nblk, lc.partials, own_stats, st);
        SB_EV_END(SB_PROF_SCAN);
        if (scan_publishes && lc.moments_event && (e = hipEventRecord(lc.moments_event, st)) != hipSuccess) return e;
        const bool lists_stand = lc.segs_stand;
        if (!lists_stand) {
            SB_EV_BEGIN(SB_PROF_PREP);
            hipLaunchKernelGGL(k_prep<T>, pg, pb, 0, st, job, (const Moments *)lc.partials, own_stats && !scan_publishes ? nblk : 0,
                               (T *)lc.stats, (Moments *)nullptr);
            SB_EV_END(SB_PROF_PREP);
            ++nl;
        }
        DiagJob<T> wj = job;
        wj.seg_trust = lists_stand ? 1 : 0;
        SB_EV_BEGIN(SB_PROF_WIND);
        launch_wind<T>(wj, lc.ncu, st);
        SB_EV_END(SB_PROF_WIND);
        nl += 2;
    }
    // ---- phase 2: statistics of all bands, theta with its ghost cells -------------------------
    if (ph2) {
        DiagJob<T> pj = job;
        if (gathered && !reuse) {
            if (job.t0_fly) {                                    // the contrast kernel merges the gathered moments in its prologue
                pj.gath = lc.gathered;
                pj.ngath = lc.ngathered;
                pj.stats_out = (T *)lc.stats;
            } else {                                             // k_t0 needs the scalars first
                hipLaunchKernelGGL(k_merge_moments<T>, dim3(1), dim3(SB_WAVE), 0, st, lc.gathered, lc.ngathered, (T *)lc.stats);
                ++nl;
            }
        }
        if (!job.t0_fly) {
            SB_EV_BEGIN(SB_PROF_T0);
            hipLaunchKernelGGL(k_t0<T>, dim3((g.nxh + 255) / 256, g.nyh), dim3(256), 0, st, job);
            SB_EV_END(SB_PROF_T0);
            ++nl;
        }
        // (the strip kernel leaves the contrast in thc whatever the order -- a loop that loads a cell's winds and state
        // next to the prefetched blocks of the march drains them at every step -- and applies thresholds and state
        // update behind its march, cell list by cell list)
        if (job.strip) {
            pj.wind_final = 1;
            pj.strip_update = 1;
            // ... and compacts the segment lists: this call's update kernel reads them, and the next call's k_wind
            // if the planes stand (no k_prep then)
            if (job.t0_fly && !lc.no_fold) { pj.fold = 1; pj.fold_partials = nullptr; pj.fold_nparts = 0; if (!pj.stats_out) pj.stats_out = (T *)lc.stats; }
        }
        SB_EV_BEGIN(SB_PROF_THC);
        if ((e = launch_contrast<T>(pj, H, lc.ncu, st)) != hipSuccess) return e;
        SB_EV_END(SB_PROF_THC);
        ++nl;
    }
    if (lc.launches) *lc.launches += nl;
    return hipGetLastError();
}

template hipError_t sb_launch_stats<float>(const float *, int, int, int, size_t, Moments *, float *, Moments *, int *,
                                           hipStream_t);
template hipError_t sb_launch_stats<double>(const double *, int, int, int, size_t, Moments *, double *, Moments *, int *,
                                            hipStream_t);
template hipError_t sb_launch_sigmoid_apply<float>(const float *, float *, size_t, const float *, hipStream_t);
template hipError_t sb_launch_sigmoid_apply<double>(const double *, double *, size_t, const double *, hipStream_t);
template hipError_t sb_launch_diag<float>(const DiagJob<float> &, int, const SbLaunchCtx &);
template hipError_t sb_launch_diag<double>(const DiagJob<double> &, int, const SbLaunchCtx &);
