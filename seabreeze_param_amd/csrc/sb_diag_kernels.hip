// sb_diag_kernels.hip -- the seabreeze_diag hot path as CDNA4 (gfx950) kernels.
//
// One call of seabreeze_diag / diag (ref: generic/sea_breeze_diag.f90:55-271,
// python_wrapper/seabreezediag/seabreeze_diag_python.f90:49-285) is four launches on
// one stream:
//
//   k_stats  sigma -> (mean, M2, min, max) -> std, r          [HBM stream, 1 field]
//   k_prep   t0 = theta - (gmma*z)*sigmoid(sigma); class + band bit planes; the
//            fill value for every cell outside the coastal band   [HBM stream]
//   k_thc    per 64x32 tile that touches the band: summed-area tables of t0 in LDS,
//            expanding-window land/sea contrast -> thc            [LDS bound]
//   k_wind   per band cell: level nearest target_plev in the p column, wind
//            speed/direction, thresholds, state update -> sb_con  [HBM gather]
//
// Memory-bound integer/fp64 work: no MFMA anywhere.
#include "sb_device.hpp"
#include "sb_launch.hpp"

// ------------------------------------------------------------------------------------
// k_stats
// ------------------------------------------------------------------------------------
#define STATS_NT 1024

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

// merge across the waves of a workgroup; the result is valid in thread 0
__device__ __forceinline__ Moments block_merge(Moments m, Moments *wpart) {
    m = wave_merge(m);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();                             // wpart may still be read from a previous use
    if (lane == 0) wpart[wv] = m;
    __syncthreads();
    Moments r = moments_empty();
    if (threadIdx.x < SB_WAVE) {
        if (threadIdx.x < STATS_NT / SB_WAVE) r = wpart[threadIdx.x];
        r = wave_merge(r);
    }
    return r;
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

// One workgroup per CU streams its share of sigma with 8 independent loads per thread in
// flight.  Sums are taken about a common shift c (the first interior value), so the inner
// loop is two adds and a multiply per element; the shifted sums become (n, mean, M2) once
// per thread and are merged pairwise (Chan) from there on.  The last workgroup to take a
// ticket merges the per-workgroup partials in index order, so the result does not depend
// on arrival order.
template <typename T>
__global__ __launch_bounds__(STATS_NT) void k_stats(const T *__restrict__ ary, int nx, int ny, int ld,
                                                    size_t off0, Moments *__restrict__ partials,
                                                    unsigned int *__restrict__ ticket,
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
    Moments acc = moments_empty();
    if (cnt > 0) {
        acc.n = (double)cnt;
        acc.mean = c + s1 / acc.n;
        acc.m2 = s2 - s1 * s1 / acc.n;
        acc.mn = mn;
        acc.mx = mx;
    }
    __shared__ Moments wpart[STATS_NT / SB_WAVE];
    __shared__ bool is_last;
    acc = block_merge(acc, wpart);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = acc;
        // publish: agent-scope release, drained, then the ticket
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned int t = atomicAdd(ticket, 1u);
        is_last = (t == gridDim.x - 1);
    }
    __syncthreads();
    if (!is_last) return;
    // last workgroup to arrive: every thread fetches one partial, then a fixed merge tree
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    Moments m = moments_empty();
    for (int b = threadIdx.x; b < (int)gridDim.x; b += STATS_NT) {
        const Moments *pp = &partials[b];
        Moments o;
        o.n = __builtin_nontemporal_load(&pp->n);
        o.mean = __builtin_nontemporal_load(&pp->mean);
        o.m2 = __builtin_nontemporal_load(&pp->m2);
        o.mn = __builtin_nontemporal_load(&pp->mn);
        o.mx = __builtin_nontemporal_load(&pp->mx);
        m = moments_merge(m, o);
    }
    m = block_merge(m, wpart);
    if (threadIdx.x == 0) {
        if (moments_out) *moments_out = m;       // band-local moments for the multi-GPU gather
        else sigmoid_scalars<T>(m, stats);
        *ticket = 0u;                            // re-arm for the next call on this stream
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

// ------------------------------------------------------------------------------------
// k_sigmoid_apply: sm = 1/(1+exp(-std*(ary-r)))   ref: generic/sea_breeze_diag.f90:480
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_sigmoid_apply(const T *__restrict__ ary, T *__restrict__ sm,
                                                       size_t n, const T *__restrict__ stats) {
    const T sd = stats[0], r = stats[1];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        sm[i] = T(1) / (T(1) + exp(-sd * (ary[i] - r)));
}

// ------------------------------------------------------------------------------------
// k_prep: one thread per cell of the (nxh, nyh) arrays; a wave is one aligned 64-cell
// longitude segment, so its ballots are exactly the words of the bit planes.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_prep(DiagJob<T> job) {
    const Geo g = job.g;
    const int X = blockIdx.x * 256 + threadIdx.x;
    const int Y = blockIdx.y;
    const bool in = X < g.nxh;
    const T gmma = T(-0.0060956);                // ref: generic/sea_breeze_diag.f90:138
    T m = T(0), t0v = T(0);
    bool cls = false, band = false;
    const int xi = X - g.h, yi = Y - g.h;        // interior coordinates
    const bool interior = in && xi >= 0 && xi < g.nx && yi >= 0 && yi < g.ny;
    if (in) {
        const size_t idx = (size_t)Y * g.nxh + X;
        const T sd = job.stats[0], r = job.stats[1];
        const T sg = job.sigma[idx];
        const T smod = T(1) / (T(1) + exp(-sd * (sg - r)));
        t0v = job.theta[idx] - ((gmma * job.z[idx]) * smod);   // ref :167
        job.t0[idx] = t0v;
        m = job.mask[idx];
        cls = (m >= T(0));                                     // ref :182, :200
        band = interior && yi < g.rows && !(fabs(m) > job.maxdist);   // ref :174
    }
    const uint64_t wc = __ballot(cls);
    const uint64_t wb = __ballot(band);
    // raise the flag of every thc tile this segment's band cells fall in (it straddles two
    // tile columns when the ghost width is not a multiple of 64); plain stores of 1
    const int tA = ((X & ~63) - g.h) >> 6;
    const uint64_t mA = __ballot(band && (xi >> 6) == tA);
    const uint64_t mB = __ballot(band && (xi >> 6) == tA + 1);
    if ((threadIdx.x & 63) == 0 && (X >> 6) < g.nw) {
        job.clsbits[(size_t)Y * g.nw + (X >> 6)] = wc;
        job.bandbits[(size_t)Y * g.nw + (X >> 6)] = wb;
        if (wb) {
            const int trow = (yi / job.thc_ty) * job.thc_ntx;
            if (mA) job.tile_nnmax[trow + tA] = 1;
            if (mB) job.tile_nnmax[trow + tA + 1] = 1;
        }
    }
    if (interior && yi < g.rows) {
        const size_t o = (size_t)yi * g.nx + xi;
        if (job.flavour == SB_FLAVOUR_GENERIC) {
            if (!band) job.sb_con[o] = job.fill;               // ref :176
        } else {
            // packed output planes, ref: seabreeze_diag_python.f90:277-280
            const size_t pl = (size_t)g.nx * g.ny;
            job.out[pl + o] = t0v;
            if (!band) {
                job.out[o] = job.fill;                         // 2.0E20, ref :173
                job.out[2 * pl + o] = job.ws[o];
                job.out[3 * pl + o] = job.wd[o];
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// k_wind: one thread per interior cell, a wave per 64-cell row segment; waves whose
// band word is empty leave at once, so HBM traffic is the band cells' p columns only.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_wind(DiagJob<T> job) {
    const Geo g = job.g;
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= g.nx) return;
    if (!sb_bit(job.bandbits, g.nw, x + g.h, y + g.h)) return;
    const size_t pl = (size_t)g.nx * g.ny;
    const size_t o = (size_t)y * g.nx + x;
    const int nz = job.nz;

    // level nearest the target pressure: first minimum of |p - target|
    // ref: generic/sea_breeze_diag.f90:223 (per column), seabreeze_diag_python.f90:228 (1-D p)
    int lev = 0;
    if (job.flavour == SB_FLAVOUR_GENERIC) {
        // 14 independent streaming loads in flight per lane (4 trips for the 56-level stub
        // layout, ref: generic/get_all_fields_mod.f90:9); each level is a separate plane so
        // a wave reads one contiguous 64-cell segment per level.  Non-temporal: the column
        // is read once per call.
        const T *pc = job.p + o;
        T best = fabs(__builtin_nontemporal_load(pc) - job.target_plev);
        int k = 1;
        constexpr int UN = 14;
        for (; k + UN <= nz; k += UN) {
            T d[UN];
#pragma unroll
            for (int q = 0; q < UN; ++q) d[q] = __builtin_nontemporal_load(pc + (size_t)(k + q) * pl);
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                const T a = fabs(d[q] - job.target_plev);
                if (a < best) { best = a; lev = k + q; }
            }
        }
        for (; k < nz; ++k) {
            const T a = fabs(__builtin_nontemporal_load(pc + (size_t)k * pl) - job.target_plev);
            if (a < best) { best = a; lev = k; }
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
    const T rad2deg = T(57.2957);                               // ref :128 (sic)
    const T n_ws = sqrt(uu * uu + vv * vv);                     // ref :225
    const T n_wd = atan2(-uu, -vv) * rad2deg;                   // ref :227
    const T n_thc = job.thc[o];                                 // written by k_thc
    T ws_old = job.ws[o], wd_old = job.wd[o];
    if (job.tn < 2) { ws_old = n_ws; wd_old = n_wd; }          // ref :235-239
    // ref :242-259
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
    if (job.flavour == SB_FLAVOUR_GENERIC) {
        job.sb_con[o] = sb;
        job.ws[o] = n_ws;                                       // ref :261 (every call)
        if (job.refresh) job.wd[o] = n_wd;                      // ref :264-266
        else if (job.tn < 2) job.wd[o] = wd_old;
    } else {
        // ref: seabreeze_diag_python.f90:268-280
        T ws_new = ws_old, wd_new = wd_old;
        if (job.refresh) { ws_new = n_ws; wd_new = n_wd; }
        if (job.refresh || job.tn < 2) { job.ws[o] = ws_new; job.wd[o] = wd_new; }
        job.out[o] = sb;
        job.out[2 * pl + o] = ws_new;
        job.out[3 * pl + o] = wd_new;
    }
}

// ------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------
template <typename T>
hipError_t sb_launch_stats(const T *ary, int nx, int ny, int ld, size_t off0, Moments *partials,
                           unsigned int *ticket, T *stats, Moments *moments_out, hipStream_t st) {
    // one workgroup per CU, fewer when the field is small (8 elements per thread per trip)
    const size_t n = (size_t)nx * ny;
    int nblk = (int)((n + (size_t)STATS_NT * 8 - 1) / ((size_t)STATS_NT * 8));
    if (nblk < 1) nblk = 1;
    if (nblk > 256) nblk = 256;
    hipLaunchKernelGGL(k_stats<T>, dim3(nblk), dim3(STATS_NT), 0, st, ary, nx, ny, ld, off0, partials, ticket,
                       stats, moments_out);
    return hipGetLastError();
}

template <typename T>
hipError_t sb_launch_merge_moments(const Moments *parts, int nparts, T *stats, hipStream_t st) {
    hipLaunchKernelGGL(k_merge_moments<T>, dim3(1), dim3(SB_WAVE), 0, st, parts, nparts, stats);
    return hipGetLastError();
}

template <typename T>
hipError_t sb_launch_sigmoid_apply(const T *ary, T *sm, size_t n, const T *stats, hipStream_t st) {
    int nblk = (int)((n + 255) / 256);
    if (nblk > 4096) nblk = 4096;
    hipLaunchKernelGGL(k_sigmoid_apply<T>, dim3(nblk), dim3(256), 0, st, ary, sm, n, stats);
    return hipGetLastError();
}

template <typename T>
hipError_t sb_launch_diag(const DiagJob<T> &job, int H, Moments *partials, unsigned int *ticket, T *stats,
                          hipStream_t st, hipEvent_t *ev, const Moments *gathered, int ngathered, int ncu) {
    const Geo &g = job.g;
    if (ev) (void)hipEventRecord(ev[0], st);
    // sigmoid statistics: over the interior of sigma, or merged from the bands' gathered moments
    hipError_t e = gathered ? sb_launch_merge_moments<T>(gathered, ngathered, stats, st)
                            : sb_launch_stats<T>(job.sigma, g.nx, g.ny, g.nxh, (size_t)g.h * g.nxh + g.h, partials,
                                                 ticket, stats, nullptr, st);
    if (e != hipSuccess) return e;
    if (ev) (void)hipEventRecord(ev[1], st);
    hipLaunchKernelGGL(k_prep<T>, dim3((g.nxh + 255) / 256, g.nyh), dim3(256), 0, st, job);
    if (ev) (void)hipEventRecord(ev[2], st);
    e = sb_launch_thc<T>(job, H, ncu, st);
    if (e != hipSuccess) return e;
    if (ev) (void)hipEventRecord(ev[3], st);
    hipLaunchKernelGGL(k_wind<T>, dim3((g.nx + 255) / 256, g.rows), dim3(256), 0, st, job);
    if (ev) (void)hipEventRecord(ev[4], st);
    return hipGetLastError();
}

template hipError_t sb_launch_stats<float>(const float *, int, int, int, size_t, Moments *, unsigned int *, float *,
                                           Moments *, hipStream_t);
template hipError_t sb_launch_stats<double>(const double *, int, int, int, size_t, Moments *, unsigned int *,
                                            double *, Moments *, hipStream_t);
template hipError_t sb_launch_sigmoid_apply<float>(const float *, float *, size_t, const float *, hipStream_t);
template hipError_t sb_launch_sigmoid_apply<double>(const double *, double *, size_t, const double *, hipStream_t);
template hipError_t sb_launch_diag<float>(const DiagJob<float> &, int, Moments *, unsigned int *, float *,
                                          hipStream_t, hipEvent_t *, const Moments *, int, int);
template hipError_t sb_launch_diag<double>(const DiagJob<double> &, int, Moments *, unsigned int *, double *,
                                           hipStream_t, hipEvent_t *, const Moments *, int, int);
