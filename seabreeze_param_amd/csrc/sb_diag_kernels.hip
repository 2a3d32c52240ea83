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
#define STATS_NT 256

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

template <typename T>
__global__ __launch_bounds__(STATS_NT) void k_stats(const T *__restrict__ ary, int nx, int ny, int ld,
                                                    size_t off0, Moments *__restrict__ partials,
                                                    unsigned int *__restrict__ ticket,
                                                    T *__restrict__ stats) {
    // grid-stride over interior cells, 4 independent loads per thread per trip
    const size_t n = (size_t)nx * ny;
    const size_t stride = (size_t)gridDim.x * STATS_NT;
    Moments acc = moments_empty();
    for (size_t base = (size_t)blockIdx.x * STATS_NT + threadIdx.x; base < n; base += 4 * stride) {
        double x[4];
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            size_t i = base + q * stride;
            if (i < n) {
                size_t row = i / nx, col = i - row * nx;
                x[cnt++] = (double)ary[off0 + row * ld + col];
            }
        }
        // exact two-pass moments of the (<=4) register values, then one merge
        double s = 0.0;
        for (int q = 0; q < cnt; ++q) s += x[q];
        Moments b;
        b.n = (double)cnt;
        b.mean = s / (double)cnt;
        b.m2 = 0.0;
        b.mn = x[0];
        b.mx = x[0];
        for (int q = 0; q < cnt; ++q) {
            double d = x[q] - b.mean;
            b.m2 += d * d;
            b.mn = x[q] < b.mn ? x[q] : b.mn;
            b.mx = x[q] > b.mx ? x[q] : b.mx;
        }
        acc = moments_merge(acc, b);
    }
    __shared__ Moments wpart[STATS_NT / SB_WAVE];
    __shared__ bool is_last;
    acc = wave_merge(acc);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wpart[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        Moments m = wpart[0];
        for (int w = 1; w < STATS_NT / SB_WAVE; ++w) m = moments_merge(m, wpart[w]);
        partials[blockIdx.x] = m;
        // publish: agent-scope release, drained, then the ticket
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned int t = atomicAdd(ticket, 1u);
        is_last = (t == gridDim.x - 1);
    }
    __syncthreads();
    if (!is_last) return;
    // last block to arrive: merge all partials in index order (deterministic tree)
    if (wv == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        Moments m = moments_empty();
        for (int b = lane; b < (int)gridDim.x; b += SB_WAVE) {
            const Moments *pp = &partials[b];
            Moments o;
            o.n = __builtin_nontemporal_load(&pp->n);
            o.mean = __builtin_nontemporal_load(&pp->mean);
            o.m2 = __builtin_nontemporal_load(&pp->m2);
            o.mn = __builtin_nontemporal_load(&pp->mn);
            o.mx = __builtin_nontemporal_load(&pp->mx);
            m = moments_merge(m, o);
        }
        m = wave_merge(m);
        if (lane == 0) {
            // std = 2/sqrt(var/N), r = (max-min)/4 in the working precision
            // ref: generic/sea_breeze_diag.f90:478-479
            const T var = (T)m.m2;
            const T cnt = (T)(nx * ny);          // reference: default-integer product
            stats[0] = T(2) / sqrt(var / cnt);
            stats[1] = ((T)m.mx - (T)m.mn) / T(4);
            stats[2] = (T)m.mean;
            stats[3] = var;
            *ticket = 0u;                        // re-arm for the next call on this stream
        }
    }
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
    if ((threadIdx.x & 63) == 0 && (X >> 6) < g.nw) {
        job.clsbits[(size_t)Y * g.nw + (X >> 6)] = wc;
        job.bandbits[(size_t)Y * g.nw + (X >> 6)] = wb;
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
// Global-memory search for cells whose window outgrows the LDS tile (rare).
// Rings are accumulated from the centre outwards; the first radius >= 1 at which the
// square holds both classes is the reference's final nn (its sums restart at every
// radius, so only that last square matters; ref: generic/sea_breeze_diag.f90:191-216).
// ------------------------------------------------------------------------------------
template <typename T>
__device__ T contrast_global(const DiagJob<T> &job, int x, int y, int cap, int &nn_used, bool &one_class) {
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
    // direct sums over the final square, offset by the centre value to keep the
    // accumulations small
    sb_map_cell(g, x, y, X, Y);
    const double c0 = (double)job.t0[(size_t)Y * g.nxh + X];
    double sl = 0.0, ss = 0.0;
    double nl = 0.0, ns = 0.0;
    for (int yy = y - nn; yy <= y + nn; ++yy)
        for (int xx = x - nn; xx <= x + nn; ++xx) {
            if (!sb_map_cell(g, xx, yy, X, Y)) continue;
            const double d = (double)job.t0[(size_t)Y * g.nxh + X] - c0;
            if (sb_bit(job.clsbits, g.nw, X, Y)) { sl += d; nl += 1.0; } else { ss += d; ns += 1.0; }
        }
    return (T)(sl / nl - ss / ns);               // 0/0 -> NaN when a class is missing
}

// ------------------------------------------------------------------------------------
// k_thc: thermal heating contrast on TX x TY tiles with an LDS halo of H cells.
// ------------------------------------------------------------------------------------
template <typename T, int TX, int TY, int H, int NT>
__global__ __launch_bounds__(NT) void k_thc(DiagJob<T> job) {
    constexpr int W = TX + 2 * H, HT = TY + 2 * H, P = W + 1;
    constexpr int CPT = TX * TY / NT;            // cells per thread
    static_assert(TX == 64, "a wave owns one 64-cell row segment");
    static_assert((TX * TY) % NT == 0 && NT % TX == 0, "tile/thread shape");
    static_assert((size_t)W * HT < 65536, "u16 count table");
    __shared__ double sA[(HT + 1) * P];          // SAT of (t0 - c0), every cell
    __shared__ double sL[(HT + 1) * P];          // SAT of (t0 - c0), land-side cells
    __shared__ unsigned short sC[(HT + 1) * P];  // SAT of land-side count

    const Geo g = job.g;
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const int lx = tid % TX, ly0 = tid / TX;

    // which of my cells are in the band?
    unsigned mine = 0;
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int x = x0 + lx, y = y0 + ly0 + q * (NT / TX);
        if (x < g.nx && y < g.rows && sb_bit(job.bandbits, g.nw, x + g.h, y + g.h)) mine |= 1u << q;
    }
    if (!__syncthreads_or((int)mine)) {
        if (tid == 0) job.tile_nnmax[blockIdx.y * gridDim.x + blockIdx.x] = 0;
        return;
    }

    // ---- stage the tile + halo through the index map ---------------------------------
    int X, Y;
    sb_map_cell(g, x0, y0, X, Y);
    const double c0 = (double)job.t0[(size_t)Y * g.nxh + X];
    for (int i = tid; i < HT + 1; i += NT) { sA[i * P] = 0.0; sL[i * P] = 0.0; sC[i * P] = 0; }
    for (int i = tid; i < P; i += NT) { sA[i] = 0.0; sL[i] = 0.0; sC[i] = 0; }
    for (int i = tid; i < W * HT; i += NT) {
        const int r = i / W, c = i - r * W;
        double d = 0.0;
        int land = 0;
        if (sb_map_cell(g, x0 - H + c, y0 - H + r, X, Y)) {
            d = (double)job.t0[(size_t)Y * g.nxh + X] - c0;
            land = sb_bit(job.clsbits, g.nw, X, Y);
        }
        const int o = (r + 1) * P + c + 1;
        sA[o] = d;
        sL[o] = land ? d : 0.0;
        sC[o] = (unsigned short)land;
    }
    __syncthreads();
    // ---- prefix along latitude: one task per (table, column) ------------------------
    for (int task = tid; task < 3 * W; task += NT) {
        const int a = task / W, c = task - a * W + 1;
        if (a == 0) { double s = 0.0; for (int r = 1; r <= HT; ++r) { s += sA[r * P + c]; sA[r * P + c] = s; } }
        else if (a == 1) { double s = 0.0; for (int r = 1; r <= HT; ++r) { s += sL[r * P + c]; sL[r * P + c] = s; } }
        else { unsigned s = 0; for (int r = 1; r <= HT; ++r) { s += sC[r * P + c]; sC[r * P + c] = (unsigned short)s; } }
    }
    __syncthreads();
    // ---- prefix along longitude: one task per (table, row) --------------------------
    for (int task = tid; task < 3 * HT; task += NT) {
        const int a = task / HT, r = task - a * HT + 1;
        if (a == 0) { double s = 0.0; for (int c = 1; c <= W; ++c) { s += sA[r * P + c]; sA[r * P + c] = s; } }
        else if (a == 1) { double s = 0.0; for (int c = 1; c <= W; ++c) { s += sL[r * P + c]; sL[r * P + c] = s; } }
        else { unsigned s = 0; for (int c = 1; c <= W; ++c) { s += sC[r * P + c]; sC[r * P + c] = (unsigned short)s; } }
    }
    __syncthreads();

    // ---- expanding-window search, O(1) per radius ------------------------------------
    int nnmax = 0;
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        if (!(mine & (1u << q))) continue;
        const int ly = ly0 + q * (NT / TX);
        const int x = x0 + lx, y = y0 + ly;
        const int cx = lx + H, cy = ly + H;
        int lim = H;
        if (g.bnd == BND_HALO) {
            int e = min(min(x + g.h, g.nx - 1 - x + g.h), min(y + g.h, g.ny - 1 - y + g.h));
            lim = min(lim, e);
        }
        int nn = 1, nl = 0, area = 0;
        bool found = false;
        for (; nn <= lim; ++nn) {
            const int r0 = (cy - nn) * P, r1 = (cy + nn + 1) * P, c0i = cx - nn, c1i = cx + nn + 1;
            nl = (int)sC[r1 + c1i] - (int)sC[r0 + c1i] - (int)sC[r1 + c0i] + (int)sC[r0 + c0i];
            area = (2 * nn + 1) * (2 * nn + 1);
            if (nl > 0 && nl < area) { found = true; break; }
        }
        T contrast;
        if (found) {
            const int r0 = (cy - nn) * P, r1 = (cy + nn + 1) * P, c0i = cx - nn, c1i = cx + nn + 1;
            const double RL = (sL[r1 + c1i] - sL[r0 + c1i]) - (sL[r1 + c0i] - sL[r0 + c0i]);
            const double RA = (sA[r1 + c1i] - sA[r0 + c1i]) - (sA[r1 + c0i] - sA[r0 + c0i]);
            contrast = (T)(RL / (double)nl - (RA - RL) / (double)(area - nl));
        } else {
            int cap = g.nx + g.ny;
            if (g.bnd == BND_HALO)
                cap = min(min(x + g.h, g.nx - 1 - x + g.h), min(y + g.h, g.ny - 1 - y + g.h));
            bool one_class;
            contrast = contrast_global(job, x, y, cap, nn, one_class);
            atomicAdd(&job.counters[0], 1);
            if (one_class) atomicAdd(&job.counters[1], 1);
        }
        nnmax = max(nnmax, nn);
        const T mul = sb_bit(job.clsbits, g.nw, x + g.h, y + g.h) ? T(1) : T(-1);   // ref :182-186
        job.thc[(size_t)y * g.nx + x] = mul * contrast;        // ref :216, :262
    }
    // per-tile largest radius (diagnostic; reduced lazily by sb_last_counters)
    __shared__ int s_nn;
    if (tid == 0) s_nn = 0;
    __syncthreads();
    if (nnmax) atomicMax(&s_nn, nnmax);
    __syncthreads();
    if (tid == 0) job.tile_nnmax[blockIdx.y * gridDim.x + blockIdx.x] = s_nn;
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
        const T *pc = job.p + o;
        T best = fabs(pc[0] - job.target_plev);
        int k = 1;
        for (; k + 8 <= nz; k += 8) {
            T d[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) d[q] = pc[(size_t)(k + q) * pl];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const T a = fabs(d[q] - job.target_plev);
                if (a < best) { best = a; lev = k + q; }
            }
        }
        for (; k < nz; ++k) {
            const T a = fabs(pc[(size_t)k * pl] - job.target_plev);
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
                           unsigned int *ticket, T *stats, hipStream_t st) {
    const size_t n = (size_t)nx * ny;
    int nblk = (int)((n + (size_t)STATS_NT * 8 - 1) / ((size_t)STATS_NT * 8));
    if (nblk < 1) nblk = 1;
    if (nblk > SB_STATS_MAX_BLOCKS) nblk = SB_STATS_MAX_BLOCKS;
    hipLaunchKernelGGL(k_stats<T>, dim3(nblk), dim3(STATS_NT), 0, st, ary, nx, ny, ld, off0, partials, ticket,
                       stats);
    return hipGetLastError();
}

template <typename T>
hipError_t sb_launch_sigmoid_apply(const T *ary, T *sm, size_t n, const T *stats, hipStream_t st) {
    int nblk = (int)((n + 255) / 256);
    if (nblk > 4096) nblk = 4096;
    hipLaunchKernelGGL(k_sigmoid_apply<T>, dim3(nblk), dim3(256), 0, st, ary, sm, n, stats);
    return hipGetLastError();
}

template <typename T, int H>
static void launch_thc(const DiagJob<T> &job, hipStream_t st) {
    constexpr int TX = 64, TY = (H <= 16 ? 32 : 16), NT = 512;
    dim3 grid((job.g.nx + TX - 1) / TX, (job.g.rows + TY - 1) / TY);
    hipLaunchKernelGGL((k_thc<T, TX, TY, H, NT>), grid, dim3(NT), 0, st, job);
}

template <typename T>
void sb_thc_tiles(int nx, int rows, int H, int &tx, int &ty) {
    const int TY = (H <= 16 ? 32 : 16);
    tx = (nx + 63) / 64;
    ty = (rows + TY - 1) / TY;
}

template <typename T>
hipError_t sb_launch_diag(const DiagJob<T> &job, int H, Moments *partials, unsigned int *ticket, T *stats,
                          hipStream_t st) {
    const Geo &g = job.g;
    // sigmoid statistics over the interior of sigma
    hipError_t e = sb_launch_stats<T>(job.sigma, g.nx, g.ny, g.nxh, (size_t)g.h * g.nxh + g.h, partials, ticket,
                                      stats, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_prep<T>, dim3((g.nxh + 255) / 256, g.nyh), dim3(256), 0, st, job);
    if (H <= 8) launch_thc<T, 8>(job, st);
    else if (H <= 16) launch_thc<T, 16>(job, st);
    else launch_thc<T, 24>(job, st);
    hipLaunchKernelGGL(k_wind<T>, dim3((g.nx + 255) / 256, g.rows), dim3(256), 0, st, job);
    return hipGetLastError();
}

template hipError_t sb_launch_stats<float>(const float *, int, int, int, size_t, Moments *, unsigned int *, float *,
                                           hipStream_t);
template hipError_t sb_launch_stats<double>(const double *, int, int, int, size_t, Moments *, unsigned int *,
                                            double *, hipStream_t);
template hipError_t sb_launch_sigmoid_apply<float>(const float *, float *, size_t, const float *, hipStream_t);
template hipError_t sb_launch_sigmoid_apply<double>(const double *, double *, size_t, const double *, hipStream_t);
template hipError_t sb_launch_diag<float>(const DiagJob<float> &, int, Moments *, unsigned int *, float *,
                                          hipStream_t);
template hipError_t sb_launch_diag<double>(const DiagJob<double> &, int, Moments *, unsigned int *, double *,
                                           hipStream_t);
template void sb_thc_tiles<float>(int, int, int, int &, int &);
template void sb_thc_tiles<double>(int, int, int, int &, int &);
