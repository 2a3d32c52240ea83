// sb_thc_kernel.hip -- thermal heating contrast (the expanding-window land/sea mean
// difference of t0) on gfx950.   ref: generic/sea_breeze_diag.f90:188-216,
// python_wrapper/seabreezediag/seabreeze_diag_python.f90:187-221
//
// The reference re-sums a (2nn+1)^2 window from scratch at every radius nn until it holds
// both classes; only the last square matters.  Here a tile of 64 x TY cells plus a halo of
// H cells is staged once into LDS as three summed-area tables (all cells, land-side
// cells, land-side count), after which any square costs four LDS reads per table and the
// smallest valid radius is found by bisection (the "both classes present" predicate is
// monotone in nn).  Sums are taken about a per-tile offset c0, which cancels exactly in
// the difference of the two means and keeps the fp64 prefix sums small.
//
// One persistent 1024-thread workgroup per CU (the tables take 113 KB of the 160 KB LDS):
// every workgroup compacts the tile flags k_prep raised into the ordered list of active
// tiles and takes entries blockIdx, blockIdx + gridDim, ... -- tiles that do not touch the
// coastal band (about 3 in 4) cost nothing, and the active ones are dealt out evenly.
#include "sb_device.hpp"
#include "sb_launch.hpp"

#ifdef SB_STAMPS
#define SB_STAMP(i) \
    do { if (threadIdx.x == 0) job.stamps[(size_t)tile * 8 + (i)] = clock64(); } while (0)
#else
#define SB_STAMP(i) do { } while (0)
#endif

#define THC_NT 1024
#define THC_MAXMINE 256          // active tiles one workgroup can own

// ------------------------------------------------------------------------------------
// Global-memory search for cells whose window outgrows the LDS tile (rare).  Rings are
// accumulated from the centre outwards; the first radius >= 1 at which the square holds
// both classes is the reference's final nn.  cap bounds the radius: the reference loop has
// none and never returns on a one-class grid (SURVEY.md §7 "Hard parts").
// ------------------------------------------------------------------------------------
// t0 of one cell of the ghost-celled frame: from the workspace (f2py flavour) or derived on the spot
template <typename T>
__device__ __forceinline__ T cell_t0(const DiagJob<T> &job, size_t idx) {
    if (!job.t0_fly) return job.t0[idx];
    return sb_t0<T>(job.theta[idx], job.z[idx], job.sigma[idx], job.stats[0], job.stats[1]);
}

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
    sb_map_cell(g, x, y, X, Y);
    const double c0 = (double)cell_t0(job, (size_t)Y * g.nxh + X);
    double sl = 0.0, ss = 0.0, nl = 0.0, ns = 0.0;
    for (int yy = y - nn; yy <= y + nn; ++yy)
        for (int xx = x - nn; xx <= x + nn; ++xx) {
            if (!sb_map_cell(g, xx, yy, X, Y)) continue;
            const double d = (double)cell_t0(job, (size_t)Y * g.nxh + X) - c0;
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
__device__ __forceinline__ int thc_build_list(const int *__restrict__ flags, int ntiles, int *s_mine, int *s_wcnt) {
    constexpr int NT = THC_NT, NWV = THC_NT / SB_WAVE;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int total_active = 0;
    for (int t0i = 0; t0i < ntiles; t0i += NT) {
        const int t = t0i + tid;
        const uint64_t bm = __ballot((t < ntiles) && (flags[t] != 0));
        if (lane == 0) s_wcnt[wv] = __popcll(bm);
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NWV; ++w) total_active += s_wcnt[w];
        __syncthreads();
    }
    const bool xcd_map = (gridDim.x % 8 == 0);
    const int per = (total_active + 7) / 8, nper = (int)gridDim.x / 8;
    int base = 0;
    for (int t0i = 0; t0i < ntiles; t0i += NT) {
        const int t = t0i + tid;
        const bool flag = (t < ntiles) && (flags[t] != 0);
        const uint64_t bm = __ballot(flag);
        if (lane == 0) s_wcnt[wv] = __popcll(bm);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) {
            const int cw = s_wcnt[w];
            before += (w < wv) ? cw : 0;
            total += cw;
        }
        if (flag) {
            const int pos = base + before + __popcll(bm & ((1ull << lane) - 1ull));
            int owner, slot;
            if (xcd_map) {
                const int xq = pos / per, ii = pos - xq * per;
                owner = xq + 8 * (ii % nper);
                slot = ii / nper;
            } else {
                owner = pos % (int)gridDim.x;
                slot = pos / (int)gridDim.x;
            }
            if (owner == (int)blockIdx.x && slot < THC_MAXMINE) s_mine[slot] = t;
        }
        base += total;
        __syncthreads();
    }
    int nmine;
    if (xcd_map) {
        const int xq = (int)blockIdx.x % 8, local = (int)blockIdx.x / 8;
        int cnt = total_active - xq * per;
        cnt = cnt < 0 ? 0 : (cnt > per ? per : cnt);
        nmine = cnt > local ? (cnt - 1 - local) / nper + 1 : 0;
    } else {
        nmine = base > (int)blockIdx.x ? (base - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    }
    return nmine > THC_MAXMINE ? THC_MAXMINE : nmine;    // the launcher sizes the grid so this never binds
}

// ------------------------------------------------------------------------------------
// k_final_tiles: thresholds, scaling and state update (ref: generic/sea_breeze_diag.f90:
// 235-266) for the band cells of every active tile, when k_thc ran side by side with k_wind
// and could not apply them itself.  Same persistent tile walk as k_thc.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(THC_NT) void k_final_tiles(DiagJob<T> job) {
    __shared__ int s_mine[THC_MAXMINE];
    __shared__ int s_wcnt[THC_NT / SB_WAVE];
    const Geo g = job.g;
    const int tid = threadIdx.x, TY = job.thc_ty, ntx = job.thc_ntx;
    const int nmine = thc_build_list(job.tile_nnmax, job.thc_ntx * job.thc_nty, s_mine, s_wcnt);
    __syncthreads();
    for (int mi = 0; mi < nmine; ++mi) {
        const int tile = s_mine[mi];
        const int x0 = (tile % ntx) * 64, y0 = (tile / ntx) * TY;
        for (int i = tid; i < 64 * TY; i += THC_NT) {
            const int x = x0 + (i & 63), y = y0 + (i >> 6);
            if (x < g.nx && y < g.rows && sb_bit(job.bandbits, g.nw, x + g.h, y + g.h)) {
                const size_t o = (size_t)y * g.nx + x;
                sb_trigger_update<T>(job, o, job.thc[o], sb_trigger_load<T>(job, o));
            }
        }
    }
}

template <typename T, int TY, int H, bool FLY, bool FUSE>
__global__ __launch_bounds__(THC_NT) void k_thc(DiagJob<T> job) {
    constexpr int TX = 64, NT = THC_NT;
    constexpr int W = TX + 2 * H, HT = TY + 2 * H, P = W + 1;
    constexpr int CPT = TX * TY / NT;            // cells per thread in the search phase
    constexpr int NWV = NT / SB_WAVE;
    constexpr int RPW = (HT + NWV - 1) / NWV;    // LDS rows staged per wave
    constexpr int NCH = (W + SB_WAVE - 1) / SB_WAVE;   // 64-column chunks per LDS row
    static_assert((TX * TY) % NT == 0 && CPT >= 1, "tile/thread shape");
    static_assert((size_t)W * HT < 65536, "u16 count table");
    static_assert(HT % 16 == 0 && W % 16 == 0, "the scans run in batches of 16");
    __shared__ double sA[(HT + 1) * P];          // SAT of (t0 - c0), every cell
    __shared__ double sL[(HT + 1) * P];          // SAT of (t0 - c0), land-side cells
    __shared__ unsigned short sC[(HT + 1) * P];  // SAT of land-side count
    __shared__ int s_mine[THC_MAXMINE];
    __shared__ int s_wcnt[NWV];
    __shared__ int s_nn;

    const Geo g = job.g;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ntx = job.thc_ntx, ntiles = job.thc_ntx * job.thc_nty;

    int nmine = thc_build_list(job.tile_nnmax, ntiles, s_mine, s_wcnt);
    for (int i = tid; i < HT + 1; i += NT) { sA[i * P] = 0.0; sL[i * P] = 0.0; sC[i * P] = 0; }
    for (int i = tid; i < P; i += NT) { sA[i] = 0.0; sL[i] = 0.0; sC[i] = 0; }
    const bool fastx = g.nx > W + 2;             // one conditional add wraps every column of the tile

    for (int mi = 0; mi < nmine; ++mi) {
        const int tile = s_mine[mi];
        const int x0 = (tile % ntx) * TX, y0 = (tile / ntx) * TY;
        SB_STAMP(0);
        if (tid == 0) s_nn = 1;

        // ---- issue every global load of the tile before touching any result -------------
        // my search cells' band bits, the tile offset c0, and RPW x NCH staged values/words
        const int lx = tid % TX, ly0 = tid / TX;
        uint64_t bw[CPT], cw[CPT];
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const int x = x0 + lx, y = y0 + ly0 + q * (NT / TX);
            bw[q] = 0;
            cw[q] = 0;
            if (x < g.nx && y < g.rows) {
                const size_t wi = (size_t)(y + g.h) * g.nw + ((x + g.h) >> 6);
                bw[q] = job.bandbits[wi];
                cw[q] = job.clsbits[wi];
            }
        }
        // t0 of one cell: read from the workspace (f2py flavour, where the t0 plane is an output),
        // or formed here as theta - gz with gz = (gmma*z)*sigmoid(sigma) (ref: generic/
        // sea_breeze_diag.f90:167) left by k_gz in the same workspace for the tiles that need it
        int X, Y;
        sb_map_cell(g, x0, y0, X, Y);
        const unsigned unxh = (unsigned)g.nxh;
        const unsigned i00 = (unsigned)Y * unxh + (unsigned)X;
        T c_th = T(0), c_gz = T(0), c_t0 = T(0);
        if constexpr (FLY) { c_th = job.theta[i00]; c_gz = job.t0[i00]; }
        else c_t0 = job.t0[i00];
        // array column of each of my NCH chunks (-1: no such cell): the longitude map is the
        // same for every row this thread stages, so it is evaluated once per tile
        int xcol[NCH];
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
            xcol[ch] = ok ? Xc : -1;
        }
        // every global load of the tile is issued before any result is used
        T dv[RPW * NCH], zv[FLY ? RPW * NCH : 1];
        uint64_t lw[RPW * NCH];
        unsigned rowmask = 0;
#pragma unroll
        for (int ri = 0; ri < RPW; ++ri) {
            const int r = wv + ri * NWV;
            const int ys = y0 - H + r;
            int Yr;
            bool rowok = r < HT;
            if (g.bnd == BND_HALO) { Yr = ys + g.h; rowok = rowok && (Yr >= 0 && Yr < g.nyh); }
            else Yr = ys < 0 ? 0 : (ys >= g.ny ? g.ny - 1 : ys);
            rowmask |= (rowok ? 1u : 0u) << ri;
            const unsigned rowbase = (unsigned)Yr * unxh, wordbase = (unsigned)Yr * (unsigned)g.nw;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int k = ri * NCH + ch;
                dv[k] = T(0);
                lw[k] = 0;
                if constexpr (FLY) zv[k] = T(0);
                if (rowok && xcol[ch] >= 0) {
                    const unsigned ii = rowbase + (unsigned)xcol[ch];
                    if constexpr (FLY) { dv[k] = job.theta[ii]; zv[k] = job.t0[ii]; }
                    else dv[k] = job.t0[ii];
                    lw[k] = job.clsbits[wordbase + ((unsigned)xcol[ch] >> 6)];
                }
            }
        }
        if constexpr (FLY) c_t0 = c_th - c_gz;
        const double c0 = (double)c_t0;
        unsigned mine = 0;
#pragma unroll
        for (int q = 0; q < CPT; ++q) mine |= (unsigned)((bw[q] >> ((x0 + lx + g.h) & 63)) & 1ull) << q;
        // ---- into LDS; the land-side count is prefixed along the row on the way in with a
        // ballot + popcount, so only the two fp64 tables need a longitude scan ----------------
#pragma unroll
        for (int ri = 0; ri < RPW; ++ri) {
            const int r = wv + ri * NWV;
            const bool rowok = (rowmask >> ri) & 1u;
            unsigned carryC = 0;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int k = ri * NCH + ch;
                const int c = ch * SB_WAVE + lane;
                const bool ok = rowok && xcol[ch] >= 0;
                const int land = ok ? (int)((lw[k] >> (xcol[ch] & 63)) & 1ull) : 0;
                T t0v = dv[k];
                if constexpr (FLY) t0v = dv[k] - zv[k];
                const double d = ok ? (double)t0v - c0 : 0.0;
                const uint64_t lm = __ballot(land);
                const unsigned cn = carryC + (unsigned)__popcll(lm & (~0ull >> (63 - lane)));
                carryC += (unsigned)__popcll(lm);
                if (r < HT && c < W) {
                    const int o = (r + 1) * P + c + 1;
                    sA[o] = d;
                    sL[o] = land ? d : 0.0;
                    sC[o] = (unsigned short)cn;
                }
            }
        }
        __syncthreads();
        SB_STAMP(1);
        // ---- prefix along longitude: one task per (fp64 table, row), batches of 16 ----------
        for (int task = tid; task < 2 * HT; task += NT) {
            const int a = task / HT, r = task - a * HT + 1;
            double *row = (a == 0 ? sA : sL) + r * P;
            double s = 0.0;
            for (int cb = 1; cb <= W; cb += 16) {
                double v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = row[cb + i];
#pragma unroll
                for (int i = 0; i < 16; ++i) { s += v[i]; v[i] = s; }
#pragma unroll
                for (int i = 0; i < 16; ++i) row[cb + i] = v[i];
            }
        }
        __syncthreads();
        SB_STAMP(2);
        // the state / wind loads of the final update: issued now, they land under the scan
        SbCellState<T> cst[CPT];
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            cst[q] = SbCellState<T>{T(0), T(0), T(0), T(0)};
            if (FUSE && ((mine >> q) & 1u))
                cst[q] = sb_trigger_load<T>(job, (size_t)(y0 + ly0 + q * (NT / TX)) * g.nx + (x0 + lx));
        }
        // ---- prefix along latitude: one task per (table, column), batches of 16 rows ------
        for (int task = tid; task < 3 * W; task += NT) {
            const int a = task / W, c = task - a * W + 1;
            if (a < 2) {
                double *tab = (a == 0 ? sA : sL) + c;
                double s = 0.0;
                for (int rb = 1; rb <= HT; rb += 16) {
                    double v[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = tab[(rb + i) * P];
#pragma unroll
                    for (int i = 0; i < 16; ++i) { s += v[i]; v[i] = s; }
#pragma unroll
                    for (int i = 0; i < 16; ++i) tab[(rb + i) * P] = v[i];
                }
            } else {
                unsigned short *tab = sC + c;
                unsigned s = 0;
                for (int rb = 1; rb <= HT; rb += 16) {
                    unsigned v[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = tab[(rb + i) * P];
#pragma unroll
                    for (int i = 0; i < 16; ++i) { s += v[i]; v[i] = s; }
#pragma unroll
                    for (int i = 0; i < 16; ++i) tab[(rb + i) * P] = (unsigned short)v[i];
                }
            }
        }
        __syncthreads();
        SB_STAMP(3);

        // ---- smallest radius whose square holds both classes: bisection, O(1) per probe ---
        int cxq[CPT], cyq[CPT], lo[CPT], hi[CPT], nlq[CPT];
        bool fnd[CPT];
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const int ly = ly0 + q * (NT / TX);
            const int x = x0 + lx, y = y0 + ly;
            cxq[q] = lx + H;
            cyq[q] = ly + H;
            int lim = H;
            if (g.bnd == BND_HALO)
                lim = min(lim, min(min(x + g.h, g.nx - 1 - x + g.h), min(y + g.h, g.ny - 1 - y + g.h)));
            lo[q] = 1;
            hi[q] = lim;
            fnd[q] = false;
            nlq[q] = 0;
            if ((mine >> q) & 1u) {
                if (lim >= 1) {
                    const int r0 = (cyq[q] - lim) * P, r1 = (cyq[q] + lim + 1) * P;
                    const int a0 = cxq[q] - lim, a1 = cxq[q] + lim + 1;
                    const int nl = (int)sC[r1 + a1] - (int)sC[r0 + a1] - (int)sC[r1 + a0] + (int)sC[r0 + a0];
                    fnd[q] = nl > 0 && nl < (2 * lim + 1) * (2 * lim + 1);
                    nlq[q] = nl;
                }
            }
            if (!fnd[q]) lo[q] = hi[q];          // nothing to bisect
        }
        constexpr int ITER = (H <= 2 ? 1 : H <= 4 ? 2 : H <= 8 ? 3 : H <= 16 ? 4 : 5);
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                if (lo[q] < hi[q]) {
                    const int mid = (lo[q] + hi[q]) >> 1;
                    const int r0 = (cyq[q] - mid) * P, r1 = (cyq[q] + mid + 1) * P;
                    const int a0 = cxq[q] - mid, a1 = cxq[q] + mid + 1;
                    const int nl = (int)sC[r1 + a1] - (int)sC[r0 + a1] - (int)sC[r1 + a0] + (int)sC[r0 + a0];
                    if (nl > 0 && nl < (2 * mid + 1) * (2 * mid + 1)) { hi[q] = mid; nlq[q] = nl; }
                    else lo[q] = mid + 1;
                }
            }
        }
        int nnmax = 0;
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            if (!((mine >> q) & 1u)) continue;
            const int ly = ly0 + q * (NT / TX);
            const int x = x0 + lx, y = y0 + ly;
            int nn = hi[q];
            T contrast;
            if (fnd[q]) {
                const int r0 = (cyq[q] - nn) * P, r1 = (cyq[q] + nn + 1) * P;
                const int a0 = cxq[q] - nn, a1 = cxq[q] + nn + 1;
                const int area = (2 * nn + 1) * (2 * nn + 1);
                const double RL = (sL[r1 + a1] - sL[r0 + a1]) - (sL[r1 + a0] - sL[r0 + a0]);
                const double RA = (sA[r1 + a1] - sA[r0 + a1]) - (sA[r1 + a0] - sA[r0 + a0]);
                contrast = (T)(RL / (double)nlq[q] - (RA - RL) / (double)(area - nlq[q]));
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
            const T mul = ((cw[q] >> ((x + g.h) & 63)) & 1ull) ? T(1) : T(-1);        // ref :182-186
            if constexpr (FUSE) sb_trigger_update<T>(job, (size_t)y * g.nx + x, mul * contrast, cst[q]);  // ref :216, :235-266
            else job.thc[(size_t)y * g.nx + x] = mul * contrast;       // k_final_tiles applies :235-266
        }
        // per-tile largest radius (diagnostic; reduced lazily by sb_last_counters)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nnmax = max(nnmax, __shfl_xor(nnmax, off));
        if (lane == 0 && nnmax > 1) atomicMax(&s_nn, nnmax);
        __syncthreads();                         // also fences the tables before the next tile
        if (tid == 0) job.tile_nnmax[tile] = s_nn;
        SB_STAMP(5);
    }
}

template <typename T, int TY, int H>
static void launch_thc(const DiagJob<T> &job, int nblocks, bool fuse, hipStream_t st) {
    if (job.t0_fly && fuse) hipLaunchKernelGGL((k_thc<T, TY, H, true, true>), dim3(nblocks), dim3(THC_NT), 0, st, job);
    else if (job.t0_fly) hipLaunchKernelGGL((k_thc<T, TY, H, true, false>), dim3(nblocks), dim3(THC_NT), 0, st, job);
    else hipLaunchKernelGGL((k_thc<T, TY, H, false, true>), dim3(nblocks), dim3(THC_NT), 0, st, job);
}

int sb_thc_tile_rows(int H) { return H <= 16 ? 32 : 16; }

template <typename T>
hipError_t sb_launch_thc(const DiagJob<T> &job, int H, int ncu, bool fuse, hipStream_t st) {
    // one workgroup per CU; more only if a workgroup could own more tiles than its list holds
    const int ntiles = job.thc_ntx * job.thc_nty;
    int nblocks = ncu;
    while ((ntiles + nblocks - 1) / nblocks + 8 > THC_MAXMINE) nblocks *= 2;
    if (H <= 8) launch_thc<T, 32, 8>(job, nblocks, fuse, st);
    else if (H <= 16) launch_thc<T, 32, 16>(job, nblocks, fuse, st);
    else launch_thc<T, 16, 24>(job, nblocks, fuse, st);
    return hipGetLastError();
}

template <typename T>
hipError_t sb_launch_final_tiles(const DiagJob<T> &job, int ncu, hipStream_t st) {
    const int ntiles = job.thc_ntx * job.thc_nty;
    int nblocks = ncu;
    while ((ntiles + nblocks - 1) / nblocks + 8 > THC_MAXMINE) nblocks *= 2;
    hipLaunchKernelGGL(k_final_tiles<T>, dim3(nblocks), dim3(THC_NT), 0, st, job);
    return hipGetLastError();
}
template hipError_t sb_launch_final_tiles<float>(const DiagJob<float> &, int, hipStream_t);
template hipError_t sb_launch_final_tiles<double>(const DiagJob<double> &, int, hipStream_t);

template hipError_t sb_launch_thc<float>(const DiagJob<float> &, int, int, bool, hipStream_t);
template hipError_t sb_launch_thc<double>(const DiagJob<double> &, int, int, bool, hipStream_t);
