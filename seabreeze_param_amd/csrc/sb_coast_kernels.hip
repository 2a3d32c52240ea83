// sb_coast_kernels.hip -- coastline detection and signed coast distance on gfx950.
//
//   k_edges  binary 3x3 Sobel of the land mask          ref: sobel.f90:19-89,
//                                                        generic/sea_breeze_diag.f90:273-373
//   k_dist   signed haversine distance to the nearest coast cell, as a GATHER: each
//            target cell scans its own window for coast cells.  The reference is a
//            racy scatter over coast cells (ref: sobel.f90:152-191);
//            the gather keeps its results, including the sweep-order reset of :188,
//            by tracking the minimum over sources swept before and after the target.
#include "sb_device.hpp"
#include "sb_launch.hpp"

template <typename T>
__device__ __forceinline__ int land_rule(T l, T c, int rule) {
    if (rule == 0) return (l + c > T(0.4)) ? 1 : 0;            // ref: sobel.f90:51,69
    if (c <= T(0.2)) return (l >= T(0.5)) ? 1 : 0;             // ref: generic :325-330
    return (l + c >= T(0.5)) ? 1 : 0;                          // ref: generic :332-336
}

// A workgroup classifies the cells of a 256-column x EDGE_ROWS-row block and of the ring round it once (18 rows read
// for 16 written: 1.125 x the compulsory reads; with 8-row blocks the PMC passes showed 1.47 x)
// (two loads per cell instead of eighteen), keeps the land flags in LDS and takes the nine-point sums
// from there.  The boundary mapping (cyclic longitudes with the f2py flavour's column quirk, clamped
// latitudes; ref: sobel.f90:60-66, generic :340-352) is applied to the staged cell, so a flag means
// exactly what the reference's inner loop would have read at that offset.
#define EDGE_ROWS 16
#define EDGE_PITCH 264

template <typename T>
__global__ __launch_bounds__(256) void k_edges(const T *__restrict__ lsm, const T *__restrict__ ci,
                                               T *__restrict__ coast, Geo g, int rule) {
    __shared__ unsigned char s_land[(EDGE_ROWS + 2) * EDGE_PITCH];
    const int x0 = blockIdx.x * 256, y0 = blockIdx.y * EDGE_ROWS;
    constexpr int NCELL = (EDGE_ROWS + 2) * 258, NIT = (NCELL + 255) / 256;
    T l[NIT], c[NIT];
#pragma unroll
    for (int j = 0; j < NIT; ++j) {                      // every load issued before the first is used
        const int i = threadIdx.x + 256 * j, r = i / 258, cc = i - r * 258;
        int X, Y;
        sb_map_cell(g, x0 - 1 + cc, y0 - 1 + (r < EDGE_ROWS + 2 ? r : 0), X, Y);
        const size_t o = (size_t)Y * g.nx + X;
        l[j] = lsm[o];
        c[j] = ci[o];
    }
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
        const int i = threadIdx.x + 256 * j, r = i / 258, cc = i - r * 258;
        const int land = land_rule(l[j], c[j], rule);
        if (i < NCELL) s_land[r * EDGE_PITCH + cc] = (unsigned char)land;
    }
    __syncthreads();
    const int x = x0 + threadIdx.x;
    if (x >= g.nx) return;
    // weight = reshape((/-1,-2,-1, 0,0,0, 1,2,1/),(3,3)) column-major: w(r,c) = (1,2,1)(r) * (-1,0,1)(c)
    // px += w(a+2, b+2)*m, py += w(b+2, a+2)*m     ref: sobel.f90:74-75
    // -> px = sum_a (1,2,1)(a) * (m[a][2] - m[a][0]),  py = sum_b (1,2,1)(b) * (m[2][b] - m[0][b])
    int m[3][3];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) m[a + 1][b] = s_land[a * EDGE_PITCH + threadIdx.x + b];
#pragma unroll
    for (int r = 0; r < EDGE_ROWS; ++r) {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            m[0][b] = m[1][b];
            m[1][b] = m[2][b];
            m[2][b] = s_land[(r + 2) * EDGE_PITCH + threadIdx.x + b];
        }
        const int px = (m[0][2] - m[0][0]) + 2 * (m[1][2] - m[1][0]) + (m[2][2] - m[2][0]);
        const int py = (m[2][0] - m[0][0]) + 2 * (m[2][1] - m[0][1]) + (m[2][2] - m[0][2]);
        const int y = y0 + r;
        if (y < g.ny) coast[(size_t)y * g.nx + x] = (px == 0 && py == 0) ? T(0) : T(1);   // sqrt(px^2+py^2) == 0
    }
}


template <typename T>
__global__ __launch_bounds__(256) void k_dist(const T *__restrict__ coast, const T *__restrict__ mask,
                                              const T *__restrict__ phi, const T *__restrict__ lamf,
                                              T *__restrict__ cdist, int nx, int ny, int k, T maxdist) {
    extern __shared__ unsigned char sflag[];   // (64+2k) x (SB_DIST_TY+2k) coast flags
    const int W = 64 + 2 * k, HT = SB_DIST_TY + 2 * k;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * SB_DIST_TY;
    for (int i = threadIdx.x; i < W * HT; i += 256) {
        const int r = i / W, c = i - r * W;
        const int ys = y0 - k + r;
        int xs = (x0 - k + c) % nx;
        if (xs < 0) xs += nx;
        unsigned char f = 0;
        if (ys >= 0 && ys < ny) f = coast[(size_t)ys * nx + xs] > T(0) ? 1 : 0;
        sflag[i] = f;
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int xx = x0 + lx, yy = y0 + ly;
    if (xx >= nx || yy >= ny) return;

    const T R = T(6370.9989);                                   // ref: sobel.f90:115
    const T big = T(12000.);
    const T phit = phi[yy], lamt = lamf[xx];
    const T cost = cos(phit);
    T m_early = big, m_late = big;
    for (int ii = -k; ii <= k; ++ii) {
        const int ys = yy + ii;
        if (ys < 0 || ys >= ny) continue;      // clamped rows add no new sources
        const unsigned char *row = &sflag[(ly + k + ii) * W + lx];   // column offset c = lx + k - jj
        for (int jj = -k; jj <= k; ++jj) {
            if (!row[k - jj]) continue;
            int xs = (xx - jj) % nx;
            if (xs < 0) xs += nx;
            const T phis = phi[ys];
            const T dphi = phis - phit;                          // phi1(i) - phi1(yy)
            const T dlam = lamf[xs] - lamt;                      // l1 - l2
            const T sp = sin(dphi / T(2)), sl = sin(dlam / T(2));
            const T a = sp * sp + (cos(phis) * (cost * (sl * sl)));   // ref: sobel.f90:176
            const T c = (R * T(2)) * atan2(sqrt(a), sqrt(T(1) - a)) + T(0.5);   // ref :177
            const bool early = (ys < yy) || (ys == yy && xs <= xx);
            if (early) m_early = c < m_early ? c : m_early;
            else m_late = c < m_late ? c : m_late;
        }
    }
    // the reference resets cdist(j,i) to 12000 when its own sweep position is reached and
    // the value so far exceeds 2*maxdist (ref: sobel.f90:188); later sources may still lower it
    if (m_early > T(2) * maxdist) m_early = big;
    const T m = m_early < m_late ? m_early : m_late;
    const size_t o = (size_t)yy * nx + xx;
    if (m >= big) cdist[o] = big;
    else cdist[o] = (mask[o] > T(0)) ? m : -m;                   // ref :179-183
}

// ------------------------------------------------------------------------------------
// Bit-plane form of k_dist for windows of at most 63 columns (k <= 31, every BASELINE
// grid): k_coastbits packs coast > 0 into one 64-bit word per 64-cell longitude segment
// (a wave ballot); k_dist_bits pulls the (2k+1)-column window of each source row out of at
// most four words and visits set bits only.  Nine targets in ten have an empty window and
// cost a few hundred instructions instead of (2k+1)^2 byte probes.
//
// Two further cuts, both exact up to the last place of atan2:
//   * the distance c = 2R atan2(sqrt(a), sqrt(1-a)) + 0.5 (ref: sobel.f90:176-177) grows with a,
//     so the minimum of c over a class of sources is c at the minimum of a: the kernel keeps
//     min(a) of the sources swept before and after the target and takes two atan2 per cell
//     instead of one per coast hit;
//   * within one source row a = sp^2 + cos(phis) cos(phit) sin^2(dlam/2) grows with the
//     longitude distance, so of the hits left of (or at) the target column only the nearest can
//     be the minimum, and likewise on the right: two hits per row instead of up to 2k+1.  On the
//     target's own row the sweep order splits each side once more where the window crosses the
//     seam (xs <= xx is decided on wrapped indices, ref: sobel.f90:188), so up to four there.
//   * a >= sp^2, which grows with the row distance: the walk away from the target row stops early.
//     `nearest` (both row cuts) is set by the host only when the longitudes step one way round the circle,
//     the window spans less than half of it and the latitudes step one way (sb_capi.hip); otherwise every
//     hit of every row is visited.
// Same arithmetic per visited hit as k_dist, same early/late bookkeeping.
// ------------------------------------------------------------------------------------
#define COASTBITS_ROWS 8              // rows per workgroup of k_coastbits: eight loads in flight per thread
template <typename T>
__global__ __launch_bounds__(256) void k_coastbits(const T *__restrict__ coast, uint64_t *__restrict__ bits,
                                                   int nx, int ny, int nw) {
    const int x = blockIdx.x * 256 + threadIdx.x, y0 = blockIdx.y * COASTBITS_ROWS;
    T v[COASTBITS_ROWS];
#pragma unroll
    for (int r = 0; r < COASTBITS_ROWS; ++r) {
        const int y = y0 + r < ny ? y0 + r : ny - 1;                 // (clamped: every load unconditional)
        v[r] = coast[(size_t)y * nx + (x < nx ? x : nx - 1)];
    }
#pragma unroll
    for (int r = 0; r < COASTBITS_ROWS; ++r) {
        const uint64_t w = __ballot(x < nx && v[r] > T(0));          // ref: sobel.f90:157
        if ((threadIdx.x & 63) == 0 && (x >> 6) < nw && y0 + r < ny) bits[(size_t)(y0 + r) * nw + (x >> 6)] = w;
    }
}

// bits p .. p+l-1 (l <= 63, p+l <= nx) of a row of the plane
__device__ __forceinline__ uint64_t row_bits(const uint64_t *__restrict__ rw, int p, int l) {
    const int w = p >> 6, o = p & 63;
    uint64_t v = rw[w] >> o;
    if (o + l > 64) v |= rw[w + 1] << (64 - o);
    return v & ((1ull << l) - 1ull);
}

__device__ __forceinline__ uint64_t top_bit(uint64_t x) { return x ? 1ull << (63 - __builtin_clzll(x)) : 0ull; }
__device__ __forceinline__ uint64_t low_bit(uint64_t x) { return x & (0ull - x); }

// k_dist_bits as rounds 1-3 had it (every word and table entry from global memory, per target and row): kept for grids
// narrower than 2k + 1 + 256 columns, where the staged reach of a workgroup would wrap round the seam more than once
template <typename T>
__global__ __launch_bounds__(256) void k_dist_bits_small(const uint64_t *__restrict__ bits, const T *__restrict__ mask,
                                                   const T *__restrict__ phi, const T *__restrict__ lamf,
                                                   const T *__restrict__ shl, const T *__restrict__ chl,
                                                   T *__restrict__ cdist, int nx, int ny, int nw, int k,
                                                   T maxdist, int nearest) {
    __shared__ T s_sp2[64], s_cosp[64], s_cost;
    const int xx = blockIdx.x * 256 + threadIdx.x, yy = blockIdx.y;
    const T big = T(12000.);
    // Which of the 2k+1 source rows hold any coast cell within reach of this wave's 64 targets?  Lane i ORs the
    // (at most three) words of row yy - k + i that cover columns x0 - k .. x0 + 63 + k; the ballot is a row mask in
    // scalar registers, and rows without a bit are skipped by the whole wave.  Four targets in five see no coast
    // at all and end here after one load round.  A wave whose reach crosses the seam keeps every row.
    uint64_t rowmask;
    {
        const int lane = threadIdx.x & 63, wx0 = blockIdx.x * 256 + ((int)threadIdx.x & ~63);
        const bool seam = wx0 - k < 0 || wx0 + 63 + k >= nx;
        const int ys = yy + lane - k;
        uint64_t any = 0;
        if (lane <= 2 * k && ys >= 0 && ys < ny) {
            if (seam) any = 1;
            else
                for (int w = (wx0 - k) >> 6; w <= (wx0 + 63 + k) >> 6; ++w) any |= bits[(size_t)ys * nw + w];
        }
        rowmask = __ballot(any != 0);
    }
    // Four workgroups in five have no coast cell within reach of any of their 256 targets: they write the "unreached"
    // value and leave before any trigonometry (round 2 formed the latitude factors of the 2k+1 source rows and
    // cos(phi) of the target row -- the same for the whole workgroup -- in every workgroup and every thread).
    if (!__syncthreads_or(rowmask != 0 ? 1 : 0)) {
        if (xx < nx) cdist[(size_t)yy * nx + xx] = big;
        return;
    }
    // the latitude factors of the 2k+1 source rows are the same for every target of this row: once per workgroup
    const T phit = phi[yy];
    if ((int)threadIdx.x <= 2 * k) {
        int ys = yy + (int)threadIdx.x - k;
        ys = ys < 0 ? 0 : (ys >= ny ? ny - 1 : ys);
        const T phis = phi[ys];
        const T dphi = phis - phit;                              // phi1(i) - phi1(yy)
        const T sp = sin(dphi / T(2));
        s_sp2[threadIdx.x] = sp * sp;
        s_cosp[threadIdx.x] = cos(phis);
    } else if ((int)threadIdx.x == 2 * k + 1) s_cost = cos(phit);
    __syncthreads();
    if (xx >= nx) return;
    if (rowmask == 0) {                                          // wave-uniform: none of this wave's targets is reached
        cdist[(size_t)yy * nx + xx] = big;
        return;
    }
    const T R = T(6370.9989);                                   // ref: sobel.f90:115
    const T lamt = lamf[xx];
    const T cost = s_cost;
    // fp64: sin((l1 - l2) / 2) = sin(l1/2) cos(l2/2) - cos(l1/2) sin(l2/2) from per-column tables (the host forms
    // them with the folded longitudes): two loads, two products and a difference per hit where the library sine took some eighty
    // instructions.  The difference of products carries an absolute error of 1e-16, i.e. <= 4e-13 relative in the
    // distance at the finest spacing in use (0.07 degrees); the tests hold 1e-12.  fp32 keeps the sine: there the same
    // identity would cost four digits.
    T sht = T(0), cht = T(0);
    if constexpr (sizeof(T) == 8) { sht = shl[xx]; cht = chl[xx]; }
    const int L = 2 * k + 1;
    int start = (xx - k) % nx;                                   // first window column, circular
    if (start < 0) start += nx;
    const int len1 = L < nx - start ? L : nx - start;
    // window bit b is column xx - k + b: bits 0 .. k lie left of or at the target, the rest right of it;
    // bits below k - xx and from nx - xx + k on have wrapped round the seam
    const uint64_t left = (2ull << k) - 1ull;
    const uint64_t wrapl = (k - xx > 0) ? (1ull << (k - xx)) - 1ull : 0ull;
    const uint64_t wrapr = (nx - xx + k < 64) ? ~0ull << (nx - xx + k) : 0ull;
    const T none = T(4);                                         // a <= 1: "no source in this class"
    T a_early = none, a_late = none;
    auto row = [&](int ii) {
        const int ys = yy + ii;
        const uint64_t *rw = bits + (size_t)ys * nw;
        uint64_t wb = row_bits(rw, start, len1);
        if (len1 < L) wb |= row_bits(rw, 0, L - len1) << len1;
        if (!wb) return;
        if (nearest) {
            const uint64_t wl = wb & left, wr = wb & ~left;
            if (ii != 0) wb = top_bit(wl) | low_bit(wr);
            else wb = top_bit(wl & ~wrapl) | top_bit(wl & wrapl) | low_bit(wr & ~wrapr) | low_bit(wr & wrapr);
        }
        const T sp2 = s_sp2[ii + k], cosp = s_cosp[ii + k];
        while (wb) {
            const int b = __builtin_ctzll(wb);
            wb &= wb - 1;
            int xs = start + b;
            if (xs >= nx) xs -= nx;
            T sl;
            // (two rounded products, no fma: for the target's own column they are the same product and the difference
            // is exactly zero -- a coast cell's own distance stays exactly 0.5 km, SURVEY.md section 4's anchor)
            if constexpr (sizeof(T) == 8) sl = shl[xs] * cht - chl[xs] * sht;
            else {
                const T dlam = lamf[xs] - lamt;                  // l1 - l2
                sl = sin(dlam / T(2));
            }
            const T a = sp2 + (cosp * (cost * (sl * sl)));       // ref: sobel.f90:176
            const bool early = (ys < yy) || (ys == yy && xs <= xx);
            if (early) a_early = a < a_early ? a : a_early;
            else a_late = a < a_late ? a : a_late;
        }
    };
    // Rows above the target are swept before it, rows below after it.  a >= sp2 of its row, and with latitudes
    // that step one way (the host's condition for `nearest`) sp2 grows with the row distance: a walk away
    // from the target can stop at the first row whose sp2 is no smaller than the class's minimum so far.
    if ((rowmask >> k) & 1) row(0);
    for (int d = 1; d <= k && yy - d >= 0; ++d) {                // clamped rows add no new sources
        if (!((rowmask >> (k - d)) & 1)) continue;               // wave-uniform
        if (nearest && !(s_sp2[k - d] < a_early)) break;
        row(-d);
    }
    for (int d = 1; d <= k && yy + d < ny; ++d) {
        if (!((rowmask >> (k + d)) & 1)) continue;
        if (nearest && !(s_sp2[k + d] < a_late)) break;
        row(d);
    }
    T m_early = big, m_late = big;
    if (a_early < none) m_early = (R * T(2)) * atan2(sqrt(a_early), sqrt(T(1) - a_early)) + T(0.5);   // ref :177
    if (a_late < none) m_late = (R * T(2)) * atan2(sqrt(a_late), sqrt(T(1) - a_late)) + T(0.5);
    if (m_early > T(2) * maxdist) m_early = big;                 // ref: sobel.f90:188 at sweep time
    const T m = m_early < m_late ? m_early : m_late;
    const size_t o = (size_t)yy * nx + xx;
    if (m >= big) cdist[o] = big;
    else cdist[o] = (mask[o] > T(0)) ? m : -m;                   // ref :179-183
}

// 64 bits of a row of the plane from circular column p on (0 <= p < nx, nx >= 64): across the seam where need be
__device__ __forceinline__ uint64_t row_bits64(const uint64_t *__restrict__ rw, int p, int nx) {
    if (p + 64 <= nx) {
        const int w = p >> 6, o = p & 63;
        uint64_t v = rw[w] >> o;
        if (o) v |= rw[w + 1] << (64 - o);
        return v;
    }
    const int l1 = nx - p;                                       // 1 .. 63 columns up to the seam, the rest from column 0
    return row_bits(rw, p, l1) | (row_bits(rw, 0, 64 - l1) << l1);
}

#define DIST_SPAN (256 + 2 * 31)                                 // columns a workgroup's 256 targets can reach (k <= 31)
#define DIST_WORDS 5                                             // ... in 64-bit words

#ifndef DIST_ROWS
#define DIST_ROWS 2                                               // target rows per workgroup (256 x 2 threads)
#endif
// WB: the word a target's window (2k+1 columns) and the wave's row mask (2k+1 rows) are held in: 32 bits for k <= 15 (the
// N1280 grid and coarser: the walk is bound by vector-integer issue, and 64-bit shifts and bit scans cost two to three
// 32-bit ones), 64 bits beyond
template <typename T, typename WB>
__global__ __launch_bounds__(256 * DIST_ROWS) void k_dist_bits(const uint64_t *__restrict__ bits, const T *__restrict__ mask,
                                                   const T *__restrict__ phi, const T *__restrict__ lamf,
                                                   const T *__restrict__ shl, const T *__restrict__ chl,
                                                   T *__restrict__ cdist, int nx, int ny, int nw, int k,
                                                   T maxdist, int nearest) {
    __shared__ T s_sp2a[DIST_ROWS][64], s_cospa[DIST_ROWS][64], s_costa[DIST_ROWS];
    // everything the 512 targets of this workgroup read more than once, staged ONCE (round 4: the per-target walk used to
    // fetch two words of the bit plane and two table entries per hit from global memory, row after row -- a chain of
    // dependent L2 round trips, 50 us per wave): the coast bits of the source rows over the columns in reach, as one
    // bit string per row that starts at column x0 - k (circular: the seam is dealt with here, once), and the longitude
    // tables over the same columns
    __shared__ uint64_t s_bw[63 + DIST_ROWS - 1][DIST_WORDS];
    __shared__ T s_ta[DIST_SPAN], s_tb[DIST_SPAN];               // fp64: sin, cos of half the folded longitude; fp32: the folded longitude
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * 256 + tx;
    const int x0 = blockIdx.x * 256, y0 = blockIdx.y * DIST_ROWS;
    const int xx = x0 + tx, yy = y0 + ty;
    const bool rowok = yy < ny;                                  // (an odd number of rows: the last workgroup's second row)
    const T big = T(12000.);
    // Which of the 2k+1 source rows hold any coast cell within reach of this wave's 64 targets?  Lane i ORs the
    // (at most three) words of row yy - k + i that cover columns x0 - k .. x0 + 63 + k; the ballot is a row mask in
    // scalar registers, and rows without a bit are skipped by the whole wave.  Four targets in five see no coast
    // at all and end here after one load round.  A wave whose reach crosses the seam keeps every row.
    uint64_t rowmask;
    {
        const int lane = tx & 63, wx0 = x0 + (tx & ~63);
        const int ys = yy + lane - k;
        uint64_t any = 0;
        if (rowok && lane <= 2 * k && ys >= 0 && ys < ny) {
            // (whole words: an over-estimate of the reach, never an under-estimate; a reach that crosses the seam is two
            // pieces -- round 3 kept every row for such a wave, and one workgroup column in five of the N1280 grid did the
            // whole walk over empty windows)
            const uint64_t *rw = bits + (size_t)ys * nw;
            int lo = wx0 - k, hi = wx0 + 63 + k;
            hi = hi < nx + lo ? hi : nx + lo - 1;                // (at most once round the circle)
            if (lo < 0) {
                for (int w = (lo + nx) >> 6; w < nw; ++w) any |= rw[w];
                lo = 0;
            }
            if (hi >= nx) {
                for (int w = 0; w <= (hi - nx) >> 6; ++w) any |= rw[w];
                hi = nx - 1;
            }
            for (int w = lo >> 6; w <= hi >> 6; ++w) any |= rw[w];
        }
        rowmask = __ballot(any != 0);
    }
    // Four workgroups in five have no coast cell within reach of any of their targets: they write the "unreached"
    // value and leave before any trigonometry.
    if (!__syncthreads_or(rowmask != 0 ? 1 : 0)) {
        if (xx < nx && rowok) cdist[(size_t)yy * nx + xx] = big;
        return;
    }
    const int L = 2 * k + 1;
    int c0 = (x0 - k) % nx;                                      // first column in reach, circular
    if (c0 < 0) c0 += nx;
    // ---- staging ----
    {
        const int yt = rowok ? yy : ny - 1;
        const T phit = phi[yt];
        if (tx <= 2 * k) {
            int ys = yt + tx - k;
            ys = ys < 0 ? 0 : (ys >= ny ? ny - 1 : ys);
            const T phis = phi[ys];
            const T dphi = phis - phit;                          // phi1(i) - phi1(yy)
            const T sp = sin(dphi / T(2));
            s_sp2a[ty][tx] = sp * sp;
            s_cospa[ty][tx] = cos(phis);
        } else if (tx == 2 * k + 1) s_costa[ty] = cos(phit);
    }
    for (int i = tid; i < (L + DIST_ROWS - 1) * DIST_WORDS; i += 256 * DIST_ROWS) {
        const int r = i / DIST_WORDS, j = i - r * DIST_WORDS;
        const int ys = y0 + r - k;
        uint64_t v = 0;
        if (ys >= 0 && ys < ny) {
            int p = c0 + 64 * j;
            p -= p >= nx ? nx : 0;                               // (nx >= 2k + 1 + 256: the launcher's condition for this kernel)
            p -= p >= nx ? nx : 0;
            v = row_bits64(bits + (size_t)ys * nw, p, nx);
        }
        s_bw[r][j] = v;
    }
    for (int i = tid; i < 256 + 2 * k; i += 256 * DIST_ROWS) {
        int xs = c0 + i;
        xs -= xs >= nx ? nx : 0;
        xs -= xs >= nx ? nx : 0;
        if constexpr (sizeof(T) == 8) { s_ta[i] = shl[xs]; s_tb[i] = chl[xs]; }
        else s_ta[i] = lamf[xs];
    }
    __syncthreads();
    if (xx >= nx || !rowok) return;
    if (rowmask == 0) {                                          // wave-uniform: none of this wave's targets is reached
        cdist[(size_t)yy * nx + xx] = big;
        return;
    }
    const T *s_sp2 = s_sp2a[ty], *s_cosp = s_cospa[ty];
    const T s_cost = s_costa[ty];
    const uint64_t (*s_bwt)[DIST_WORDS] = s_bw + ty;             // row ii of this target: s_bwt[ii + k]
    const T R = T(6370.9989);                                   // ref: sobel.f90:115
    const T cost = s_cost;
    // fp64: sin((l1 - l2) / 2) = sin(l1/2) cos(l2/2) - cos(l1/2) sin(l2/2) from the per-column tables: two products and
    // a difference per hit where the library sine took some eighty instructions.  The difference of products carries an
    // absolute error of 1e-16, i.e. <= 4e-13 relative in the distance at the finest spacing in use (0.07 degrees); the
    // tests hold 1e-12.  fp32 keeps the sine: there the same identity would cost four digits.
    const int t = tx;                                            // this target's window starts at bit t of the rows' bit strings
    const T ta_t = s_ta[t + k], tb_t = sizeof(T) == 8 ? s_tb[t + k] : T(0);
    int start = c0 + t;                                          // first window column, circular
    start -= start >= nx ? nx : 0;
    // window bit b is column xx - k + b: bits 0 .. k lie left of or at the target, the rest right of it;
    // bits below k - xx and from nx - xx + k on have wrapped round the seam
    const WB one = 1;
    const WB left = (WB)((WB)(one << k) << 1) - one;
    const WB wrapl = (k - xx > 0) ? (WB)(one << (k - xx)) - one : (WB)0;
    const WB wrapr = (nx - xx + k < (int)(8 * sizeof(WB))) ? (WB)((WB)~(WB)0 << (nx - xx + k)) : (WB)0;
    const WB lmask = (WB)(((uint64_t)1 << L) - 1ull);
    const int tw = t >> 6, to = t & 63;
    auto topb = [](WB x) -> WB {
        if constexpr (sizeof(WB) == 8) return (WB)top_bit((uint64_t)x);
        else return x ? (WB)(1u << (31 - __builtin_clz((unsigned)x))) : (WB)0;
    };
    auto lowb = [](WB x) -> WB { return (WB)(x & ((WB)0 - x)); };
    auto ctzw = [](WB x) -> int {
        if constexpr (sizeof(WB) == 8) return __builtin_ctzll((uint64_t)x);
        else return __builtin_ctz((unsigned)x);
    };
    const T none = T(4);                                         // a <= 1: "no source in this class"
    T a_early = none, a_late = none;
    auto row = [&](int ii) {
        const int ys = yy + ii;
        uint64_t w64 = s_bwt[ii + k][tw] >> to;
        if (to) w64 |= s_bwt[ii + k][tw + 1] << (64 - to);       // (tw + 1 <= 4)
        WB wb = (WB)w64 & lmask;
        if (!wb) return;
        if (nearest) {
            const WB wl = wb & left, wr = wb & (WB)~left;
            if (ii != 0) wb = topb(wl) | lowb(wr);
            else wb = topb(wl & (WB)~wrapl) | topb(wl & wrapl) | lowb(wr & (WB)~wrapr) | lowb(wr & wrapr);
        }
        const T sp2 = s_sp2[ii + k], cosp = s_cosp[ii + k];
        while (wb) {
            const int b = ctzw(wb);
            wb &= (WB)(wb - one);
            int xs = start + b;
            if (xs >= nx) xs -= nx;
            T sl;
            // (two rounded products, no fma: for the target's own column they are the same product and the difference
            // is exactly zero -- a coast cell's own distance stays exactly 0.5 km, SURVEY.md section 4's anchor)
            if constexpr (sizeof(T) == 8) sl = s_ta[t + b] * tb_t - s_tb[t + b] * ta_t;
            else {
                const T dlam = s_ta[t + b] - ta_t;               // l1 - l2
                sl = sin(dlam / T(2));
            }
            const T a = sp2 + (cosp * (cost * (sl * sl)));       // ref: sobel.f90:176
            const bool early = (ys < yy) || (ys == yy && xs <= xx);
            if (early) a_early = a < a_early ? a : a_early;
            else a_late = a < a_late ? a : a_late;
        }
    };
    // Rows above the target are swept before it, rows below after it.  a >= sp2 of its row, and with latitudes
    // that step one way (the host's condition for `nearest`) sp2 grows with the row distance: a walk away
    // from the target can stop at the first row whose sp2 is no smaller than the class's minimum so far.  Only rows
    // whose bit is set in the wave's row mask are visited (scalar bit scans: the mask is wave-uniform).
    if ((rowmask >> k) & 1) row(0);
    for (WB m = (WB)(rowmask & ((1ull << k) - 1ull)); m;) {      // rows above, nearest first
        const int idx = (int)(8 * sizeof(WB)) - 1 - (sizeof(WB) == 8 ? __builtin_clzll((uint64_t)m) : __builtin_clz((unsigned)m));
        m &= (WB)~(WB)(one << idx);
        if (nearest && !(s_sp2[idx] < a_early)) break;
        row(idx - k);
    }
    for (WB m = (WB)(rowmask >> (k + 1)); m;) {                  // rows below, nearest first
        const int d = ctzw(m) + 1;
        m &= (WB)(m - one);
        if (nearest && !(s_sp2[k + d] < a_late)) break;
        row(d);
    }
    // The distance grows with a, so of the two classes' distances only the smaller a's is needed -- one atan2 for the
    // whole wave -- unless it is the early class's and the sweep-time reset (ref: sobel.f90:188) throws it away: then,
    // rarely, the late class's as well.
    auto dist_of = [&](T a) { return (R * T(2)) * atan2(sqrt(a), sqrt(T(1) - a)) + T(0.5); };   // ref :177
    // atan2(sqrt(a), sqrt(1 - a)) = asin(sqrt(a)) = sqrt(a) (1 + a/6 + 3a^2/40 + 15a^3/336 + 105a^4/3456 + 945a^5/42240 + ...):
    // for a < 2^-10 (distances below 400 km: every hit of a maxdist = 180 km window) five terms are exact to 1e-20
    // relative, and the result is within two or three units in the last place of the library's atan2 of the two rounded
    // roots (the tests hold 1e-12) -- for a fifth of the instructions.  a = 0 gives exactly 0.5 km either way.
    auto dist_small = [&](T a) {
        const double ad = (double)a;
        double pl = __builtin_fma(ad, 945.0 / 42240.0, 105.0 / 3456.0);
        pl = __builtin_fma(pl, ad, 15.0 / 336.0);
        pl = __builtin_fma(pl, ad, 3.0 / 40.0);
        pl = __builtin_fma(pl, ad, 1.0 / 6.0);
        pl = __builtin_fma(pl, ad, 1.0);
        return (R * T(2)) * (T)(sqrt(ad) * pl) + T(0.5);
    };
    const bool late_wins = a_late < a_early;
    const T a_sel = late_wins ? a_late : a_early;
    T m = big;
    if (sizeof(T) == 8 && __ballot(a_sel < none && a_sel >= T(0x1p-10)) == 0) {      // wave-uniform
        if (a_sel < none) m = dist_small(a_sel);
    } else if (a_sel < none) m = dist_of(a_sel);
    const bool again = !late_wins && m > T(2) * maxdist;         // ref: sobel.f90:188 at sweep time
    if (__ballot(again) != 0) {                                  // wave-uniform
        const T m2 = a_late < none ? dist_of(a_late) : big;
        if (again) m = m2;
    }
    const size_t o = (size_t)yy * nx + xx;
    if (m >= big) cdist[o] = big;
    else cdist[o] = (mask[o] > T(0)) ? m : -m;                   // ref :179-183
}

template <typename T>
hipError_t sb_launch_edges(const T *lsm, const T *ci, T *coast, int nx, int ny, int rule, int bnd, hipStream_t st) {
    Geo g;
    g.nx = nx; g.ny = ny; g.h = 0; g.nxh = nx; g.nyh = ny; g.nw = (nx + 63) / 64; g.bnd = bnd; g.rows = ny;
    hipLaunchKernelGGL(k_edges<T>, dim3((nx + 255) / 256, (ny + EDGE_ROWS - 1) / EDGE_ROWS), dim3(256), 0, st, lsm, ci, coast, g, rule);
    return hipGetLastError();
}

template <typename T>
hipError_t sb_launch_dist(const T *coast, const T *mask, const T *phi, const T *lamf, const T *shl, const T *chl, T *cdist,
                          int nx, int ny, int k, T maxdist, uint64_t *bits, int nearest, hipStream_t st) {
    if (bits && k <= 31 && 2 * k + 1 <= nx) {
        const int nw = (nx + 63) / 64;
        hipLaunchKernelGGL(k_coastbits<T>, dim3((nx + 255) / 256, (ny + COASTBITS_ROWS - 1) / COASTBITS_ROWS), dim3(256), 0, st, coast, bits, nx, ny, nw);
        if (2 * k + 1 + 256 <= nx)
        {
            const dim3 gr((nx + 255) / 256, (ny + DIST_ROWS - 1) / DIST_ROWS), bl(256, DIST_ROWS);
            if (k <= 15) hipLaunchKernelGGL((k_dist_bits<T, uint32_t>), gr, bl, 0, st, bits, mask, phi, lamf, shl, chl, cdist, nx, ny, nw, k, maxdist, nearest);
            else hipLaunchKernelGGL((k_dist_bits<T, uint64_t>), gr, bl, 0, st, bits, mask, phi, lamf, shl, chl, cdist, nx, ny, nw, k, maxdist, nearest);
        }
        else
            hipLaunchKernelGGL(k_dist_bits_small<T>, dim3((nx + 255) / 256, ny), dim3(256), 0, st, bits, mask, phi, lamf, shl, chl, cdist,
                               nx, ny, nw, k, maxdist, nearest);
        return hipGetLastError();
    }
    const size_t lds = (size_t)(64 + 2 * k) * (SB_DIST_TY + 2 * k);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_dist<T>, dim3((nx + 63) / 64, (ny + SB_DIST_TY - 1) / SB_DIST_TY), dim3(256), lds, st, coast, mask,
                       phi, lamf, cdist, nx, ny, k, maxdist);
    return hipGetLastError();
}

template hipError_t sb_launch_edges<float>(const float *, const float *, float *, int, int, int, int, hipStream_t);
template hipError_t sb_launch_edges<double>(const double *, const double *, double *, int, int, int, int, hipStream_t);
template hipError_t sb_launch_dist<float>(const float *, const float *, const float *, const float *, const float *, const float *,
                                          float *, int, int, int, float, uint64_t *, int, hipStream_t);
template hipError_t sb_launch_dist<double>(const double *, const double *, const double *, const double *, const double *,
                                           const double *, double *, int, int, int, double, uint64_t *, int, hipStream_t);
