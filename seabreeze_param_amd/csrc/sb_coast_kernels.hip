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
__device__ __forceinline__ int land_rule(const T *lsm, const T *ci, size_t i, int rule) {
    const T l = lsm[i], c = ci[i];
    if (rule == 0) return (l + c > T(0.4)) ? 1 : 0;            // ref: sobel.f90:51,69
    if (c <= T(0.2)) return (l >= T(0.5)) ? 1 : 0;             // ref: generic :325-330
    return (l + c >= T(0.5)) ? 1 : 0;                          // ref: generic :332-336
}

template <typename T>
__global__ __launch_bounds__(256) void k_edges(const T *__restrict__ lsm, const T *__restrict__ ci,
                                               T *__restrict__ coast, Geo g, int rule) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= g.nx) return;
    int m[3][3];   // [lat offset + 1][lon offset + 1]
#pragma unroll
    for (int a = -1; a <= 1; ++a)
#pragma unroll
        for (int b = -1; b <= 1; ++b) {
            int X, Y;
            sb_map_cell(g, x + b, y + a, X, Y);
            m[a + 1][b + 1] = land_rule(lsm, ci, (size_t)Y * g.nx + X, rule);
        }
    // weight = reshape((/-1,-2,-1, 0,0,0, 1,2,1/),(3,3)) column-major: w(r,c) = (1,2,1)(r) * (-1,0,1)(c)
    // px += w(a+2, b+2)*m, py += w(b+2, a+2)*m     ref: sobel.f90:74-75
    const int w3[3] = {1, 2, 1};
    int px = 0, py = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            px += w3[a] * (b - 1) * m[a][b];
            py += w3[b] * (a - 1) * m[a][b];
        }
    coast[(size_t)y * g.nx + x] = (px == 0 && py == 0) ? T(0) : T(1);   // sqrt(px^2+py^2) == 0
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
// cost a few hundred instructions instead of (2k+1)^2 byte probes.  Same arithmetic per
// coast hit as k_dist, same early/late bookkeeping, same results.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_coastbits(const T *__restrict__ coast, uint64_t *__restrict__ bits,
                                                   int nx, int ny, int nw) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    const bool c = x < nx && coast[(size_t)y * nx + x] > T(0);       // ref: sobel.f90:157
    const uint64_t w = __ballot(c);
    if ((threadIdx.x & 63) == 0 && (x >> 6) < nw) bits[(size_t)y * nw + (x >> 6)] = w;
}

// bits p .. p+l-1 (l <= 63, p+l <= nx) of a row of the plane
__device__ __forceinline__ uint64_t row_bits(const uint64_t *__restrict__ rw, int p, int l) {
    const int w = p >> 6, o = p & 63;
    uint64_t v = rw[w] >> o;
    if (o + l > 64) v |= rw[w + 1] << (64 - o);
    return v & ((1ull << l) - 1ull);
}

template <typename T>
__global__ __launch_bounds__(256) void k_dist_bits(const uint64_t *__restrict__ bits, const T *__restrict__ mask,
                                                   const T *__restrict__ phi, const T *__restrict__ lamf,
                                                   T *__restrict__ cdist, int nx, int ny, int nw, int k,
                                                   T maxdist) {
    const int xx = blockIdx.x * 256 + threadIdx.x, yy = blockIdx.y;
    if (xx >= nx) return;
    const T R = T(6370.9989);                                   // ref: sobel.f90:115
    const T big = T(12000.);
    const T phit = phi[yy], lamt = lamf[xx];
    const T cost = cos(phit);
    const int L = 2 * k + 1;
    int start = (xx - k) % nx;                                   // first window column, circular
    if (start < 0) start += nx;
    const int len1 = L < nx - start ? L : nx - start;
    T m_early = big, m_late = big;
    for (int ii = -k; ii <= k; ++ii) {
        const int ys = yy + ii;
        if (ys < 0 || ys >= ny) continue;                        // clamped rows add no new sources
        const uint64_t *rw = bits + (size_t)ys * nw;
        uint64_t wb = row_bits(rw, start, len1);
        if (len1 < L) wb |= row_bits(rw, 0, L - len1) << len1;
        if (!wb) continue;
        const T phis = phi[ys];
        const T dphi = phis - phit;                              // phi1(i) - phi1(yy)
        const T sp = sin(dphi / T(2));
        const T cosp = cos(phis);
        while (wb) {
            const int b = __builtin_ctzll(wb);
            wb &= wb - 1;
            int xs = start + b;
            if (xs >= nx) xs -= nx;
            const T dlam = lamf[xs] - lamt;                      // l1 - l2
            const T sl = sin(dlam / T(2));
            const T a = sp * sp + (cosp * (cost * (sl * sl)));   // ref: sobel.f90:176
            const T c = (R * T(2)) * atan2(sqrt(a), sqrt(T(1) - a)) + T(0.5);   // ref :177
            const bool early = (ys < yy) || (ys == yy && xs <= xx);
            if (early) m_early = c < m_early ? c : m_early;
            else m_late = c < m_late ? c : m_late;
        }
    }
    if (m_early > T(2) * maxdist) m_early = big;                 // ref: sobel.f90:188 at sweep time
    const T m = m_early < m_late ? m_early : m_late;
    const size_t o = (size_t)yy * nx + xx;
    if (m >= big) cdist[o] = big;
    else cdist[o] = (mask[o] > T(0)) ? m : -m;                   // ref :179-183
}

template <typename T>
hipError_t sb_launch_edges(const T *lsm, const T *ci, T *coast, int nx, int ny, int rule, int bnd, hipStream_t st) {
    Geo g;
    g.nx = nx; g.ny = ny; g.h = 0; g.nxh = nx; g.nyh = ny; g.nw = (nx + 63) / 64; g.bnd = bnd; g.rows = ny;
    hipLaunchKernelGGL(k_edges<T>, dim3((nx + 255) / 256, ny), dim3(256), 0, st, lsm, ci, coast, g, rule);
    return hipGetLastError();
}

template <typename T>
hipError_t sb_launch_dist(const T *coast, const T *mask, const T *phi, const T *lamf, T *cdist, int nx, int ny,
                          int k, T maxdist, uint64_t *bits, hipStream_t st) {
    if (bits && k <= 31 && 2 * k + 1 <= nx) {
        const int nw = (nx + 63) / 64;
        hipLaunchKernelGGL(k_coastbits<T>, dim3((nx + 255) / 256, ny), dim3(256), 0, st, coast, bits, nx, ny, nw);
        hipLaunchKernelGGL(k_dist_bits<T>, dim3((nx + 255) / 256, ny), dim3(256), 0, st, bits, mask, phi, lamf, cdist,
                           nx, ny, nw, k, maxdist);
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
template hipError_t sb_launch_dist<float>(const float *, const float *, const float *, const float *, float *, int,
                                          int, int, float, uint64_t *, hipStream_t);
template hipError_t sb_launch_dist<double>(const double *, const double *, const double *, const double *, double *,
                                           int, int, int, double, uint64_t *, hipStream_t);
