// sb_device.hpp -- shared device-side definitions for the sea-breeze HIP kernels (gfx950).
//
// Everything here is written for CDNA4 directly: 64-lane wavefronts (ballots are 64 bit,
// one ballot word == one 64-cell longitude segment of a row), LDS-resident summed-area
// tiles, no CUDA compatibility layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SB_WAVE 64

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

// Grid geometry shared by every kernel of one call.
struct Geo {
    int nx, ny;      // interior cells (lon, lat)
    int h;           // ghost-cell width around the 2-D input fields (0 unless bnd == BND_HALO)
    int nxh, nyh;    // nx + 2h, ny + 2h
    int nw;          // 64-bit words per row of the bit planes (ceil(nxh / 64))
    int bnd;         // BND_*
    int rows;        // rows processed: ny (generic) or ny-1 (wrapper, ref: seabreeze_diag_python.f90:165)
};

// Map an interior 0-based (xs, ys), possibly outside [0,nx) x [0,ny), to coordinates in
// the (nxh, nyh) arrays.  Returns false when the cell does not exist (BND_HALO only).
__device__ __forceinline__ bool sb_map_cell(const Geo &g, int xs, int ys, int &X, int &Y) {
    if (g.bnd == BND_HALO) {
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
    const T *theta, *mask, *z, *sigma;   // (nxh, nyh)
    // state / outputs, (nx, ny)
    T *ws, *wd, *thc, *sb_con;
    T *out;                         // wrapper: (nx,ny,4) packed output, else nullptr
    // workspace
    T *t0;                          // (nxh, nyh)
    uint64_t *bandbits;             // nyh * nw words: interior cells with |mask| <= maxdist
    uint64_t *clsbits;              // nyh * nw words: mask >= 0 ("land side")
    const T *stats;                 // [0]=std  [1]=r  (sigmoid scalars)
    int thc_ty, thc_ntx, thc_nty;   // k_thc tile rows and tile-grid shape (tiles are 64 x thc_ty cells)
    int *tile_nnmax;                // per thc tile: 0 = no band cell; k_prep raises 1, k_thc leaves the largest radius
    int *counters;                  // [0] cells on the global-memory path, [1] one-class cells
    long long *stamps;              // diagnostic build (-DSB_STAMPS) only: 8 clock stamps per thc tile
};

__device__ __forceinline__ int sb_bit(const uint64_t *bits, int nw, int X, int Y) {
    return (int)((bits[(size_t)Y * nw + (X >> 6)] >> (X & 63)) & 1ull);
}

// Fortran MODULO(a, p) for reals as flang evaluates it: fmod, then fold into [0,p) for p > 0.
template <typename T>
__host__ __device__ inline T sb_modulo(T a, T p) {
    T r = fmod(a, p);
    if ((a < T(0)) != (p < T(0))) {
        if (r == T(0)) r = -r; else r += p;
    }
    return r;
}
