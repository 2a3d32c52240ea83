// sb_halo_kernels.hip -- local part of the ghost-cell fill of a latitude band.
//
// A band owns `ny` rows with full longitude circles inside a frame of `h` ghost cells.
// North/south ghost rows arrive from the band neighbours (RCCL send/recv, sb_capi.hip);
// what is left is local: the E-W ghost columns are the periodic wrap of the row itself,
// and a band that touches a pole replicates its edge row (the latitude clamp of the
// global-grid rule).  One thread per ghost cell of the rim (the interior is not visited).   replaces: swap_bounds, ref: generic/halo_exchange_mod.f90:12-17
#include "sb_device.hpp"
#include "sb_launch.hpp"

template <typename T>
__global__ __launch_bounds__(256) void k_fill_ghosts(T *__restrict__ f, int nx, int ny, int h, int south, int north) {
    // Only the rim is visited: section A = the 2h ghost columns of every row of the frame (periodic wrap of the
    // row itself; rows that arrived from a band neighbour included), section B = the interior columns of the
    // pole-side ghost rows (replicas of the edge row).  Sources are interior cells or received ghost rows, never
    // targets of this kernel, so the order of threads does not matter.
    const int nxh = nx + 2 * h, nyh = ny + 2 * h;
    const long long nA = (long long)nyh * 2 * h, nB = (long long)((south ? h : 0) + (north ? h : 0)) * nx;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nA + nB; i += (long long)gridDim.x * 256) {
        int X, Y;
        if (i < nA) {
            Y = (int)(i / (2 * h));
            const int j = (int)(i - (long long)Y * 2 * h);
            X = j < h ? j : nx + j;                          // 0..h-1 and nx+h..nx+2h-1
        } else {
            const long long k = i - nA;
            int r = (int)(k / nx);
            X = h + (int)(k - (long long)r * nx);
            if (south && r < h) Y = r;
            else { if (south) r -= h; Y = ny + h + r; }
        }
        int Ys = Y;
        if (south && Y < h) Ys = h;
        else if (north && Y >= ny + h) Ys = ny + h - 1;
        int Xs = X;
        if (X < h) Xs = X + nx;
        else if (X >= nx + h) Xs = X - nx;
        if (Xs != X || Ys != Y) f[(size_t)Y * nxh + X] = f[(size_t)Ys * nxh + Xs];
    }
}

template <typename T>
hipError_t sb_launch_fill_ghosts(T *field, int nx, int ny, int h, int south, int north, hipStream_t st) {
    if (h < 1) return hipSuccess;
    const long long n = (long long)(ny + 2 * h) * 2 * h + (long long)((south ? h : 0) + (north ? h : 0)) * nx;
    int nblk = (int)((n + 255) / 256);
    if (nblk > 2048) nblk = 2048;
    if (nblk < 1) nblk = 1;
    hipLaunchKernelGGL(k_fill_ghosts<T>, dim3(nblk), dim3(256), 0, st, field, nx, ny, h, south, north);
    return hipGetLastError();
}

template hipError_t sb_launch_fill_ghosts<float>(float *, int, int, int, int, int, hipStream_t);
template hipError_t sb_launch_fill_ghosts<double>(double *, int, int, int, int, int, hipStream_t);
