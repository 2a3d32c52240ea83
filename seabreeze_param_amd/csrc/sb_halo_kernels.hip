// sb_halo_kernels.hip -- local part of the ghost-cell fill of a latitude band.
//
// A band owns `ny` rows with full longitude circles inside a frame of `h` ghost cells.
// North/south ghost rows arrive from the band neighbours (RCCL send/recv, sb_capi.hip);
// what is left is local: the E-W ghost columns are the periodic wrap of the row itself,
// and a band that touches a pole replicates its edge row (the latitude clamp of the
// global-grid rule).  One thread per ghost cell; sources are never targets, so the order
// of threads does not matter.   replaces: swap_bounds, ref: generic/halo_exchange_mod.f90:12-17
#include "sb_device.hpp"
#include "sb_launch.hpp"

template <typename T>
__global__ __launch_bounds__(256) void k_fill_ghosts(T *__restrict__ f, int nx, int ny, int h, int south, int north) {
    const int nxh = nx + 2 * h;
    const int X = blockIdx.x * 256 + threadIdx.x, Y = blockIdx.y;
    if (X >= nxh) return;
    int Ys = Y;
    if (south && Y < h) Ys = h;
    else if (north && Y >= ny + h) Ys = ny + h - 1;
    int Xs = X;
    if (X < h) Xs = X + nx;
    else if (X >= nx + h) Xs = X - nx;
    if (Xs != X || Ys != Y) f[(size_t)Y * nxh + X] = f[(size_t)Ys * nxh + Xs];
}

template <typename T>
hipError_t sb_launch_fill_ghosts(T *field, int nx, int ny, int h, int south, int north, hipStream_t st) {
    if (h < 1) return hipSuccess;
    hipLaunchKernelGGL(k_fill_ghosts<T>, dim3((nx + 2 * h + 255) / 256, ny + 2 * h), dim3(256), 0, st, field, nx, ny, h,
                       south, north);
    return hipGetLastError();
}

template hipError_t sb_launch_fill_ghosts<float>(float *, int, int, int, int, int, hipStream_t);
template hipError_t sb_launch_fill_ghosts<double>(double *, int, int, int, int, int, hipStream_t);
