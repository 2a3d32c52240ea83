// sb_launch.hpp -- host-side launcher declarations shared by the kernel files and the C ABI.
#pragma once
#include "sb_device.hpp"

#define SB_STATS_MAX_BLOCKS 2048
#define SB_MAX_LDS_HALO 24          // largest LDS halo k_thc is instantiated for
#define SB_DIST_TY 4                 // rows per k_dist tile

template <typename T>
hipError_t sb_launch_stats(const T *ary, int nx, int ny, int ld, size_t off0, Moments *partials,
                           unsigned int *ticket, T *stats, Moments *moments_out, hipStream_t st);
template <typename T>
hipError_t sb_launch_sigmoid_apply(const T *ary, T *sm, size_t n, const T *stats, hipStream_t st);
template <typename T>
hipError_t sb_launch_diag(const DiagJob<T> &job, int H, Moments *partials, unsigned int *ticket, T *stats,
                          hipStream_t st, hipEvent_t *ev,    // ev: 5 events bracketing the 4 kernels, or nullptr
                          const Moments *gathered, int ngathered,    // non-null: merge these instead of scanning sigma
                          int ncu);                                  // compute units (k_thc runs one workgroup per CU)
int sb_thc_tile_rows(int H);                                         // k_thc tiles are 64 x this many cells
template <typename T>
hipError_t sb_launch_thc(const DiagJob<T> &job, int H, int ncu, hipStream_t st);

template <typename T>
hipError_t sb_launch_edges(const T *lsm, const T *ci, T *coast, int nx, int ny, int rule, int bnd, hipStream_t st);
// phi: d2r*lat (ny), lamf: folded d2r*lon (nx), both device pointers
template <typename T>
hipError_t sb_launch_dist(const T *coast, const T *mask, const T *phi, const T *lamf, T *cdist, int nx, int ny,
                          int k, T maxdist, hipStream_t st);
