// sb_launch.hpp -- host-side launcher declarations shared by the kernel files and the C ABI.
#pragma once
#include "sb_device.hpp"

#define SB_STATS_MAX_BLOCKS 2048
#define SB_MAX_LDS_HALO 32          // largest LDS halo a contrast kernel is instantiated for (k_thc2; 24: k_thc)
#define SB_DIST_TY 4                 // rows per k_dist tile
#define SB_PROF_EVENTS 8            // events one profiled diag call records
#define SB_PROF_KERNELS 5           // k_scan, k_wind, k_t0, k_thc2, (unused)

// Everything of the context a diag launch needs besides the job itself.
struct SbLaunchCtx {
    hipStream_t stream;             // every kernel of the call is enqueued here
    hipEvent_t *prof;               // SB_PROF_EVENTS timing events of this call, or nullptr
    Moments *partials;              // per-workgroup reduction partials
    void *stats;                    // sigmoid scalars (4 x T)
    const Moments *gathered;        // per-band sigma moments to merge instead of scanning sigma, or nullptr
    int ngathered;
    Moments *moments_out;           // band step: k_scan's own moments go here (k_moments_final) ...
    hipEvent_t moments_event;       // ... and this event is recorded behind them, or nullptr
    int ncu;                        // compute units (k_scan / k_thc run one workgroup per CU)
    int phases;                     // bit 0: k_scan + k_wind (no ghost cells, no statistics needed);
                                    // bit 1: statistics merge, k_t0, k_thc2.  3 = the whole call
};

template <typename T>
hipError_t sb_launch_stats(const T *ary, int nx, int ny, int ld, size_t off0, Moments *partials, T *stats,
                           Moments *moments_out, hipStream_t st);   // moments_out: publish moments, not scalars
template <typename T>
hipError_t sb_launch_sigmoid_apply(const T *ary, T *sm, size_t n, const T *stats, hipStream_t st);
template <typename T>
hipError_t sb_launch_diag(const DiagJob<T> &job, int H, const SbLaunchCtx &lc);
void sb_thc_tile_shape(int H, int nx, int rows, int ncu, int *tx, int *ty);         // tile size of the contrast kernel that will run                                         // k_thc tiles are 64 x this many cells
// the fused second half (H <= 16): partials/nparts: k_scan's moments to merge (0: read job.stats)
template <typename T>
hipError_t sb_launch_thc2(const DiagJob<T> &job, int H, int ncu, const Moments *partials, int nparts, T *stats_out,
                          hipStream_t st);

template <typename T>
hipError_t sb_launch_edges(const T *lsm, const T *ci, T *coast, int nx, int ny, int rule, int bnd, hipStream_t st);
// phi: d2r*lat (ny), lamf: folded d2r*lon (nx), both device pointers
template <typename T>
hipError_t sb_launch_dist(const T *coast, const T *mask, const T *phi, const T *lamf, T *cdist, int nx, int ny,
                          int k, T maxdist, uint64_t *bits,   // bits: ny*ceil(nx/64) words of workspace, or nullptr
                          hipStream_t st);

// local part of swap_bounds: E-W periodic ghost columns, pole-side ghost rows replicate the edge row
template <typename T>
hipError_t sb_launch_fill_ghosts(T *field, int nx, int ny, int h, int south, int north, hipStream_t st);
