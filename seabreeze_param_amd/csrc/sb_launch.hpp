// sb_launch.hpp -- host-side launcher declarations shared by the kernel files and the C ABI.
#pragma once
#include "sb_device.hpp"

#define SB_STATS_MAX_BLOCKS 2048
#define SB_MAX_LDS_HALO 32          // largest LDS halo the contrast kernel (k_thc3) is instantiated for
#define SB_DIST_TY 4                // rows per k_dist tile
// kernels of a diag call that sb_profile_begin / sb_profile_end time, each with its own pair of events
enum { SB_PROF_SCAN = 0, SB_PROF_WIND = 1, SB_PROF_T0 = 2, SB_PROF_THC = 3, SB_PROF_PREP = 4, SB_PROF_KERNELS = 5 };
#define SB_PROF_EVENTS (2 * SB_PROF_KERNELS)
#ifndef SB_WIND_UN
#define SB_WIND_UN 8                // levels of a p column in flight per lane of k_wind
#endif
#ifndef SB_WIND_WGS_PER_CU
#define SB_WIND_WGS_PER_CU 4        // persistent 256-thread workgroups of k_wind per compute unit (= its waves per SIMD:
                                    // the gather is bound by the memory system from two workgroups per CU upwards,
                                    // tools/probe_gather.hip, and 128 registers keep the update code free of spills)
#endif
#ifndef SB_WIND_UN_F32
#define SB_WIND_UN_F32 14           // ... and the two constants in single precision: a load brings half the bytes, so a wave
#endif                              // keeps more of them in flight and the update's registers (57) allow eight waves per SIMD.
#ifndef SB_WIND_WGS_PER_CU_F32      // Measured on one box (k_wind, 5120x3840x56 fp32): 8 loads x 4 workgroups 138.6 us,
#define SB_WIND_WGS_PER_CU_F32 8    // 16 x 4 131.8, 16 x 6 126.4, 8 x 8 120.3, 14 x 7 119.1, 19 x 8 117.6, 28 x 8 118.1,
#endif                              // 14 x 8 114.3-118.1; 2560x1920x56 fp32: 50.1 -> 39.7 us

// Everything of the context a diag launch needs besides the job itself.
struct SbLaunchCtx {
    hipStream_t stream;             // every kernel of the call is enqueued here
    hipEvent_t *prof;               // SB_PROF_EVENTS timing events of this call, or nullptr
    unsigned *prof_mask;            // bit k set: kernel k was launched (and its event pair recorded) in this call
    Moments *partials;              // per-workgroup reduction partials
    void *stats;                    // sigmoid scalars (4 x T)
    const Moments *gathered;        // per-band sigma moments to merge instead of scanning sigma, or nullptr
    int ngathered;
    Moments *moments_out;           // band step: this band's sigma moments go here (k_scan's last workgroup merges them) ...
    hipEvent_t moments_event;       // ... and this event is recorded behind k_scan, or nullptr
    int *stats_ticket;              // ... with this device word (zero between launches) as the workgroups' ticket
    int ncu;                        // compute units (k_scan and the contrast kernel run one workgroup per CU)
    int phases;                     // bit 0: k_scan + k_prep + k_wind (no ghost cells, no statistics needed);
                                    // bit 1: statistics of all bands, k_t0, k_thc3.  3 = the whole call
    bool reuse_stats;               // the sigmoid scalars in `stats` stand (static sigma): no moments, no merge
    bool no_fold;                   // keep k_prep as a kernel of its own (sb_set_fold(ctx, 0): measurement and tests)
    bool segs_stand;                // band step on the strip kernel: the segment lists of the call before are in place
                                    // (same geometry; k_wind checks on the device that the planes did not change): no k_prep
    int *launches;                  // += kernels enqueued by the call, or nullptr
};

template <typename T>
hipError_t sb_launch_stats(const T *ary, int nx, int ny, int ld, size_t off0, Moments *partials, T *stats,
                           Moments *moments_out, int *ticket, hipStream_t st);   // moments_out: publish moments, not scalars;
                                                                                // ticket: a device word that is zero between launches
template <typename T>
hipError_t sb_launch_sigmoid_apply(const T *ary, T *sm, size_t n, const T *stats, hipStream_t st);
template <typename T>
hipError_t sb_launch_diag(const DiagJob<T> &job, int H, const SbLaunchCtx &lc);
// tile size of the tile contrast kernel (k_thc3) for an LDS halo of H = 24 or 32 cells
void sb_thc_tile_shape(int H, int *tx, int *ty);
// the tile contrast kernel (halos of 24 and 32 cells): reads the list of active tiles and the sigmoid scalars k_prep left
template <typename T>
hipError_t sb_launch_thc(const DiagJob<T> &job, int H, int ncu, hipStream_t st);

// the marching-strip contrast kernel (sb_strip_kernel.hip): LDS halo of 16 cells, ranks k_scan's flags itself
template <typename T>
hipError_t sb_launch_strip(const DiagJob<T> &job, int ncu, hipStream_t st);
// its block grid (strips x 16-row blocks) for nx x rows interior cells; false: the grid is too large for it
bool sb_strip_shape(int nx, int rows, int *ntx, int *nty);
// the marching-strip kernel for search radii up to 31 (sb_strip32_kernel.hip): single precision only (hipErrorInvalidValue
// for double); its flags carry TWO virtual blocks above and below every strip
template <typename T>
hipError_t sb_launch_strip32(const DiagJob<T> &job, int ncu, hipStream_t st);
template <>
hipError_t sb_launch_strip32<float>(const DiagJob<float> &job, int ncu, hipStream_t st);
template <>
hipError_t sb_launch_strip32<double>(const DiagJob<double> &job, int ncu, hipStream_t st);
bool sb_strip32_shape(int nx, int rows, int *ntx, int *nty);

// theta <- theta - (gmma*z)*sigmoid(sigma) over n cells, with the scalars the last diag call left in `stats`
template <typename T>
hipError_t sb_launch_theta_to_t0(T *theta, const T *z, const T *sigma, size_t n, const T *stats, hipStream_t st);

template <typename T>
hipError_t sb_launch_edges(const T *lsm, const T *ci, T *coast, int nx, int ny, int rule, int bnd, hipStream_t st);
// phi: d2r*lat (ny), lamf: folded d2r*lon (nx), both device pointers
template <typename T>
hipError_t sb_launch_dist(const T *coast, const T *mask, const T *phi, const T *lamf,
                          const T *shl, const T *chl,         // sin, cos of half the folded longitudes (k_dist_bits, fp64)
                          T *cdist, int nx, int ny,
                          int k, T maxdist, uint64_t *bits,   // bits: ny*ceil(nx/64) words of workspace, or nullptr
                          int nearest,                        // 1: nearest hit per side of a source row only (see k_dist_bits)
                          hipStream_t st);

// local part of swap_bounds: E-W periodic ghost columns, pole-side ghost rows replicate the edge row
template <typename T>
hipError_t sb_launch_fill_ghosts(T *field, int nx, int ny, int h, int south, int north, hipStream_t st);
