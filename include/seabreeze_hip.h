/*
 * seabreeze_hip.h -- C ABI of libseabreeze_hip.so, the MI355X (gfx950) implementation
 * of the sea-breeze trigger diagnostic's hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch / C++ types.
 * The Fortran host modules (fortran/ *.f90, ISO_C_BINDING) and the f2py-surface Python
 * module (python_wrapper/) bind exactly these entry points; INTEGRATION.md shows the
 * reference-side stubs.  Paths cited as "ref:" are relative to the reference tree
 * (antarcticrainforest/seabreeze_param).
 *
 * Conventions
 *   - Arrays are Fortran order, longitude fastest: f(lon, lat[, lev]); contiguous.
 *   - Two precisions: _f32 (default REAL) and _f64 (REAL under -fdefault-real-8),
 *     mirroring how the reference picks its working precision at compile time.
 *   - Entry points without a suffix take HOST pointers (drop-in semantics: the caller
 *     owns every array, ref: SURVEY.md §8(b) "Ownership"); the library stages them
 *     through device buffers it owns.  `_dev` entry points take DEVICE pointers and
 *     enqueue on `stream` (a hipStream_t passed as void*; NULL = the context's own
 *     stream) without synchronising -- this is what a GPU-resident host model and
 *     bench.py use.
 *   - Every function returns SB_OK (0) or an sb_status; sb_last_error() gives text.
 *     The reference has no error reporting on this path (generic + wrapper) and an
 *     `error` out-argument in the UM copy (ref: UM/vn10.7/sea_breeze_diag.F90:102,
 *     198-202); the Fortran modules map a non-zero status to `error stop` / `error`.
 *   - There is NO CPU fallback: without a usable gfx950 device sb_create fails with
 *     SB_ERR_NO_DEVICE and nothing else can be called.
 */
#ifndef SEABREEZE_HIP_H
#define SEABREEZE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sb_ctx sb_ctx;

typedef enum sb_status {
    SB_OK = 0,
    SB_ERR_ARG = 1,        /* bad dimension / NULL pointer / unsupported combination   */
    SB_ERR_HIP = 2,        /* a HIP runtime call failed                                */
    SB_ERR_NO_DEVICE = 3,  /* no gfx950 device visible                                 */
    SB_ERR_ALLOC = 4,      /* device workspace allocation failed                       */
    SB_ERR_COMM = 5        /* RCCL halo exchange failed / not initialised              */
} sb_status;

/* Window / neighbour index rule at the domain edge. */
typedef enum sb_boundary {
    /* lat clamp, lon `max(1, modulo(jj, nlons))` -- reproduces the f2py kernels
       verbatim, including the lon quirk (ref: python_wrapper/seabreezediag/
       seabreeze_diag_python.f90:201-202, sobel.f90:67-68) */
    SB_BND_WRAPPER = 0,
    /* lat clamp, true periodic lon: the single-domain reading of the generic code's
       unguarded window (ref: generic/sea_breeze_diag.f90:198-200)                    */
    SB_BND_GLOBAL = 1,
    /* raw reads into `halo` ghost cells supplied by the caller around theta, mask, z
       and sigma (UM-style bounds 1-h:n+h, ref: UM/vn10.7/sea_breeze_diag.F90:94-99,
       241-243); the ghost cells are what swap_bounds fills                            */
    SB_BND_HALO = 2
} sb_boundary;

/* Tunables of the trigger in SI units (ref: generic/sea_breeze_diag.f90:127-138). */
typedef struct sb_tunables {
    double target_plev_pa;   /* 70000.  */
    double thresh_wind;      /* 11.     */
    double thresh_winddir;   /* 90.     */
    double thresh_windch;    /* 5.      */
    double thresh_thc;       /* 0.75    */
    double target_time_s;    /* 21600.  */
    double maxdist_km;       /* 180.    */
} sb_tunables;

/* -------------------------------------------------------------------------------- */
/* context                                                                          */
/* -------------------------------------------------------------------------------- */
int  sb_create(sb_ctx **ctx, int device);            /* device < 0: current device   */
int  sb_destroy(sb_ctx *ctx);
const char *sb_last_error(const sb_ctx *ctx);        /* ctx may be NULL               */
const char *sb_version(void);
void sb_default_tunables(sb_tunables *t);
/* Expected largest search radius of the land/sea window: up to 16 the marching-strip contrast
   kernel runs (LDS halo 16); beyond it, in single precision, the 96-column strip kernel (radii up to 31 from LDS),
   in double precision the tile kernel with a halo of 24 or 32 cells.  Correctness never depends on it -- cells that
   need more take a global-memory path --, and t0 is formed by one sequence of operations in every kernel; the window
   means are exact fixed-point sums in the strip kernels and rounded sums in the tile kernel and on the global path,
   so thc may differ between kernels in its last bits (fp64: a few ulp; the tests hold 1e-7 against the reference). */
int  sb_set_search_radius_hint(sb_ctx *ctx, int radius);
/* Single-domain host-model calls let the strip contrast kernel merge k_scan's statistics and compact
   k_wind's segment lists itself (on, the default) or leave that to a kernel of its own between k_scan and
   the contrast kernel (off).  A measurement and test knob: results never depend on it.                  */
int  sb_set_fold(sb_ctx *ctx, int on);
/* The strip contrast kernel's plan -- which blocks each workgroup marches over, in which order, and where their
   coastal-band cells lie -- follows from the band plane (|mask| <= maxdist) alone.  It is kept in device memory from
   call to call; k_scan compares every word of the plane it writes with the word the call before left, and the first
   difference makes the kernel plan afresh (on, the default).  Off: it plans every call.  A measurement and test
   knob: results never depend on it.                                                                       */
int  sb_set_plan_cache(sb_ctx *ctx, int on);
/* Search radii beyond 16 (radius hints of 17 .. 32): in single precision the 96-column marching-strip kernel answers
   windows of up to 31 cells from LDS (on, the default); off: the tile kernel, as in double precision.  A measurement and
   test knob: results agree to single-precision rounding of the window means.                                      */
int  sb_set_wide_strip(sb_ctx *ctx, int on);
/* A band step (sb_band_seabreeze_diag_*_dev, or a diag call with gathered moments in use) on the strip kernel runs
   k_scan and k_wind ahead of the join with the communication stream and lets the contrast kernel apply the update
   (0, the default), or k_scan | join | contrast kernel, k_wind -- the three kernels of a single-domain call, one
   launch less, but less work to cover the communication with (1).  A measurement knob: results never depend on it.  */
int  sb_set_band_order(sb_ctx *ctx, int contrast_first);
/* The kernels that run one persistent workgroup per compute unit (k_scan, the contrast kernel; k_wind four per unit)
   take n workgroups instead (1 .. the device's compute units; 0: back to the default).  A test knob -- with a few
   workgroups a small grid exercises what only grids far beyond the BASELINE sizes reach otherwise: shares of the
   strip kernel's march that span several rounds and hold more query steps than a stored plan does.  Results never
   depend on it.                                                                                                  */
int  sb_set_workgroups(sb_ctx *ctx, int n);
/* Opt-in, off by default: the caller states that sigma (the sub-grid orography deviation, an ancillary that a
   host model reads once; ref: generic/sea_breeze_diag.f90:159-166 recomputes its mean and deviation every
   call) does not change between calls.  The first complete diag / band step after the switch forms the
   statistics as usual; later calls that pass the same array (address, shape, ghost width, precision) reuse its
   sigmoid scalars: k_scan stops reading sigma (one of its three planes) and a band step drops the moments
   all-gather and the merge.  Results are those of the default path as long as the statement holds; any other
   sigma array, or switching the option again, forms the statistics anew.                                  */
int  sb_set_static_sigma(sb_ctx *ctx, int on);
/* What the last diag call or band step enqueued on this rank: [0] kernel launches, [1] RCCL operations (sends,
   receives, all-gathers), [2] RCCL groups, [3] device-to-device copies.  No synchronisation.               */
int  sb_last_step_report(sb_ctx *ctx, int report[4]);
/* Counters of the last diag call: [0] band cells, [1] cells that left the LDS path,
   [2] cells whose search found only one class (result NaN; the reference loops
   forever there, ref: generic/sea_breeze_diag.f90:191-214), [3] max radius used.
   Synchronises the context's stream.                                                */
int  sb_last_counters(sb_ctx *ctx, long long counters[4]);
int  sb_synchronize(sb_ctx *ctx);
/* Per-kernel timing with HIP events on the stream the kernels run on.  Between sb_profile_begin and
   sb_profile_end the first `max_calls` diag calls record a pair of events around each of their
   launches; sb_profile_end synchronises and returns the average duration in ms of
   [0] k_scan [1] k_wind [2] k_t0 (f2py flavour only; 0 where a call does not launch it) [3] k_thc3
   [4] k_prep.  An event interval also holds the gap to the previous launch (about 5 us).          */
int  sb_profile_begin(sb_ctx *ctx, int max_calls);
int  sb_profile_end(sb_ctx *ctx, double avg_ms[5], int *ncalls);

/* -------------------------------------------------------------------------------- */
/* seabreeze_diag -- host-model flavour                                              */
/* replaces: subroutine seabreeze_diag(timestep, timestep_number, p, u, v, theta,   */
/*           mask, z, sigma, windspeed, winddir, thc, sb_con)                        */
/*           ref: generic/sea_breeze_diag.f90:55-271                                 */
/* p,u,v: (nx,ny,nz).  theta,mask,z,sigma: (nx+2*halo, ny+2*halo), interior at       */
/* offset `halo` (halo must be 0 unless bnd == SB_BND_HALO).                         */
/* windspeed,winddir,thc,sb_con: (nx,ny), updated in place.                          */
/* tun == NULL selects the reference's compile-time constants.                       */
/* -------------------------------------------------------------------------------- */
int sb_seabreeze_diag_f64(sb_ctx *ctx, double timestep_s, int timestep_number,
                          int nx, int ny, int nz, int halo, int bnd,
                          const double *p, const double *u, const double *v,
                          const double *theta, const double *mask, const double *z,
                          const double *sigma, double *windspeed, double *winddir,
                          double *thc, double *sb_con, const sb_tunables *tun);
int sb_seabreeze_diag_f32(sb_ctx *ctx, float timestep_s, int timestep_number,
                          int nx, int ny, int nz, int halo, int bnd,
                          const float *p, const float *u, const float *v,
                          const float *theta, const float *mask, const float *z,
                          const float *sigma, float *windspeed, float *winddir,
                          float *thc, float *sb_con, const sb_tunables *tun);
int sb_seabreeze_diag_f64_dev(sb_ctx *ctx, double timestep_s, int timestep_number,
                          int nx, int ny, int nz, int halo, int bnd,
                          const double *p, const double *u, const double *v,
                          const double *theta, const double *mask, const double *z,
                          const double *sigma, double *windspeed, double *winddir,
                          double *thc, double *sb_con, const sb_tunables *tun,
                          void *stream);
int sb_seabreeze_diag_f32_dev(sb_ctx *ctx, float timestep_s, int timestep_number,
                          int nx, int ny, int nz, int halo, int bnd,
                          const float *p, const float *u, const float *v,
                          const float *theta, const float *mask, const float *z,
                          const float *sigma, float *windspeed, float *winddir,
                          float *thc, float *sb_con, const sb_tunables *tun,
                          void *stream);

/* -------------------------------------------------------------------------------- */
/* seabreeze_diag -- UM vn10.7 field layout                                          */
/* replaces: subroutine seabreeze_diag(timestep, timestep_number, p, u, v, theta, z, */
/*           sigma, mask, windspeed, winddir, thc, sb_con, error)                    */
/*           ref: UM/vn10.7/sea_breeze_diag.F90:55-56 (argument order), :66-117      */
/*           (bounds), :198-202 (error), :210-211 (theta <- t0), :265-274 (level)    */
/* p, u, v, sb_con on pdims and windspeed, winddir, thc on tdims: (nx,ny[,nz]), no    */
/* ghost cells.  theta, z, sigma on tdims_s: (nx+2*halo_s, ny+2*halo_s).  mask (the   */
/* signed coast distance) on tdims_l: (nx+2*halo_l, ny+2*halo_l), halo_l >= halo_s.   */
/* The window reads both t0 and mask, so it can use halo_s ghost cells; a cell whose  */
/* window would need more yields NaN and is counted (sb_last_counters [2]).           */
/* *error = 1 and nothing done when ny < 1 or nz < 1 (UM :198-202), else 0.           */
/* flags: SB_UM_THETA_TO_T0 -- theta comes back as t0 = theta - gmma*z*sigmoid(sigma), */
/*        ghost cells included (UM :210-211; the arithmetic does not depend on it);    */
/*        SB_UM_LEVEL_WALK -- the UM copy's level rule: upwards from level 1 while     */
/*        |p - 70000| does not grow (ties move on), stop at the first increase         */
/*        (:265-274), instead of the generic file's first minimum over all levels.     */
/* The UM file cannot be compiled outside the UM: this entry point    */
/* follows its text; parity is pinned only through the generic/wrapper arithmetic.     */
/* -------------------------------------------------------------------------------- */
#define SB_UM_THETA_TO_T0 1
#define SB_UM_LEVEL_WALK  2
int sb_seabreeze_diag_um_f64(sb_ctx *ctx, double timestep_s, int timestep_number, int nx, int ny, int nz,
                             int halo_s, int halo_l, const double *p, const double *u, const double *v,
                             double *theta, const double *z, const double *sigma, const double *mask,
                             double *windspeed, double *winddir, double *thc, double *sb_con, int flags,
                             int *error);
int sb_seabreeze_diag_um_f32(sb_ctx *ctx, float timestep_s, int timestep_number, int nx, int ny, int nz,
                             int halo_s, int halo_l, const float *p, const float *u, const float *v,
                             float *theta, const float *z, const float *sigma, const float *mask,
                             float *windspeed, float *winddir, float *thc, float *sb_con, int flags,
                             int *error);
int sb_seabreeze_diag_um_f64_dev(sb_ctx *ctx, double timestep_s, int timestep_number, int nx, int ny, int nz,
                             int halo_s, int halo_l, const double *p, const double *u, const double *v,
                             double *theta, const double *z, const double *sigma, const double *mask,
                             double *windspeed, double *winddir, double *thc, double *sb_con, int flags,
                             int *error, void *stream);
int sb_seabreeze_diag_um_f32_dev(sb_ctx *ctx, float timestep_s, int timestep_number, int nx, int ny, int nz,
                             int halo_s, int halo_l, const float *p, const float *u, const float *v,
                             float *theta, const float *z, const float *sigma, const float *mask,
                             float *windspeed, float *winddir, float *thc, float *sb_con, int flags,
                             int *error, void *stream);

/* -------------------------------------------------------------------------------- */
/* diag -- f2py-surface flavour (whole global grid, 1-D p, packed output)            */
/* replaces: subroutine diag(timestep_number, p, z, std, theta, v, u, cdist,         */
/*           windspeed, winddir, thc, target_plev, thresh_wind, thresh_winddir,      */
/*           thresh_windch, thresh_thc, target_time, maxdist, timestep, nps, nlons,  */
/*           nlats, output)                                                          */
/*           ref: python_wrapper/seabreezediag/seabreeze_diag_python.f90:49-285      */
/* Units as at that surface: target_plev hPa, target_time h, timestep min; unlike    */
/* the reference they are NOT overwritten in place (:146-148).  Rows 1..nlats-1 are  */
/* processed, row nlats of `output` is left untouched (:165).  windspeed, winddir,   */
/* thc are updated in place like the reference's dummies.                            */
/* output: (nlons,nlats,4) = sb_con, t0, windspeed, winddir (:277-280).              */
/* -------------------------------------------------------------------------------- */
int sb_diag_f64(sb_ctx *ctx, int timestep_number, const double *p, const double *z,
                const double *std, const double *theta, const double *v, const double *u,
                const double *cdist, double *windspeed, double *winddir, double *thc,
                double target_plev, double thresh_wind, double thresh_winddir,
                double thresh_windch, double thresh_thc, double target_time,
                double maxdist, double timestep, int nps, int nlons, int nlats,
                double *output);
int sb_diag_f32(sb_ctx *ctx, int timestep_number, const float *p, const float *z,
                const float *std, const float *theta, const float *v, const float *u,
                const float *cdist, float *windspeed, float *winddir, float *thc,
                float target_plev, float thresh_wind, float thresh_winddir,
                float thresh_windch, float thresh_thc, float target_time,
                float maxdist, float timestep, int nps, int nlons, int nlats,
                float *output);
int sb_diag_f64_dev(sb_ctx *ctx, int timestep_number, const double *p, const double *z,
                const double *std, const double *theta, const double *v, const double *u,
                const double *cdist, double *windspeed, double *winddir, double *thc,
                double target_plev, double thresh_wind, double thresh_winddir,
                double thresh_windch, double thresh_thc, double target_time,
                double maxdist, double timestep, int nps, int nlons, int nlats,
                double *output, void *stream);
int sb_diag_f32_dev(sb_ctx *ctx, int timestep_number, const float *p, const float *z,
                const float *std, const float *theta, const float *v, const float *u,
                const float *cdist, float *windspeed, float *winddir, float *thc,
                float target_plev, float thresh_wind, float thresh_winddir,
                float thresh_windch, float thresh_thc, float target_time,
                float maxdist, float timestep, int nps, int nlons, int nlats,
                float *output, void *stream);

/* -------------------------------------------------------------------------------- */
/* diag, streamed: many timesteps on one grid                */
/* replaces: the per-timestep loop of the reference's Python driver,                 */
/*           ref: python_wrapper/seabreezediag/__init__.py:222-245 -- every step it   */
/*           hands diag the same z, std, cdist and the state of the step before.      */
/* sb_diag_stream_begin uploads those six planes once; sb_diag_stream_step takes one  */
/* step's p(nps), theta(nlons,nlats), v, u (nlons,nlats,nps) -- host pointers, units  */
/* and tunables as sb_diag_* -- uploads theta and the one u, v plane p selects through */
/* pinned double-buffered staging and enqueues the step; it hands back the sb_con     */
/* plane of the PREVIOUS step (rows 1..nlats-1 of sb_con_prev, as double, what the    */
/* reference's driver accumulates; *have_prev = 0 on the first step), so that staging  */
/* step i+1 overlaps the device work of step i.  sb_diag_stream_end waits, returns the */
/* last step's sb_con, the last step's four output planes and the final state (any of  */
/* them may be NULL).  Results are those of sb_diag_* called step by step.             */
/* sb_diag_stream_stats: steps so far and where the host spent its time, in seconds:   */
/* [0] host copies (staging, sb_con out) [1] enqueueing [2] waiting for the device.    */
/* -------------------------------------------------------------------------------- */
int sb_diag_stream_begin_f32(sb_ctx *ctx, int nlons, int nlats, const float *z, const float *std,
                             const float *cdist, const float *windspeed, const float *winddir, const float *thc);
int sb_diag_stream_begin_f64(sb_ctx *ctx, int nlons, int nlats, const double *z, const double *std,
                             const double *cdist, const double *windspeed, const double *winddir, const double *thc);
int sb_diag_stream_step_f32(sb_ctx *ctx, int timestep_number, const float *p, int nps, const float *theta,
                            const float *v, const float *u, float target_plev, float thresh_wind,
                            float thresh_winddir, float thresh_windch, float thresh_thc, float target_time,
                            float maxdist, float timestep, double *sb_con_prev, int *have_prev);
int sb_diag_stream_step_f64(sb_ctx *ctx, int timestep_number, const double *p, int nps, const double *theta,
                            const double *v, const double *u, double target_plev, double thresh_wind,
                            double thresh_winddir, double thresh_windch, double thresh_thc, double target_time,
                            double maxdist, double timestep, double *sb_con_prev, int *have_prev);
int sb_diag_stream_end_f32(sb_ctx *ctx, double *sb_con_last, float *output4, float *windspeed, float *winddir,
                           float *thc);
int sb_diag_stream_end_f64(sb_ctx *ctx, double *sb_con_last, double *output4, double *windspeed, double *winddir,
                           double *thc);
int sb_diag_stream_stats(sb_ctx *ctx, long *steps, double seconds[3]);

/* -------------------------------------------------------------------------------- */
/* sigmoid                                                                           */
/* replaces: subroutine sigmoid(ary, sm)  ref: generic/sea_breeze_diag.f90:457-481,  */
/*           seabreeze_diag_python.f90:287-311                                       */
/* -------------------------------------------------------------------------------- */
int sb_sigmoid_f64(sb_ctx *ctx, int nlons, int nlats, const double *ary, double *sm);
int sb_sigmoid_f32(sb_ctx *ctx, int nlons, int nlats, const float *ary, float *sm);
int sb_sigmoid_f64_dev(sb_ctx *ctx, int nlons, int nlats, const double *ary, double *sm, void *stream);
int sb_sigmoid_f32_dev(sb_ctx *ctx, int nlons, int nlats, const float *ary, float *sm, void *stream);

/* Latitude-band decomposition: the sigmoid needs GLOBAL statistics of sigma
   (ref: generic/sea_breeze_diag.f90:466-479).  Each band computes its own moments
   {count, mean, sum of squared deviations, min, max} (5 doubles, device memory), the
   host model all-gathers them (RCCL), and every following diag call on this context
   merges the gathered array (rank order, bit-identical on all ranks) instead of
   scanning its local sigma.  sb_use_gathered_moments(ctx, NULL, 0) switches back.     */
int sb_sigma_moments_f64_dev(sb_ctx *ctx, int nx, int ny, int halo, const double *sigma,
                             double *moments5, void *stream);
int sb_sigma_moments_f32_dev(sb_ctx *ctx, int nx, int ny, int halo, const float *sigma,
                             double *moments5, void *stream);
int sb_use_gathered_moments(sb_ctx *ctx, const double *gathered_dev, int nparts);

/* -------------------------------------------------------------------------------- */
/* get_edges -- coastline by binary 3x3 Sobel                                        */
/* replaces: get_edges(lsm, ci, nlons, nlats, coast)  ref: sobel.f90:19-89 (rule 0)  */
/*           get_edges(mask, icefrac, landfrac, halo_size)                           */
/*                                    ref: generic/sea_breeze_diag.f90:273-373 (rule 1) */
/* rule 0: land <=> lsm+ci > 0.4; rule 1: ice-aware two-branch mask (:325-337).      */
/* -------------------------------------------------------------------------------- */
int sb_get_edges_f64(sb_ctx *ctx, int nlons, int nlats, const double *lsm, const double *ci,
                     int rule, int bnd, double *coast);
int sb_get_edges_f32(sb_ctx *ctx, int nlons, int nlats, const float *lsm, const float *ci,
                     int rule, int bnd, float *coast);
int sb_get_edges_f64_dev(sb_ctx *ctx, int nlons, int nlats, const double *lsm, const double *ci,
                     int rule, int bnd, double *coast, void *stream);
int sb_get_edges_f32_dev(sb_ctx *ctx, int nlons, int nlats, const float *lsm, const float *ci,
                     int rule, int bnd, float *coast, void *stream);

/* -------------------------------------------------------------------------------- */
/* get_dist -- signed great-circle distance (km) to the nearest coast cell           */
/* replaces: get_dist(coast, mask, lon, lat, nlons, nlats, maxdist, cdist)           */
/*                                   ref: sobel.f90:91-193                           */
/*           get_dist(coast, landfrac, lon, lat, maxdist, cdist, halo_size)          */
/*                                   ref: generic/sea_breeze_diag.f90:375-455        */
/* kwin < 0: window half-width from the 70-degree grid spacing (sobel.f90:129-137);  */
/* kwin >= 0: fixed +-kwin cells (the generic/UM halo width, generic :422,425).      */
/* lon, lat: HOST pointers in every variant (tiny vectors; the window size is        */
/* derived from them on the host).                                                   */
/* -------------------------------------------------------------------------------- */
int sb_get_dist_f64(sb_ctx *ctx, int nlons, int nlats, const double *coast, const double *mask,
                    const double *lon, const double *lat, double maxdist, int kwin, double *cdist);
int sb_get_dist_f32(sb_ctx *ctx, int nlons, int nlats, const float *coast, const float *mask,
                    const float *lon, const float *lat, float maxdist, int kwin, float *cdist);
int sb_get_dist_f64_dev(sb_ctx *ctx, int nlons, int nlats, const double *coast, const double *mask,
                    const double *lon, const double *lat, double maxdist, int kwin, double *cdist,
                    void *stream);
int sb_get_dist_f32_dev(sb_ctx *ctx, int nlons, int nlats, const float *coast, const float *mask,
                    const float *lon, const float *lat, float maxdist, int kwin, float *cdist,
                    void *stream);
/* The half-width sb_get_dist would pick for kwin < 0. */
int sb_dist_window_f64(int nlons, int nlats, const double *lon, const double *lat, double maxdist, int *k);
int sb_dist_window_f32(int nlons, int nlats, const float *lon, const float *lat, float maxdist, int *k);

/* -------------------------------------------------------------------------------- */
/* swap_bounds -- ghost-cell fill of one latitude band of a multi-GPU run            */
/* replaces: subroutine swap_bounds(field, halo_size)                                */
/*           ref: generic/halo_exchange_mod.f90:12-17 (an empty stub; the UM copy    */
/*           calls a real one, ref: UM/vn10.7/sea_breeze_diag.F90:408-410)            */
/* One process per GPU owns `ny` rows x full longitude circles inside a frame of     */
/* `halo` ghost cells: field is (nx+2*halo, ny+2*halo) in DEVICE memory.  North and   */
/* south ghost rows are exchanged with the band neighbours (ranks rank-1 / rank+1)    */
/* by ncclSend/ncclRecv in one group over RCCL (xGMI); bands at rank 0 / nranks-1     */
/* replicate their pole-side edge row; E-W ghost columns are the periodic wrap.       */
/* Without sb_comm_init the context is a single band owning the globe (local fill).  */
/* librccl.so is opened on demand by sb_comm_get_unique_id / sb_comm_init.           */
/*   rank 0:    sb_comm_get_unique_id(id);  -- hand the 128 bytes to every rank       */
/*   every rank: sb_comm_init(ctx, id, rank, nranks);                                 */
/* sb_allgather_moments_dev: the 5-double sigma moments of every band, in rank order, */
/* for sb_use_gathered_moments (ncclAllGather).                                       */
/* -------------------------------------------------------------------------------- */
int sb_comm_get_unique_id(unsigned char id[128]);
int sb_comm_init(sb_ctx *ctx, const unsigned char id[128], int rank, int nranks);
int sb_comm_finalize(sb_ctx *ctx);
int sb_swap_bounds_f64_dev(sb_ctx *ctx, double *field, int nx, int ny, int halo, void *stream);
int sb_swap_bounds_f32_dev(sb_ctx *ctx, float *field, int nx, int ny, int halo, void *stream);
int sb_allgather_moments_dev(sb_ctx *ctx, const double *mine5, double *gathered, void *stream);
/* The local part of sb_swap_bounds_*_dev on its own, for a host model that moves the north-south ghost rows with its
   own message passing (ref: generic/halo_exchange_mod.f90:12-17, where swap_bounds stands for "various routines about
   communications across processors"): the east-west ghost columns of every row -- ghost rows included -- become the
   periodic wrap, and a band that touches a pole (south / north != 0) replicates its edge row into the ghost rows on
   that side first.  No communicator is needed or used.                                                          */
int sb_fill_ghosts_f64_dev(sb_ctx *ctx, double *field, int nx, int ny, int halo, int south, int north, void *stream);
int sb_fill_ghosts_f32_dev(sb_ctx *ctx, float *field, int nx, int ny, int halo, int south, int north, void *stream);
/* One seabreeze_diag step of a latitude band, device pointers, as sb_seabreeze_diag_*_dev with
   bnd = SB_BND_HALO -- plus the band's communication: the sigma moments are reduced over all
   bands and theta's ghost cells are filled (swap_bounds) inside the call, on the context's
   second stream, while the kernels that need neither (k_scan, k_wind) run on `stream`.
   mask, z, sigma must already carry their ghost cells (static: exchange them once).       */
int sb_band_seabreeze_diag_f64_dev(sb_ctx *ctx, double timestep_s, int timestep_number, int nx, int ny, int nz,
                                   int halo, const double *p, const double *u, const double *v, double *theta,
                                   const double *mask, const double *z, const double *sigma, double *windspeed,
                                   double *winddir, double *thc, double *sb_con, const sb_tunables *tun,
                                   void *stream);
int sb_band_seabreeze_diag_f32_dev(sb_ctx *ctx, float timestep_s, int timestep_number, int nx, int ny, int nz,
                                   int halo, const float *p, const float *u, const float *v, float *theta,
                                   const float *mask, const float *z, const float *sigma, float *windspeed,
                                   float *winddir, float *thc, float *sb_con, const sb_tunables *tun,
                                   void *stream);

/* Host-pointer forms of the two for host models whose fields live in host memory (what the Fortran
   modules fortran/halo_exchange_mod.f90 and fortran/sea_breeze_diag_mod.F90 bind): the arrays are staged
   through device buffers of the context, the exchange runs over RCCL between the devices, the results come
   back.  sb_swap_bounds_* replaces swap_bounds(field, halo_size), ref: generic/halo_exchange_mod.f90:12-17
   and its call sites generic/sea_breeze_diag.f90:342,371; theta's ghost cells are filled on return from
   sb_band_seabreeze_diag_*.                                                                          */
int sb_swap_bounds_f64(sb_ctx *ctx, double *field, int nx, int ny, int halo);
int sb_swap_bounds_f32(sb_ctx *ctx, float *field, int nx, int ny, int halo);
int sb_band_seabreeze_diag_f64(sb_ctx *ctx, double timestep_s, int timestep_number, int nx, int ny, int nz,
                               int halo, const double *p, const double *u, const double *v, double *theta,
                               const double *mask, const double *z, const double *sigma, double *windspeed,
                               double *winddir, double *thc, double *sb_con, const sb_tunables *tun);
int sb_band_seabreeze_diag_f32(sb_ctx *ctx, float timestep_s, int timestep_number, int nx, int ny, int nz,
                               int halo, const float *p, const float *u, const float *v, float *theta,
                               const float *mask, const float *z, const float *sigma, float *windspeed,
                               float *winddir, float *thc, float *sb_con, const sb_tunables *tun);
/* rank of this context's band and the number of bands; nranks = 0 without a communicator */
int sb_comm_rank(sb_ctx *ctx, int *rank, int *nranks);

/* -------------------------------------------------------------------------------- */
/* device memory for host models that keep their fields resident between calls      */
/* (Fortran: type(c_ptr) handles passed to the _dev entry points); upload and         */
/* download synchronise.                                                              */
/* -------------------------------------------------------------------------------- */
int sb_device_malloc(sb_ctx *ctx, size_t nbytes, void **dptr);
int sb_device_free(sb_ctx *ctx, void *dptr);
int sb_device_upload(sb_ctx *ctx, void *dst_dev, const void *src_host, size_t nbytes);
int sb_device_download(sb_ctx *ctx, void *dst_host, const void *src_dev, size_t nbytes);

/* -------------------------------------------------------------------------------- */
/* get_threads  ref: sobel.f90:195-206 (OpenMP thread count there; here the number   */
/* of visible HIP devices -- the unit of parallelism a caller can spread bands over) */
/* -------------------------------------------------------------------------------- */
int sb_get_threads(int *nt);

#ifdef __cplusplus
}
#endif
#endif /* SEABREEZE_HIP_H */
