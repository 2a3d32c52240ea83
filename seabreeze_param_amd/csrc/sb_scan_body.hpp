// sb_scan_body.hpp -- the one pass over sigma and mask: k_scan's loop (sb_diag_kernels.hip), kept apart from the kernel's
// hand-over code (round 4 built a kernel that ran this pass and the strip kernel's march in one launch; it measured
// slower than two launches and was removed: profiles/r04_fused_scan_strip_ab.txt).
//   * sigma  -> shifted sums about its first interior value -> one (count, s1, s2, min, max) partial per workgroup
//   * mask   -> land-side bit  mask >= 0              ref: generic/sea_breeze_diag.f90:182,200
//            -> band bit  !(|mask| > maxdist)         ref :174
//            -> flag of the contrast-kernel tile(s) the segment's band cells fall in
//            -> fill value outside the band           ref :176 / seabreeze_diag_python.f90:173,279-280
#pragma once
#include "sb_device.hpp"
#ifndef SB_EAGER_PLANES
#define SB_EAGER_PLANES 0          // 1: every word of the planes is written every call (A/B, debugging)
#endif

// WR: f2py flavour; ST: accumulate sigma's moments.  Returns this THREAD's shifted sums in `mine` and whether this
// thread's wave saw a word of the planes change.
template <typename T, int SPT, bool WR, bool ST>
__device__ __forceinline__ bool sb_scan_pass(const DiagJob<T> &job, Moments &mine, unsigned kernarg_job_off = 0) {
    constexpr bool wrapper = WR, do_stats = ST;
    constexpr int SCAN_NT = SB_STATS_NT;
    const Geo g = job.g;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int NWV = SCAN_NT / SB_WAVE;              // waves per workgroup; SPT segments per trip
    const unsigned nseg = (unsigned)g.nyh * (unsigned)g.nw;
    // wave w of the grid takes segments w, w + W, w + 2W, ... (W = waves in the grid): every
    // wave gets floor or ceil of nseg/W segments, and neighbouring waves read neighbouring memory
    const unsigned nwaves = gridDim.x * NWV;
    const size_t pl = (size_t)g.nx * g.ny;
    if (blockIdx.x == 0 && threadIdx.x == 0 && job.ticket) *job.ticket = 0;   // spare device word, zeroed every call
    // a copy of the job in device memory for the kernels that take only its hot part by value (the job lies
    // kernarg_job_off bytes into this kernel's argument segment)
    if (blockIdx.x == 0 && job.self && threadIdx.x < sizeof(DiagJob<T>) / 4)
        ((unsigned *)job.self)[threadIdx.x] =
            ((const __attribute__((address_space(4))) unsigned *)__builtin_amdgcn_kernarg_segment_ptr())[kernarg_job_off / 4 + threadIdx.x];
    const double c = do_stats ? (double)job.sigma[(size_t)g.h * g.nxh + g.h] : 0.0;
    double s1 = 0.0, s2 = 0.0, mn = 1.0e308, mx = -1.0e308;
    int cnt = 0;

    // (row, word) of the wave's first segment and of the stride, kept wave-uniform: the
    // segment walk then needs no division (this kernel is issue-bound, not byte-bound)
    const unsigned w0 = __builtin_amdgcn_readfirstlane(blockIdx.x * NWV + wv);
    const unsigned unw = (unsigned)g.nw;
    unsigned Yc = w0 / unw, Xc = w0 - Yc * unw;
    const unsigned dY = nwaves / unw, dX = nwaves - dY * unw;
    const unsigned nxh = (unsigned)g.nxh, unx = (unsigned)g.nx;

    // One trip = SPT segments.  The loads of the NEXT trip are issued before this trip's values are
    // used, unconditionally and from clamped addresses: a load under a branch is waited for inside the
    // branch (one round trip per segment), and a load issued behind this trip's stores would make the
    // next wait sit out those stores as well (loads and stores share the in-order vmcnt counter).
    struct Trip {
        T sg[SPT], mk[SPT], ws[SPT], wd[SPT];
        uint64_t old;                                        // lanes 2q, 2q + 1: the band and the land-side word the call before
                                                             // left for segment q of the trip (one load for all of them)
        unsigned Y[SPT], W[SPT];
    };
    bool plane_changed = false;
    static_assert(2 * SPT <= SB_WAVE, "two lanes per segment of a trip");
    auto issue = [&](unsigned s0, Trip &t) {
        {
            const unsigned seg = s0 + (unsigned)(lane >> 1) * nwaves;
            const uint64_t *plane = (lane & 1) ? job.clsbits : job.bandbits;
            t.old = plane[(lane < 2 * SPT && seg < nseg) ? seg : 0u];     // (measured: 0.2-0.6 us of k_scan's 24, A/B on one box)
        }
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const unsigned seg = s0 + q * nwaves;
            t.Y[q] = Yc;
            t.W[q] = Xc;
            const int X = (int)(Xc * 64u) + lane;
            const int xi = X - g.h, yi = (int)Yc - g.h;
            const bool in = seg < nseg && X < g.nxh;
            const bool interior = in && xi >= 0 && xi < g.nx && yi >= 0 && yi < g.ny;
            const unsigned idx = in ? Yc * nxh + (unsigned)X : 0u;
            t.mk[q] = job.mask[job.mask_off + (in ? Yc * (unsigned)job.mask_ld + (unsigned)X : 0u)];
            t.sg[q] = T(0); t.ws[q] = T(0); t.wd[q] = T(0);
            if (do_stats) t.sg[q] = job.sigma[idx];                      // wave-uniform condition
            if (wrapper) {                                               // wave-uniform condition
                const unsigned o = (interior && yi < g.rows) ? (unsigned)yi * unx + (unsigned)xi : 0u;
                t.ws[q] = job.ws[o];
                t.wd[q] = job.wd[o];
            }
            Yc += dY; Xc += dX;
            if (Xc >= unw) { Xc -= unw; Yc += 1; }
        }
    };
    auto process = [&](unsigned s0, const Trip &t) {
        uint64_t now = t.old;                                // (lanes 2q, 2q + 1: the words this call writes for segment q)
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const unsigned seg = s0 + q * nwaves;
            if (seg >= nseg) break;                              // wave-uniform
            const int X = (int)(t.W[q] * 64u) + lane, xi = X - g.h, yi = (int)t.Y[q] - g.h;
            const bool in = X < g.nxh;
            const bool interior = in && xi >= 0 && xi < g.nx && yi >= 0 && yi < g.ny;
            if (do_stats && interior) {
                const double x = (double)t.sg[q], d = x - c;
                s1 += d;
                s2 = __builtin_fma(d, d, s2);
                mn = fmin(mn, x);
                mx = fmax(mx, x);
                ++cnt;
            }
            const bool cls = in && (t.mk[q] >= T(0));
            const bool band = interior && yi < g.rows && !(fabs(t.mk[q]) > job.maxdist);
            const uint64_t wc = __ballot(cls);
            const uint64_t wb = __ballot(band);
            // (lanes 2q, 2q + 1 hold the words the call before left: a word is written only where it differs -- on a coast
            // that stands, no store is issued into the planes at all)
            if ((lane >> 1) == q) {
                now = (lane & 1) ? wc : wb;
                if (SB_EAGER_PLANES || now != t.old) ((lane & 1) ? job.clsbits : job.bandbits)[seg] = now;
            }
            if (wb) {                                            // wave-uniform
                // the tile columns the segment's band cells fall in (two of 32 cells, three when the ghost
                // width is not a multiple of the tile width): lane j looks at the bits of column tA + j
                // in the ballot and raises that tile's flag -- one exec-masked store, no further ballots
                const int txs = job.thc_txs, tw = 1 << txs;
                const int xi0 = (int)(t.W[q] * 64u) - g.h;       // interior longitude of lane 0 (may be negative)
                const int tA = xi0 >> txs;
                int lo = ((tA + lane) << txs) - xi0, hi = lo + tw;
                lo = lo < 0 ? 0 : lo;
                hi = hi > 64 ? 64 : hi;
                if (lane <= (64 >> txs) && lo < hi) {
                    const uint64_t m = (hi - lo == 64 ? ~0ull : ((1ull << (hi - lo)) - 1ull)) << lo;
                    if (wb & m) job.tile_nnmax[(tA + lane) * job.tile_sx + (yi / job.thc_ty) * job.tile_sy + job.tile_off] = 1;   // benign duplicates
                }
            }
            if (interior && yi < g.rows && !band) {
                const unsigned o = (unsigned)yi * unx + (unsigned)xi;
                if (!wrapper) job.sb_con[o] = job.fill;     // (non-temporal stores measured the same: 118.0 against 118.6 us per call)
                else {
                    job.out[o] = job.fill;
                    job.out[2 * pl + o] = t.ws[q];
                    job.out[3 * pl + o] = t.wd[q];
                }
            }
        }
        // the strip kernel's plan stands while both planes do (where the band cells lie; the radius of every cell's
        // window, the land-side cells in it and its own class)
        plane_changed |= now != t.old;
    };
    // two register sets, alternating: the loads of trip n+1 are issued before trip n is used
    const unsigned step = SPT * nwaves;
    Trip ta, tb;
    unsigned s0 = w0;
    if (s0 < nseg) issue(s0, ta);
    while (s0 < nseg) {
        if (s0 + step < nseg) issue(s0 + step, tb);              // wave-uniform
        process(s0, ta);
        s0 += step;
        if (s0 >= nseg) break;
        if (s0 + step < nseg) issue(s0 + step, ta);
        process(s0, tb);
        s0 += step;
    }
    mine.n = (double)cnt; mine.mean = s1; mine.m2 = s2; mine.mn = mn; mine.mx = mx;
    return plane_changed;
}
