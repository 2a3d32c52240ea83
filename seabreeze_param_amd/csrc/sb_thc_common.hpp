// sb_thc_common.hpp -- helpers shared by the two contrast kernels (k_strip: marching strips, LDS halo <= 16;
// k_thc3: LDS tiles, halos of 24 and 32 cells).   ref: generic/sea_breeze_diag.f90:188-216
#pragma once
#include "sb_device.hpp"
#include "sb_launch.hpp"

// ------------------------------------------------------------------------------------
// Global-memory search for cells whose window outgrows the LDS tile (rare).  Rings are
// accumulated from the centre outwards; the first radius >= 1 at which the square holds
// both classes is the reference's final nn.  cap bounds the radius: the reference loop has
// none and never returns on a one-class grid (SURVEY.md §7 "Hard parts").
// ------------------------------------------------------------------------------------
// t0 of one cell of the ghost-celled frame: from the workspace (f2py flavour) or derived on the spot
template <typename T>
__device__ __forceinline__ T cell_t0(const DiagJob<T> &job, size_t idx, T sd, T rr) {
    if (!job.t0_fly) return job.t0[idx];
    return sb_t0<T>(job.theta[idx], job.z[idx], job.sigma[idx], sd, rr);
}

template <typename T>
__device__ __forceinline__ T contrast_global(const DiagJob<T> &job, int x, int y, int cap, T sd, T rr, int &nn_used,
                                          bool &one_class) {
    const Geo g = job.g;
    int X, Y;
    bool has_l = false, has_s = false;
    if (sb_map_cell(g, x, y, X, Y)) {
        if (sb_bit(job.clsbits, g.nw, X, Y)) has_l = true; else has_s = true;
    }
    int nn = 0;
    bool found = false;
    while (nn < cap) {
        ++nn;
        for (int e = -nn; e <= nn; ++e) {
            const int xs[4] = {x + e, x + e, x - nn, x + nn};
            const int ys[4] = {y - nn, y + nn, y + e, y + e};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (sb_map_cell(g, xs[q], ys[q], X, Y)) {
                    if (sb_bit(job.clsbits, g.nw, X, Y)) has_l = true; else has_s = true;
                }
        }
        if (has_l && has_s) { found = true; break; }
    }
    nn_used = nn;
    one_class = !found;
    sb_map_cell(g, x, y, X, Y);
    const double c0 = (double)cell_t0(job, (size_t)Y * g.nxh + X, sd, rr);
    double sl = 0.0, ss = 0.0, nl = 0.0, ns = 0.0;
    for (int yy = y - nn; yy <= y + nn; ++yy)
        for (int xx = x - nn; xx <= x + nn; ++xx) {
            if (!sb_map_cell(g, xx, yy, X, Y)) continue;
            const double d = (double)cell_t0(job, (size_t)Y * g.nxh + X, sd, rr) - c0;
            if (sb_bit(job.clsbits, g.nw, X, Y)) { sl += d; nl += 1.0; } else { ss += d; ns += 1.0; }
        }
    return (T)(sl / nl - ss / ns);               // 0/0 -> NaN when a class is missing
}

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- buffer loads: a 128-bit resource descriptor per field (wave-uniform base + size), a 32-bit byte offset per
// lane and a scalar byte offset per row.  One instruction per load, no 64-bit address arithmetic: the staging of a
// tile is issue-bound, and a row's offset is the same for all 64 lanes of the wave that owns it.
typedef unsigned int sb_u2 __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ T sb_buf_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff);
template <>
__device__ __forceinline__ double sb_buf_ld<double>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const sb_u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}
template <>
__device__ __forceinline__ float sb_buf_ld<float>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sb_make_rsrc(const void *p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)(bytes > 0xfffffff0ull ? 0xfffffff0ull : bytes), 0x00020000);
}

// reciprocal of a small positive integer held in a double: v_rcp_f64 and two Newton steps (within an ulp of 1/n)
__device__ __forceinline__ double sb_inv(double n) {
    double q = __builtin_amdgcn_rcp(n);
    q = __builtin_fma(q, __builtin_fma(-n, q, 1.0), q);
    q = __builtin_fma(q, __builtin_fma(-n, q, 1.0), q);
    return q;
}

// largest value of a wave, valid in lane 63 (DPP, no LDS crossbar)
__device__ __forceinline__ int sb_wave_max_to_last(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));
    return v;                                    // values are >= 0, lanes without a source contribute 0
}

// exclusive prefix of one int per thread over an NT-thread workgroup; total in `total`.  Two barriers.
template <int NT>
__device__ __forceinline__ int thc_block_excl_scan(int v, int *s_w, int &total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int incl = sb_wave_scan_add(v);
    __syncthreads();
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    const int wt = lane < NT / SB_WAVE ? s_w[lane] : 0;
    const int wincl = sb_wave_scan_add(wt);
    total = __shfl(wincl, NT / SB_WAVE - 1);
    return incl - v + __shfl(wincl, wv) - __shfl(wt, wv);
}

template <bool FLY>
struct ThcBufs {
    __amdgpu_buffer_rsrc_t th, zz, sg, cls;       // theta (FLY) or t0; z; sigma; land-side plane
};
