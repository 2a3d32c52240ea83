// probe_overlap.hip -- two hardware questions behind DESIGN.md's plan for the diag call (diagnostic, not product):
//
//  (1) fetch granule: does a load that touches only one 64-byte half of a 128-byte line cost the
//      HBM traffic of the half or of the whole line?  (k_wind's p-column walk ends every run of band
//      cells inside a line: 1.30x the algorithmic bytes with 128-byte granules, 1.14x with 64-byte ones.)
//  (2) co-residency: how much of a VALU/LDS-bound persistent kernel (the shape of k_thc2: one 512-thread
//      workgroup per CU, ~100 KB LDS, <= 168 registers) hides under an HBM-bound gather (the shape of
//      k_wind: 256-thread workgroups walking 56 planes) when the two run on two streams?
//
//   hipcc -O3 --offload-arch=gfx950 tools/probe_overlap.hip -o /tmp/probe_overlap && /tmp/probe_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- (1) fetch granule ---------------------------------------------------------------------------
// mode 0: every lane reads 8 bytes, lanes contiguous (512 B per wave-load: 4 whole lines)
// mode 1: lanes read the FIRST 64 bytes of 8 consecutive lines (8 lanes per line)
// mode 2: lanes read the first 64 bytes of every second line pair... (= mode 1 with stride 256: control for DRAM page effects)
// mode 3: 16 lanes per line over 4 lines but only lines 0,2,4,6 of 8 (whole lines, half of them skipped)
__global__ __launch_bounds__(256) void k_fetch(const double *__restrict__ buf, size_t nlines, int mode, double *out) {
    const size_t gw = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * 256) >> 6;
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
    // a wave-step covers 8 lines (1 KB of address space) in modes 1..3, 4 lines in mode 0
    const size_t lines_per_step = mode == 0 ? 4 : 8;
    for (size_t s = gw * 8; (s + 8) * lines_per_step <= nlines; s += nw * 8) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const size_t base = (s + q) * lines_per_step * 16;      // doubles
            size_t idx;
            if (mode == 0) idx = base + lane;
            else if (mode == 1) idx = base + (lane >> 3) * 16 + (lane & 7);
            else if (mode == 2) idx = base + (lane >> 3) * 16 + 8 + (lane & 7);
            else idx = base + (lane >> 4) * 32 + (lane & 15);
            v[q] = __builtin_nontemporal_load(buf + idx);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += v[q];
    }
    if (acc == 12345.678) out[0] = acc;
}

// ---- (2) co-residency ----------------------------------------------------------------------------
// M: k_wind-shaped gather.  `nseg` segments of 64 cells; a wave takes segments gw, gw+W, ...; its first
// `act` lanes walk nz planes (stride `plane` doubles) with UN loads in flight.
template <int UN>
__global__ __launch_bounds__(256) void k_gather(const double *__restrict__ p, size_t plane, int nz, int nseg, int act,
                                                double *__restrict__ out) {
    const int gw = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
    const int lane = threadIdx.x & 63;
    for (int s = gw; s < nseg; s += nw) {
        // segments spread over the plane like band segments: every 4th..5th segment of a row
        const size_t cell = (size_t)s * 287 + lane;
        if (lane < act) {
            double best = 1e300;
            int lev = 0;
            for (int k0 = 0; k0 < nz; k0 += UN) {
                double d[UN];
#pragma unroll
                for (int q = 0; q < UN; ++q) d[q] = __builtin_nontemporal_load(p + cell + (size_t)(k0 + q < nz ? k0 + q : nz - 1) * plane);
#pragma unroll
                for (int q = 0; q < UN; ++q) { const double a = fabs(d[q] - 70000.0); if (a < best) { best = a; lev = k0 + q; } }
            }
            out[cell] = best + lev;
        }
    }
}

// V: k_thc2-shaped persistent VALU/LDS kernel: one 512-thread workgroup per CU, `lds_kb` of LDS, `iters`
// rounds of fp64 work on LDS-resident data separated by barriers.
template <int MINW>
__global__ __launch_bounds__(512, MINW) void k_valu(int iters, int inner, double *__restrict__ out) {
    extern __shared__ double sm[];
    const int tid = threadIdx.x;
    double a[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) a[i] = 1.0 + 1e-9 * (tid + i);
    for (int i = tid; i < 8192; i += 512) sm[i] = 0.5 + 1e-7 * i;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        for (int j = 0; j < inner; ++j) {
            const double x = sm[(tid * 9 + j * 513 + it) & 8191];
#pragma unroll
            for (int i = 0; i < 12; ++i) a[i] = __builtin_fma(a[i], 0.999999, x * 1e-6);
        }
        __syncthreads();
        sm[(tid + it * 7) & 8191] = a[it % 12 == 0 ? 0 : 1];
        __syncthreads();
    }
    // hold the register footprint of the kernel this stands in for: 2 waves per SIMD -> up to 256 registers,
    // 3 -> 168, 4 -> 128 (the allocation, not the use, is what decides co-residency)
    if (MINW == 2) asm volatile("" ::: "v250");
    else if (MINW == 3) asm volatile("" ::: "v166");
    else asm volatile("" ::: "v126");
    double s = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += a[i];
    if (s == 42.4242) out[blockIdx.x] = s;
}

static double median(std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    CHK(hipSetDevice(0));
    hipDeviceProp_t pr;
    CHK(hipGetDeviceProperties(&pr, 0));
    const int ncu = pr.multiProcessorCount;
    printf("device %s, %d CUs\n", pr.gcnArchName, ncu);
    hipStream_t s1, s2, shi;
    CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    int plo, phi;
    CHK(hipDeviceGetStreamPriorityRange(&plo, &phi));
    CHK(hipStreamCreateWithPriority(&shi, hipStreamNonBlocking, phi));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    double *out;
    CHK(hipMalloc(&out, 64 << 20));

    // ---------------- (1) ----------------
    {
        const size_t bytes = (size_t)2 << 30, nlines = bytes / 128;
        double *buf;
        CHK(hipMalloc(&buf, bytes));
        CHK(hipMemset(buf, 0, bytes));
        const char *names[4] = {"whole lines, contiguous              ", "first 64 B of every line             ",
                                "second 64 B of every line            ", "whole lines, every second one skipped"};
        for (int mode = 0; mode < 4; ++mode) {
            std::vector<float> t;
            for (int r = 0; r < 7; ++r) {
                CHK(hipEventRecord(e0, s1));
                hipLaunchKernelGGL(k_fetch, dim3(ncu * 8), dim3(256), 0, s1, buf, nlines, mode, out);
                CHK(hipEventRecord(e1, s1));
                CHK(hipEventSynchronize(e1));
                float ms;
                CHK(hipEventElapsedTime(&ms, e0, e1));
                t.push_back(ms);
            }
            const double ms = median(t);
            const double touched = mode == 0 ? (double)bytes : (double)bytes / 2;
            printf("fetch mode %d (%s): %.3f ms, %.0f GB/s of touched bytes, %.0f GB/s if whole lines are fetched\n", mode,
                   names[mode], ms, touched / ms / 1e6, (mode == 3 ? (double)bytes / 2 : (double)bytes) / ms / 1e6);
        }
        CHK(hipFree(buf));
    }

    // ---------------- (2) ----------------
    {
        const int nz = 56, nseg = 17085, act = 28;
        const size_t plane = (size_t)2560 * 1920;
        double *p;
        CHK(hipMalloc(&p, plane * nz * sizeof(double)));
        CHK(hipMemset(p, 0, plane * nz * sizeof(double)));
        auto runM = [&](int wgs_per_cu, int un, hipStream_t st) {
            const dim3 g(ncu * wgs_per_cu), b(256);
            if (un == 8) hipLaunchKernelGGL(k_gather<8>, g, b, 0, st, p, plane, nz, nseg, act, out);
            else if (un == 14) hipLaunchKernelGGL(k_gather<14>, g, b, 0, st, p, plane, nz, nseg, act, out);
            else hipLaunchKernelGGL(k_gather<28>, g, b, 0, st, p, plane, nz, nseg, act, out);
        };
        auto runV = [&](int minw, int iters, int inner, int lds_kb, hipStream_t st) {
            const dim3 g(ncu), b(512);
            if (minw == 2) hipLaunchKernelGGL(k_valu<2>, g, b, lds_kb * 1024, st, iters, inner, out + (32 << 17));
            else if (minw == 3) hipLaunchKernelGGL(k_valu<3>, g, b, lds_kb * 1024, st, iters, inner, out + (32 << 17));
            else hipLaunchKernelGGL(k_valu<4>, g, b, lds_kb * 1024, st, iters, inner, out + (32 << 17));
        };
        CHK(hipFuncSetAttribute((const void *)k_valu<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        CHK(hipFuncSetAttribute((const void *)k_valu<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        CHK(hipFuncSetAttribute((const void *)k_valu<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        auto timeit = [&](const char *what, auto fn) {
            std::vector<float> t;
            for (int r = 0; r < 9; ++r) {
                CHK(hipDeviceSynchronize());
                CHK(hipEventRecord(e0, s1));
                fn();
                CHK(hipEventRecord(e1, s1));
                CHK(hipEventSynchronize(e1));
                CHK(hipDeviceSynchronize());
                float ms;
                CHK(hipEventElapsedTime(&ms, e0, e1));
                t.push_back(ms);
            }
            printf("  %-64s %.1f us\n", what, median(t) * 1e3);
        };
        hipEvent_t fork, join;
        CHK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        CHK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
        printf("gather: %d segments x %d lanes x %d planes = %.0f MB algorithmic\n", nseg, act, nz, (double)nseg * act * nz * 8 / 1e6);
        for (int un : {8, 14, 28})
            for (int w : {2, 3, 4, 5, 6, 8}) {
                char nm[128];
                snprintf(nm, sizeof nm, "M alone: %d WG/CU, %d loads in flight", w, un);
                timeit(nm, [&] { runM(w, un, s1); });
            }
        const int iters = 40;
        for (int minw : {2, 3, 4}) {
            // calibrate V to ~25 us
            char nm[128];
            for (int inner : {40}) {
                snprintf(nm, sizeof nm, "V alone: launch_bounds(512,%d), 100 KB LDS, inner %d", minw, inner);
                timeit(nm, [&] { runV(minw, iters, inner, 100, s1); });
            }
            for (int un : {14, 28})
                for (int w : {3, 4, 5}) {
                    snprintf(nm, sizeof nm, "V(minw %d) on s2 first || M %d WG/CU un %d on s1", minw, w, un);
                    timeit(nm, [&] {
                        CHK(hipEventRecord(fork, s1));
                        CHK(hipStreamWaitEvent(s2, fork, 0));
                        runV(minw, iters, 40, 100, s2);
                        runM(w, un, s1);
                        CHK(hipEventRecord(join, s2));
                        CHK(hipStreamWaitEvent(s1, join, 0));
                    });
                    snprintf(nm, sizeof nm, "V(minw %d) on high-priority stream || M %d WG/CU un %d", minw, w, un);
                    timeit(nm, [&] {
                        CHK(hipEventRecord(fork, s1));
                        CHK(hipStreamWaitEvent(shi, fork, 0));
                        runV(minw, iters, 40, 100, shi);
                        runM(w, un, s1);
                        CHK(hipEventRecord(join, shi));
                        CHK(hipStreamWaitEvent(s1, join, 0));
                    });
                }
            snprintf(nm, sizeof nm, "V(minw %d) then M 5 WG/CU un 14, same stream (serial)", minw);
            timeit(nm, [&] { runV(minw, iters, 40, 100, s1); runM(5, 14, s1); });
        }
        CHK(hipFree(p));
    }
    printf("done\n");
    return 0;
}
