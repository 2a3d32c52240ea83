// probe_cumask.hip -- can two kernels be given disjoint sets of compute units?  (diagnostic, not product)
//
// hipExtStreamCreateWithCUMask creates a stream whose kernels run only on the CUs of a bit mask.  This program
//  (1) launches a census kernel on masked streams and prints where its workgroups ran (XCC id, SE, CU from the
//      hardware id registers), to learn how mask bits map to compute units on an 8-XCD part;
//  (2) times a k_wind-shaped gather (probe_gather.hip, k_wind-like case) on masks of different sizes: how many
//      compute units does the HBM-bound kernel need?
//  (3) runs a k_thc3-shaped VALU/LDS kernel and the gather on disjoint masks at the same time.
//
//   hipcc -O3 --offload-arch=gfx950 tools/probe_cumask.hip -o tools/_build/probe_cumask && tools/_build/probe_cumask
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_census(unsigned *out, int spin) {
    // hwreg(HW_REG_HW_ID = 4, 0, 32), hwreg(HW_REG_XCC_ID = 20, 0, 32)
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    // stay resident for a while so that the whole grid spreads over the allowed CUs
    long long t0 = clock64();
    while (clock64() - t0 < spin) { }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

template <int UN>
__global__ __launch_bounds__(256) void k_gather(const double *__restrict__ p, size_t plane, int nz, int nseg, int act,
                                                double *__restrict__ out) {
    const int gw = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
    const int lane = threadIdx.x & 63;
    for (int s = gw; s < nseg; s += nw) {
        const size_t cell = (size_t)s * 287 + (s * 7) % 16 + lane;
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

// VALU/LDS stand-in: `tiles` rounds of work per workgroup, dealt from a ticket so that any number of workgroups finishes the same total
__global__ __launch_bounds__(512) void k_valu(int total_tiles, int inner, int *ticket, double *__restrict__ out) {
    extern __shared__ double sm[];
    __shared__ int s_t;
    const int tid = threadIdx.x;
    double a[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) a[i] = 1.0 + 1e-9 * (tid + i);
    for (int i = tid; i < 8192; i += 512) sm[i] = 0.5 + 1e-7 * i;
    __syncthreads();
    for (;;) {
        if (tid == 0) s_t = atomicAdd(ticket, 1);
        __syncthreads();
        const int t = s_t;
        __syncthreads();
        if (t >= total_tiles) break;
        for (int j = 0; j < inner; ++j) {
            const double x = sm[(tid * 9 + j * 513 + t) & 8191];
#pragma unroll
            for (int i = 0; i < 12; ++i) a[i] = __builtin_fma(a[i], 0.999999, x * 1e-6);
            if ((j & 15) == 15) __syncthreads();
        }
    }
    asm volatile("" ::: "v250");
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
    unsigned *cens;
    CHK(hipMalloc(&cens, 4096 * 8));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    auto make_stream = [&](const std::vector<unsigned> &mask) {
        hipStream_t st;
        CHK(hipExtStreamCreateWithCUMask(&st, (unsigned)mask.size(), mask.data()));
        return st;
    };
    auto census = [&](const char *what, hipStream_t st) {
        CHK(hipMemset(cens, 0xff, 4096 * 8));
        hipLaunchKernelGGL(k_census, dim3(2048), dim3(256), 0, st, cens, 200000);
        CHK(hipStreamSynchronize(st));
        std::vector<unsigned> h(4096);
        CHK(hipMemcpy(h.data(), cens, 4096 * 4, hipMemcpyDeviceToHost));
        std::map<unsigned, int> per_xcc, cus;
        for (int b = 0; b < 2048; ++b) {
            const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            per_xcc[xcc]++;
            cus[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
        }
        printf("%s: %zu distinct (xcc,se,sh,cu); workgroups per XCC:", what, cus.size());
        for (auto &kv : per_xcc) printf(" %u:%d", kv.first, kv.second);
        printf("\n   CUs used per XCC:");
        std::map<unsigned, int> cu_per_xcc;
        for (auto &kv : cus) cu_per_xcc[kv.first >> 12]++;
        for (auto &kv : cu_per_xcc) printf(" %u:%d", kv.first, kv.second);
        printf("\n");
    };
    // (1) census: full, low half of the bits, every second bit, first 32 bits, bits 32..63
    const int nwords = (ncu + 31) / 32;
    std::vector<unsigned> full(nwords, 0xffffffffu), low(nwords, 0), even(nwords, 0x55555555u), first32(nwords, 0), second32(nwords, 0),
        q1(nwords, 0x0000ffffu), q3(nwords, 0xffff0000u), three4(nwords, 0xfffffff0u & 0xffffffffu);
    for (int w = 0; w < nwords / 2; ++w) low[w] = 0xffffffffu;
    first32[0] = 0xffffffffu;
    second32[1] = 0xffffffffu;
    hipStream_t s_full = make_stream(full), s_low = make_stream(low), s_even = make_stream(even), s_f32 = make_stream(first32),
                s_s32 = make_stream(second32), s_q1 = make_stream(q1), s_q3 = make_stream(q3);
    census("full mask          ", s_full);
    census("low half of bits   ", s_low);
    census("every second bit   ", s_even);
    census("bits 0..31         ", s_f32);
    census("bits 32..63        ", s_s32);
    census("low 16 of each word", s_q1);
    census("high 16 of each word", s_q3);

    // (2), (3)
    const int nz = 56, nseg = 17085, act = 28;
    const size_t plane = (size_t)2560 * 1920;
    double *p, *out;
    int *ticket;
    CHK(hipMalloc(&p, plane * nz * sizeof(double)));
    CHK(hipMemset(p, 0, plane * nz * sizeof(double)));
    CHK(hipMalloc(&out, plane * sizeof(double) + (1 << 20)));
    CHK(hipMalloc(&ticket, 4));
    CHK(hipFuncSetAttribute((const void *)k_valu, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    auto timeit = [&](const char *what, hipStream_t ts, auto fn) {
        std::vector<float> t;
        for (int r = 0; r < 9; ++r) {
            CHK(hipDeviceSynchronize());
            CHK(hipMemset(ticket, 0, 4));
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0, ts));
            fn();
            CHK(hipEventRecord(e1, ts));
            CHK(hipEventSynchronize(e1));
            CHK(hipDeviceSynchronize());
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            t.push_back(ms);
        }
        printf("  %-72s %.1f us\n", what, median(t) * 1e3);
    };
    struct MaskCase { const char *name; std::vector<unsigned> mask; int cus; };
    std::vector<MaskCase> mc;
    mc.push_back({"all bits", full, ncu});
    mc.push_back({"high 16 of each word (1/2)", q3, ncu / 2});
    { std::vector<unsigned> m(nwords, 0xff000000u); mc.push_back({"high 8 of each word (1/4)", m, ncu / 4}); }
    { std::vector<unsigned> m(nwords, 0xfff00000u); mc.push_back({"high 12 of each word (3/8)", m, ncu * 3 / 8}); }
    for (auto &c : mc) {
        hipStream_t st = make_stream(c.mask);
        for (int wgs : {2, 4, 8}) {
            char nm[160];
            snprintf(nm, sizeof nm, "gather alone on %s, %d WG per allowed CU", c.name, wgs);
            timeit(nm, st, [&] { hipLaunchKernelGGL(k_gather<8>, dim3(c.cus * wgs), dim3(256), 0, st, p, plane, nz, nseg, act, out); });
        }
        CHK(hipStreamDestroy(st));
    }
    // V: 670 "tiles" of work; calibrate inner so that V alone on all CUs takes ~45 us
    const int tiles = 670, inner = 150;
    {
        hipStream_t st = make_stream(full);
        timeit("V alone, all CUs (256 workgroups)", st, [&] { hipLaunchKernelGGL(k_valu, dim3(ncu), dim3(512), 100 * 1024, st, tiles, inner, ticket, out + plane); });
        CHK(hipStreamDestroy(st));
    }
    hipEvent_t fork, join;
    CHK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CHK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    struct Split { const char *name; unsigned vmask, gmask; int vcus, gcus; };
    const Split splits[] = {
        {"V on low 16 bits of each word, gather on high 16", 0x0000ffffu, 0xffff0000u, ncu / 2, ncu / 2},
        {"V on low 20 bits, gather on high 12", 0x000fffffu, 0xfff00000u, ncu * 5 / 8, ncu * 3 / 8},
        {"V on low 24 bits, gather on high 8", 0x00ffffffu, 0xff000000u, ncu * 3 / 4, ncu / 4},
    };
    for (const Split &sp : splits) {
        std::vector<unsigned> vm(nwords, sp.vmask), gm(nwords, sp.gmask);
        hipStream_t sv = make_stream(vm), sg = make_stream(gm);
        char nm[200];
        snprintf(nm, sizeof nm, "V alone on its share (%d CUs)", sp.vcus);
        timeit(nm, sv, [&] { hipLaunchKernelGGL(k_valu, dim3(sp.vcus), dim3(512), 100 * 1024, sv, tiles, inner, ticket, out + plane); });
        for (int wgs : {4, 8}) {
            snprintf(nm, sizeof nm, "%s (%d WG/CU): both at once", sp.name, wgs);
            timeit(nm, sg, [&] {
                CHK(hipEventRecord(fork, sg));
                CHK(hipStreamWaitEvent(sv, fork, 0));
                hipLaunchKernelGGL(k_valu, dim3(sp.vcus), dim3(512), 100 * 1024, sv, tiles, inner, ticket, out + plane);
                hipLaunchKernelGGL(k_gather<8>, dim3(sp.gcus * wgs), dim3(256), 0, sg, p, plane, nz, nseg, act, out);
                CHK(hipEventRecord(join, sv));
                CHK(hipStreamWaitEvent(sg, join, 0));
            });
        }
        CHK(hipStreamDestroy(sv));
        CHK(hipStreamDestroy(sg));
    }
    printf("done\n");
    return 0;
}
