// probe_ramp.hip -- how long does the dispatcher take to get a grid's waves going?  Every wave leaves the 100 MHz wall
// clock at its first instruction; the spread of those marks over the grid is the start ramp a kernel pays before its
// last wave has begun.  Varied: threads per workgroup, workgroups, registers per lane, LDS per workgroup.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_ramp.hip -o /tmp/probe_ramp && /tmp/probe_ramp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int NT, int BIGV>
__global__ __launch_bounds__(NT) void k_mark(long long *marks, int spin) {
    extern __shared__ int dyn[];
    const long long t = wall_clock64();
    if (BIGV) asm volatile("v_mov_b32 v120, 0" ::: "v120");      // (forces an allocation of more than 120 registers per lane)
    const int w = (blockIdx.x * NT + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) marks[w] = t;
    if (spin) {                                                   // keep the wave resident for a while (a real kernel's waves do not leave at once)
        const long long t1 = t + spin;
        while (wall_clock64() < t1) __builtin_amdgcn_s_sleep(8);
        if (threadIdx.x == 0 && dyn[0] == 12345) marks[w] = 0;
    }
}

template <int NT, int BIGV>
static void run(const char *what, int nwg, size_t lds, int spin, long long *d) {
    const int nw = nwg * NT / 64;
    std::vector<long long> h(nw);
    double best = 1e9, bestk = 0;
    hipFuncSetAttribute((const void *)k_mark<NT, BIGV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 6; ++rep) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_mark<NT, BIGV>), dim3(nwg), dim3(NT), lds, 0, d, spin);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), d, nw * sizeof(long long), hipMemcpyDeviceToHost);
        const auto mm = std::minmax_element(h.begin(), h.end());
        const double ramp = (*mm.second - *mm.first) / 100.0;
        if (rep > 0 && ramp < best) { best = ramp; bestk = ms * 1e3; }
    }
    printf("%-58s %5d waves: last wave starts %6.2f us after the first (kernel %6.1f us)\n", what, nw, best, bestk);
}

int main() {
    long long *d;
    hipMalloc(&d, 1 << 22);
    for (int spin : {0, 2000}) {
        printf("---- waves stay resident for %d us\n", spin / 100);
        run<1024, 1>("256 WG x 1024 thr, >120 VGPR, 160 KB LDS", 256, 160 * 1024, spin, d);
        run<1024, 1>("256 WG x 1024 thr, >120 VGPR, no LDS", 256, 0, spin, d);
        run<1024, 0>("256 WG x 1024 thr, few VGPR, no LDS", 256, 0, spin, d);
        run<1024, 0>("256 WG x 1024 thr, few VGPR, 160 KB LDS", 256, 160 * 1024, spin, d);
        run<512, 0>("256 WG x 512 thr, few VGPR, no LDS", 256, 0, spin, d);
        run<512, 0>("512 WG x 512 thr, few VGPR, no LDS", 512, 0, spin, d);
        run<256, 0>("256 WG x 256 thr, few VGPR, no LDS", 256, 0, spin, d);
        run<256, 0>("1024 WG x 256 thr, few VGPR, no LDS", 1024, 0, spin, d);
        run<256, 0>("2048 WG x 256 thr, few VGPR, no LDS", 2048, 0, spin, d);
        run<128, 0>("2048 WG x 128 thr, few VGPR, no LDS", 2048, 0, spin, d);
        run<64, 0>("4096 WG x 64 thr, few VGPR, no LDS", 4096, 0, spin, d);
    }
    return 0;
}
