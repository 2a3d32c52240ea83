// probe_gather.hip -- what bounds a k_wind-shaped gather on MI355X?  (diagnostic, not product)
//
// k_wind walks, for every coastal-band cell, the nz-level pressure column of a caller-owned (lon, lat, lev)
// array: runs of band cells along longitude (28 cells = 224 bytes on average inside a 64-cell segment) times
// nz planes 39 MB apart.  This program times that access pattern with the run length, its alignment, the
// order of the walk and the loads in flight as parameters, next to a plain stream of the same bytes, and a
// k_scan-shaped stream (two arrays in, one out).
//
//   hipcc -O3 --offload-arch=gfx950 tools/probe_gather.hip -o tools/_build/probe_gather && tools/_build/probe_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// a wave takes segments gw, gw+NW, ...; segment s starts at cell s*stride + shift(s); its first `act` lanes walk
template <int UN, bool LEVEL_MAJOR>
__global__ __launch_bounds__(256) void k_gather(const double *__restrict__ p, size_t plane, int nz, int nseg, int act,
                                                int stride, int misalign, double *__restrict__ out) {
    const int gw = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
    const int lane = threadIdx.x & 63;
    auto cell_of = [&](int s) { return (size_t)s * stride + (misalign ? (s * 7) % 16 : 0) + lane; };
    if (!LEVEL_MAJOR) {
        for (int s = gw; s < nseg; s += nw) {
            const size_t cell = cell_of(s);
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
    } else {
        // the wave's (up to 4) segments advance through the levels together: at any time the chip touches few planes
        constexpr int MS = 4;
        for (int base = gw; base < nseg; base += MS * nw) {
            double best[MS];
            int lev[MS];
#pragma unroll
            for (int j = 0; j < MS; ++j) { best[j] = 1e300; lev[j] = 0; }
            for (int k0 = 0; k0 < nz; k0 += UN) {
#pragma unroll
                for (int j = 0; j < MS; ++j) {
                    const int s = base + j * nw;
                    if (s < nseg && lane < act) {
                        const size_t cell = cell_of(s);
                        double d[UN];
#pragma unroll
                        for (int q = 0; q < UN; ++q) d[q] = __builtin_nontemporal_load(p + cell + (size_t)(k0 + q < nz ? k0 + q : nz - 1) * plane);
#pragma unroll
                        for (int q = 0; q < UN; ++q) { const double a = fabs(d[q] - 70000.0); if (a < best[j]) { best[j] = a; lev[j] = k0 + q; } }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < MS; ++j) {
                const int s = base + j * nw;
                if (s < nseg && lane < act) out[cell_of(s)] = best[j] + lev[j];
            }
        }
    }
}

// a wave takes MS ADJACENT segments (neighbours along longitude: the same pages of every plane) and walks them together,
// UN levels of each in flight: per level one page of the plane serves MS loads of the wave
template <int UN, int MS>
__global__ __launch_bounds__(256) void k_gather_adj(const double *__restrict__ p, size_t plane, int nz, int nseg, int act,
                                                    int stride, int misalign, double *__restrict__ out) {
    const int gw = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
    const int lane = threadIdx.x & 63;
    auto cell_of = [&](int s) { return (size_t)s * stride + (misalign ? (s * 7) % 16 : 0) + lane; };
    for (int base = gw * MS; base < nseg; base += MS * nw) {
        double best[MS];
        int lev[MS];
        size_t cell[MS];
#pragma unroll
        for (int j = 0; j < MS; ++j) { best[j] = 1e300; lev[j] = 0; cell[j] = cell_of(base + j < nseg ? base + j : nseg - 1); }
        if (lane < act) {
            for (int k0 = 0; k0 < nz; k0 += UN) {
                double d[MS][UN];
#pragma unroll
                for (int q = 0; q < UN; ++q)
#pragma unroll
                    for (int j = 0; j < MS; ++j) d[j][q] = __builtin_nontemporal_load(p + cell[j] + (size_t)(k0 + q < nz ? k0 + q : nz - 1) * plane);
#pragma unroll
                for (int j = 0; j < MS; ++j)
#pragma unroll
                    for (int q = 0; q < UN; ++q) { const double a = fabs(d[j][q] - 70000.0); if (a < best[j]) { best[j] = a; lev[j] = k0 + q; } }
            }
#pragma unroll
            for (int j = 0; j < MS; ++j) if (base + j < nseg) out[cell[j]] = best[j] + lev[j];
        }
    }
}

// k_scan-shaped stream: read a and b, write c where a lane-dependent predicate holds (90 % of the cells)
__global__ __launch_bounds__(1024) void k_stream2in1out(const double *__restrict__ a, const double *__restrict__ b,
                                                        double *__restrict__ c, size_t n, int wr) {
    const size_t stride = (size_t)gridDim.x * 1024;
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += 4 * stride) {
        double x[4], y[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const size_t j = i + q * stride < n ? i + q * stride : i; x[q] = a[j]; y[q] = b[j]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc += x[q];
            if (wr && i + q * stride < n && !(fabs(y[q]) <= 180.0)) c[i + q * stride] = 0.0;
        }
    }
    if (acc == 1.2345) c[0] = acc;
}

static double median(std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    CHK(hipSetDevice(0));
    hipDeviceProp_t pr;
    CHK(hipGetDeviceProperties(&pr, 0));
    const int ncu = pr.multiProcessorCount;
    hipStream_t s1;
    CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int nz = 56;
    const size_t plane = (size_t)2560 * 1920;
    double *p, *out;
    CHK(hipMalloc(&p, plane * nz * sizeof(double)));
    CHK(hipMemset(p, 0, plane * nz * sizeof(double)));
    CHK(hipMalloc(&out, plane * sizeof(double)));
    auto timeit = [&](auto fn) {
        std::vector<float> t;
        for (int r = 0; r < 9; ++r) {
            CHK(hipEventRecord(e0, s1));
            fn();
            CHK(hipEventRecord(e1, s1));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            t.push_back(ms);
        }
        return median(t) * 1e3;
    };
    struct Case { const char *name; int nseg, act, stride, misalign; };
    // all cases move ~214 MB of algorithmic bytes except where noted
    const Case cases[] = {
        {"28 lanes, runs start anywhere (k_wind-like)      ", 17085, 28, 287, 1},
        {"28 lanes, runs start on a line boundary          ", 17085, 28, 272, 0},
        {"32 lanes = 2 whole lines                         ", 14950, 32, 320, 0},
        {"16 lanes = 1 whole line                          ", 29900, 16, 160, 0},
        {"64 lanes = 4 whole lines (512 B pieces)          ", 7475, 64, 640, 0},
        {"64 lanes, segments contiguous (stream by planes) ", 7475, 64, 64, 0},
    };
    for (const Case &c : cases) {
        const double alg = (double)c.nseg * c.act * nz * 8;
        // lines touched per piece
        double lines = 0;
        for (int s = 0; s < c.nseg; ++s) {
            const size_t b0 = ((size_t)s * c.stride + (c.misalign ? (s * 7) % 16 : 0)) * 8, b1 = b0 + c.act * 8 - 1;
            lines += (double)(b1 / 128 - b0 / 128 + 1);
        }
        const double fetched = lines * 128 * nz;
        printf("%s alg %.0f MB, lines %.0f MB\n", c.name, alg / 1e6, fetched / 1e6);
        for (int wgs : {4, 8}) {
            const dim3 g(ncu * wgs), b(256);
            const double t8 = timeit([&] { hipLaunchKernelGGL((k_gather<8, false>), g, b, 0, s1, p, plane, nz, c.nseg, c.act, c.stride, c.misalign, out); });
            const double t28 = timeit([&] { hipLaunchKernelGGL((k_gather<28, false>), g, b, 0, s1, p, plane, nz, c.nseg, c.act, c.stride, c.misalign, out); });
            const double t56 = timeit([&] { hipLaunchKernelGGL((k_gather<56, false>), g, b, 0, s1, p, plane, nz, c.nseg, c.act, c.stride, c.misalign, out); });
            const double l8 = timeit([&] { hipLaunchKernelGGL((k_gather<8, true>), g, b, 0, s1, p, plane, nz, c.nseg, c.act, c.stride, c.misalign, out); });
            const double l14 = timeit([&] { hipLaunchKernelGGL((k_gather<14, true>), g, b, 0, s1, p, plane, nz, c.nseg, c.act, c.stride, c.misalign, out); });
            printf("   %d WG/CU: cell-major un8 %.1f us (%.0f GB/s of lines)  un28 %.1f  un56 %.1f | level-major un8 %.1f (%.0f GB/s)  un14 %.1f\n",
                   wgs, t8, fetched / t8 / 1e3, t28, t56, l8, fetched / l8 / 1e3, l14);
            const double a24 = timeit([&] { hipLaunchKernelGGL((k_gather_adj<4, 2>), g, b, 0, s1, p, plane, nz, c.nseg, c.act, c.stride, c.misalign, out); });
            const double a44 = timeit([&] { hipLaunchKernelGGL((k_gather_adj<4, 4>), g, b, 0, s1, p, plane, nz, c.nseg, c.act, c.stride, c.misalign, out); });
            const double a28 = timeit([&] { hipLaunchKernelGGL((k_gather_adj<2, 8>), g, b, 0, s1, p, plane, nz, c.nseg, c.act, c.stride, c.misalign, out); });
            const double a82 = timeit([&] { hipLaunchKernelGGL((k_gather_adj<8, 2>), g, b, 0, s1, p, plane, nz, c.nseg, c.act, c.stride, c.misalign, out); });
            printf("              adjacent segments walked together (levels x segments in flight): 4x2 %.1f us  4x4 %.1f (%.0f GB/s of lines)  2x8 %.1f  8x2 %.1f\n",
                   a24, a44, fetched / a44 / 1e3, a28, a82);
        }
    }
    // k_scan-shaped stream
    {
        double *a, *b, *c;
        const size_t n = plane;
        CHK(hipMalloc(&a, n * 8 * 8));
        CHK(hipMalloc(&b, n * 8 * 8));
        CHK(hipMalloc(&c, n * 8 * 8));
        CHK(hipMemset(a, 0, n * 8 * 8));
        CHK(hipMemset(b, 0x7f, n * 8 * 8));      // |b| huge: every cell is written
        CHK(hipMemset(c, 0, n * 8 * 8));
        int rot = 0;
        for (int wr : {0, 1})
            for (int wgs : {1, 2}) {
                const double t = timeit([&] {
                    rot = (rot + 1) % 8;          // rotate through 8 copies: nothing is served from the caches
                    hipLaunchKernelGGL(k_stream2in1out, dim3(ncu * wgs), dim3(1024), 0, s1, a + rot * n, b + rot * n, c + rot * n, n, wr);
                });
                printf("stream 2 in%s, %d x 1024 threads per CU: %.1f us, %.0f GB/s\n", wr ? " 1 out" : "      ", wgs, t,
                       (double)n * 8 * (2 + wr) / t / 1e3);
            }
    }
    printf("done\n");
    return 0;
}
