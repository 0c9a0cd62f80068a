// What does a MIXED read + write stream reach on an MI355X?  (k_rg_h moves 0.27 GB in and
// 0.40 GB out per pass, k_pyramid_bands_xyb 0.02 in / 0.13 out, k_ref_blur 0.13 / 0.13 -- their ceiling is not the
// 5.9-6.0 TB/s of a pure read stream.)
//
// One kernel, 16 bytes per lane per access, R loads and W stores per lane and iteration over buffers far larger than
// the 256 MiB Infinity Cache; each workgroup walks its own contiguous chunk (whole DRAM pages per workgroup, as
// ssimu2_measure_read_stream does).  Reported: total bytes moved / time, for R:W = 1:0, 0:1, 1:1, 2:3 (k_rg_h's mix),
// 1:4 (the conversion kernel's), with default-policy and with nontemporal stores.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int R, int W, bool NT>
__global__ __launch_bounds__(256) void k_mix(const f4* __restrict__ src, f4* __restrict__ dst, size_t n16_per_stream, float* sink) {
    // every stream s in [0, R) / [0, W) is its own region of n16_per_stream f4 elements
    const size_t chunk = (n16_per_stream + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * chunk;
    const size_t hi = lo + chunk < n16_per_stream ? lo + chunk : n16_per_stream;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
        f4 v[R > 0 ? R : 1];
#pragma unroll
        for (int s = 0; s < R; ++s) v[s] = src[(size_t)s * n16_per_stream + i];
#pragma unroll
        for (int s = 0; s < R; ++s) acc += v[s];
#pragma unroll
        for (int s = 0; s < W; ++s) {
            const f4 o = acc + (float)s;
            if (NT) __builtin_nontemporal_store(o, &dst[(size_t)s * n16_per_stream + i]);
            else dst[(size_t)s * n16_per_stream + i] = o;
        }
    }
    if (W == 0 && acc.x == 12345.678f) *sink = acc.y;
}

template <int R, int W, bool NT>
static void run(const char* name, const f4* src, f4* dst, size_t n16, float* sink, hipEvent_t e0, hipEvent_t e1) {
    double best = 1e30;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_mix<R, W, NT>), dim3(4096), dim3(256), 0, 0, src, dst, n16, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)(R + W) * n16 * 16.0;
    printf("  %-34s %6.2f GB in %7.1f us  ->  %5.2f TB/s\n", name, bytes / 1e9, best * 1e3, bytes / (best * 1e-3) / 1e12);
}

int main() {
    const size_t n16 = (size_t)512 << 16;  // 512 MiB per stream (f4 elements = bytes / 16)
    f4 *src, *dst;
    float* sink;
    if (hipMalloc(&src, 2 * n16 * 16) != hipSuccess || hipMalloc(&dst, 4 * n16 * 16) != hipSuccess) {
        printf("hipMalloc failed\n");
        return 1;
    }
    hipMalloc(&sink, 4);
    hipMemset(src, 0, 2 * n16 * 16);
    hipMemset(dst, 0, 4 * n16 * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("mixed read + write streams, 16 B per lane, 512 MiB per stream, best of 5 launches:\n");
    run<1, 0, false>("read only (1:0)", src, dst, n16, sink, e0, e1);
    run<2, 0, false>("read only, two streams (2:0)", src, dst, n16, sink, e0, e1);
    run<0, 1, false>("write only (0:1)", src, dst, n16, sink, e0, e1);
    run<0, 1, true>("write only, nontemporal", src, dst, n16, sink, e0, e1);
    run<1, 1, false>("copy (1:1)", src, dst, n16, sink, e0, e1);
    run<1, 1, true>("copy (1:1), nontemporal stores", src, dst, n16, sink, e0, e1);
    run<2, 3, false>("k_rg_h's mix (2:3)", src, dst, n16, sink, e0, e1);
    run<2, 3, true>("k_rg_h's mix (2:3), nt stores", src, dst, n16, sink, e0, e1);
    run<1, 4, true>("conversion's mix (1:4), nt stores", src, dst, n16, sink, e0, e1);
    return 0;
}
