// VALU issue-rate microbenchmark for gfx950: wave-instructions per cycle per SIMD for
// v_fma_f32 / v_mul+v_add / v_pk_fma_f32 at 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float float2_ __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float2_ p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
    float2_ pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                             "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                             "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));
            }
        }
    }
    float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
void run(const char* name, float* d) {
    const int iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        // block = 256 threads = 4 waves = 1 wave per SIMD; wps blocks per CU
        int blocks = 256 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 0.999f, 0.001f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f, 0.001f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double instr_per_wave = (double)iters * 64;
        double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * wps);
        printf("%-14s waves/SIMD=%d  %.3f ms  -> %.3f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, wps, ms, ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4);
    }
}
int main() {
    float* d; hipMalloc(&d, 1024);
    run<0>("v_fma_f32", d);
    run<1>("v_mul/v_add", d);
    run<2>("v_pk_fma_f32", d);
    return 0;
}
