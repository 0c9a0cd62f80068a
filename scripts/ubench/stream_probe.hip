// Which pairs of HIP streams overlap their work?  For every pair among N streams created back to
// back: wall time of K dependent 100-us spin kernels per stream, enqueued alternately.
// distinct hardware queues: ~K x 100 us; a shared queue: more (barrier packets order the queue).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
__global__ void k_spin(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void k_busy(float* p, int iters) {  // fills the chip: 1024 workgroups
    float x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
    if (x == 12345.f) p[0] = x;
}
static double now_us() {
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 6, K = argc > 2 ? atoi(argv[2]) : 2;
    hipStream_t s[16];
    for (int i = 0; i < N; ++i) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
    float* d;
    hipMalloc(&d, 1024);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s[0], 2000LL);
    hipDeviceSynchronize();
    for (int mode = 0; mode < 2; ++mode) {
        printf(mode ? "busy kernels (1024 x 256 threads), K=%d per stream:\n" : "spin kernels (1 wave, 100 us), K=%d per stream:\n", K);
        for (int i = 0; i < N; ++i) {
            printf("  %d:", i);
            for (int j = 0; j < N; ++j) {
                if (j <= i) { printf("     --"); continue; }
                double best = 1e9;
                for (int rep = 0; rep < 3; ++rep) {
                    hipDeviceSynchronize();
                    const double t0 = now_us();
                    for (int k = 0; k < K; ++k) {
                        if (mode) {
                            hipLaunchKernelGGL(k_busy, dim3(1024), dim3(256), 0, s[i], d, 20000);
                            hipLaunchKernelGGL(k_busy, dim3(1024), dim3(256), 0, s[j], d, 20000);
                        } else {
                            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s[i], 10000LL);
                            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s[j], 10000LL);
                        }
                    }
                    hipStreamSynchronize(s[i]);
                    hipStreamSynchronize(s[j]);
                    const double us = now_us() - t0;
                    if (us < best) best = us;
                }
                printf(" %6.0f", best);
            }
            printf("\n");
        }
    }
    return 0;
}
