// Can the matrix pipe take the vertical 9-tap pass off the vector units?  gfx950.
//
// v_mfma_f32_4x4x1_16B_f32 computes, in each of 16 four-lane blocks, D[i][j] += A[i] * B[j]:
// with B = one blurred row (lane = column) and A = the four weights this input row has for
// four consecutive output rows (lane l holds the weight of output row l % 4), the four result
// registers are four output rows in the SAME lane = column layout.  Twelve instructions finish
// four output rows (K = 12 input rows for 9 useful taps: 1.33x redundant).
//
//   1. exactness: is the result the chain acc = fmaf(w, x, acc) over the 12 input rows in order?
//   2. rates: ns per wave-instruction per SIMD of the MFMA alone, of v_fmac alone, and of both
//      interleaved in one wave -- do the two pipes overlap?
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k_exact(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ d, int K) {
    const int l = threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k)
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k * 64 + l], b[k * 64 + l], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[i * 64 + l] = acc[i];
}

enum { M_MFMA, M_VALU, M_BOTH, M_BOTH2, NMODES };
static const char* kNames[NMODES] = {"mfma 4x4x1 alone (12 per iter)", "v_fmac alone (36 per iter)",
                                     "12 mfma + 36 v_fmac interleaved", "12 mfma + 72 v_fmac interleaved"};

template <int MODE>
__global__ void k_rate(float* out, int iters, float w, float x) {
    f4 acc[3];
    float v[12];
    for (int i = 0; i < 3; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 12; ++i) v[i] = threadIdx.x + i;
    const float sw = __builtin_amdgcn_readfirstlane(__float_as_int(w)) ? w : x;
    float a = w + (threadIdx.x & 3), b = x + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (MODE != M_VALU) {
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[g], 0, 0, 0);
            }
            if (MODE != M_MFMA) {
#pragma unroll
                for (int rep = 0; rep < (MODE == M_BOTH2 ? 2 : 1); ++rep)
#pragma unroll
                    for (int j = 0; j < 9; ++j)
                        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[j]) : "s"(sw), "v"(b));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 3; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 12; ++i) s += v[i];
    if (s == 12345.678f) out[0] = s;
}

// The arrangement MI355X_MICROARCH.md describes for the two pipes ("a MFMA-only wave and a
// VALU-only wave on the same CU run concurrently"): blocks alternate between the MFMA loop and the
// v_fmac loop, so every SIMD holds wps / 2 waves of each kind.  If the pipes co-execute ACROSS
// waves the launch takes about max(MFMA waves alone, VALU waves alone); if they share issue, the sum.
__global__ void k_split(float* out, int iters, float w, float x) {
    f4 acc[3];
    float v[12];
    for (int i = 0; i < 3; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 12; ++i) v[i] = threadIdx.x + i;
    const float sw = __builtin_amdgcn_readfirstlane(__float_as_int(w)) ? w : x;
    float a = w + (threadIdx.x & 3), b = x + threadIdx.x;
    if (blockIdx.x & 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[g], 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 9; ++j) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[j]) : "s"(sw), "v"(b));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 3; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 12; ++i) s += v[i];
    if (s == 12345.678f) out[0] = s;
}

static void rate_split(float* d) {
    const int iters = 20000;
    for (int wps = 2; wps <= 8; wps += 2) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k_split, dim3(256 * wps), dim3(256), 0, 0, d, 4000, 0.999f, 0.001f);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_split, dim3(256 * wps), dim3(256), 0, 0, d, iters, 0.999f, 0.001f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("split: %d mfma-only + %d v_fmac-only waves per SIMD  %.3f ms -> %.2f ns per round of iterations "
               "(every wave one iteration: %d x 12 mfma + %d x 36 v_fmac per SIMD)\n",
               wps / 2, wps / 2, ms, ms * 1e6 / iters, wps / 2, wps / 2);
    }
}

template <int MODE>
static void rate(float* d) {
    const int iters = 20000;
    for (int wps = 1; wps <= 8; wps = wps == 4 ? 6 : (wps == 6 ? 8 : wps * 2)) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k_rate<MODE>, dim3(256 * wps), dim3(256), 0, 0, d, 4000, 0.999f, 0.001f);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate<MODE>, dim3(256 * wps), dim3(256), 0, 0, d, iters, 0.999f, 0.001f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s waves/SIMD=%d  %.3f ms -> %.2f ns per loop iteration per wave-slot\n", kNames[MODE], wps,
               ms, ms * 1e6 / ((double)iters * wps));
    }
}

int main() {
    // ---- 1. exactness
    const int K = 12;
    std::vector<float> a(K * 64), b(K * 64), d(4 * 64), want(4 * 64);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    int bad_total = 0;
    float *da, *db, *dd;
    (void)hipMalloc(&da, K * 64 * 4);
    (void)hipMalloc(&db, K * 64 * 4);
    (void)hipMalloc(&dd, 4 * 64 * 4);
    for (int trial = 0; trial < 200; ++trial) {
        const float scale = trial % 4 == 0 ? 1e-3f : trial % 4 == 1 ? 1.0f : trial % 4 == 2 ? 1e-20f : 1e-36f;
        for (int k = 0; k < K; ++k)
            for (int l = 0; l < 64; ++l) {
                const int i = l & 3, t = k - i;  // banded weights, zero outside 0..8
                a[k * 64 + l] = (t >= 0 && t <= 8) ? 0.05f + 0.2f * rnd() : 0.0f;
                b[k * 64 + l] = (rnd() - (trial & 1 ? 0.5f : 0.0f)) * scale;
            }
        (void)hipMemcpy(da, a.data(), K * 64 * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(db, b.data(), K * 64 * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_exact, dim3(1), dim3(64), 0, 0, da, db, dd, K);
        (void)hipMemcpy(d.data(), dd, 4 * 64 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) {
                float acc = 0.f;
                for (int k = 0; k < K; ++k) acc = fmaf(a[k * 64 + (l & ~3) + i], b[k * 64 + l], acc);
                if (memcmp(&acc, &d[i * 64 + l], 4) != 0) {
                    if (bad_total + bad < 5) printf("  trial %d lane %d row %d: mfma %a  fmaf chain %a\n", trial, l, i, d[i * 64 + l], acc);
                    ++bad;
                }
            }
        bad_total += bad;
    }
    printf("exactness: %d of %d results differ from the in-order fmaf chain (scales 1e-3, 1, 1e-20, 1e-36)\n",
           bad_total, 200 * 256);
    // ---- 2. rates
    float* dz;
    (void)hipMalloc(&dz, 1024);
    rate<M_MFMA>(dz);
    rate<M_VALU>(dz);
    rate<M_BOTH>(dz);
    rate<M_BOTH2>(dz);
    rate_split(dz);
    return 0;
}
