// VALU issue-rate microbenchmark for gfx950: ns per wave-instruction per SIMD for the
// instruction kinds the marching kernel is made of, at 1..8 waves per SIMD.
// Eight independent dependency chains per wave, 64 instructions per loop iteration.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float float2_ __attribute__((ext_vector_type(2)));

enum { FMA, MULADD, PKFMA, FMAC_S, FMAMK, MULHI, RCP, CNDMASK, SDWA, LSHLADD64, NMODES };
static const char* kNames[NMODES] = {"v_fma_f32", "v_mul/v_add", "v_pk_fma_f32", "v_fmac_f32 (sgpr)",
                                     "v_fmamk_f32 (literal)", "v_mul_hi_u32", "v_rcp_f32",
                                     "v_cndmask_b32 (sgpr mask)", "v_lshlrev_b32_sdwa", "v_lshl_add_u64"};

#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
    float x[8];
    float2_ p[8];
    unsigned long long q[8];
    for (int i = 0; i < 8; ++i) {
        x[i] = threadIdx.x + i + 1.0f;
        p[i] = float2_{x[i], x[i] + 1.0f};
        q[i] = threadIdx.x * 8 + i;
    }
    float2_ pa = {a, a}, pb = {b, b};
    const float sa = __builtin_amdgcn_readfirstlane(__float_as_int(a)) ? a : b;  // uniform
    const unsigned long long mask = 0x5555555555555555ull;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE == FMA) {
#define OP(j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(a), "v"(b));
                REP8(OP)
#undef OP
            } else if (MODE == MULADD) {
#define OP(j) if (j & 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[j]) : "v"(b)); else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(a));
                REP8(OP)
#undef OP
            } else if (MODE == PKFMA) {
#define OP(j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(pa), "v"(pb));
                REP8(OP)
#undef OP
            } else if (MODE == FMAC_S) {
#define OP(j) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[j]) : "s"(sa), "v"(b));
                REP8(OP)
#undef OP
            } else if (MODE == FMAMK) {
#define OP(j) asm volatile("v_fmamk_f32 %0, %0, 0x3f7fbe77, %1" : "+v"(x[j]) : "v"(b));
                REP8(OP)
#undef OP
            } else if (MODE == MULHI) {
#define OP(j) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[j]) : "s"(0xAAAAAAABu));
                REP8(OP)
#undef OP
            } else if (MODE == RCP) {
#define OP(j) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[j]));
                REP8(OP)
#undef OP
            } else if (MODE == CNDMASK) {
#define OP(j) asm volatile("v_cndmask_b32 %0, 0, %0, %1" : "+v"(x[j]) : "s"(mask));
                REP8(OP)
#undef OP
            } else if (MODE == SDWA) {
#define OP(j) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(x[j]) : "v"(2));
                REP8(OP)
#undef OP
            } else if (MODE == LSHLADD64) {
#define OP(j) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[j]) : "s"(mask));
                REP8(OP)
#undef OP
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i] + p[i].x + p[i].y + (float)q[i];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
void run(float* d) {
    const int iters = 8000;
    for (int wps = 1; wps <= 8; wps = wps == 4 ? 6 : (wps == 6 ? 8 : wps * 2)) {
        // block = 256 threads = 4 waves = 1 wave per SIMD; wps blocks per CU
        int blocks = 256 * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 2000, 0.999f, 0.001f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f, 0.001f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double instr_per_wave = (double)iters * 64;
        double ns = ms * 1e6 / (instr_per_wave * wps);
        printf("%-26s waves/SIMD=%d  %.3f ms  -> %.3f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n",
               kNames[MODE], wps, ms, ns, ns * 2.4);
    }
}

int main() {
    float* d;
    hipMalloc(&d, 1024);
    run<FMA>(d);
    run<MULADD>(d);
    run<PKFMA>(d);
    run<FMAC_S>(d);
    run<FMAMK>(d);
    run<MULHI>(d);
    run<RCP>(d);
    run<CNDMASK>(d);
    run<SDWA>(d);
    run<LSHLADD64>(d);
    return 0;
}
