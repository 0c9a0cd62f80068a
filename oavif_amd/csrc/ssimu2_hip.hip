// SSIMULACRA2 scorer for MI355X (gfx950) -- kernels + the C ABI of include/ssimu2_hip.h.
//
// Replaces the one scorer call of oavif's target-quality search,
//   /root/reference/src/tq.zig:37  fssimu2.computeSsimu2(allocator, ref, dist, w, h, 3, null)
// (fssimu2 0.1.1 source is absent from the reference tree; the arithmetic follows the
// published SSIMULACRA2 v2.1 definition, see DESIGN.md "Oracle").
//
// Data layout in HBM (all per ctx, allocated once for the largest frame seen):
//   u8  frames  : ref, dist, interleaved RGB8, w*h*3 bytes each (the reference's layout)
//   lin pyramid : scales 1..5 of both frames, planar fp32 linear RGB [3][h_s][w_s]
//                 (scale 0 is read straight from the u8 frames through the sRGB LUT)
//   partials    : fp64 [scale][18 stats][blocks] per-workgroup partial sums
//   result      : fp64 [108 averages][score][nscales]
//
// Kernels (wave64, no MFMA: stencil + pointwise work):
//   k_down_u8 / k_down_f32 : 2x2 box average in linear light (edge replicated)
//   k_scale                : per scale, per workgroup: stage (tile + 4 px halo) of both
//                            frames into LDS as positive-XYB, horizontal 9-tap blur of the
//                            five planes {x, y, x^2, y^2, xy} per channel into LDS, vertical
//                            9-tap + SSIM / edge-difference maps in registers, wave-shuffle
//                            + LDS reduction to one fp64 partial per statistic
//   k_finalize             : fixed-order fp64 reduction of the partials, 108 averages,
//                            weighted sum, polynomial, score
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>
#include <type_traits>

#include "../../include/ssimu2_hip.h"

namespace {

constexpr int kNumScales = SSIMU2_NUM_SCALES;
constexpr int kStats = SSIMU2_STATS_PER_SCALE;

// ---- constants of the published algorithm (DESIGN.md "Algorithm")  ------------------
constexpr float kC2 = 0.0009f;
constexpr float kM00 = 0.30f, kM01 = 0.622f, kM02 = 0.078f;
constexpr float kM10 = 0.23f, kM11 = 0.692f, kM12 = 0.078f;
constexpr float kM20 = 0.24342268924547819f, kM21 = 0.20476744424496821f,
                kM22 = 0.55180986650955360f;
constexpr float kOpsinBias = 0.0037930732552754493f;

struct DevConst {
    float lut[256];     // 8-bit sRGB -> linear, fp32(rounded from fp64)
    float taps[5];      // FIR taps |d| = 0..4 of the sigma-1.5 recursive Gaussian
    float cbrt_bias;    // cbrtf(kOpsinBias)
    double weights[108];
};
__constant__ DevConst c_k;

// ---- device helpers ---------------------------------------------------------------------

// Arithmetic contract: this translation unit is compiled with
// -ffp-contract=off; every fused multiply-add is an explicit fmaf().  The sequence of IEEE
// operations per pixel is fixed and is the same one the CPU checker evaluates, because the
// SSIM map cancels hard in fp32 (a 1-ulp difference upstream moves the score by ~1e-3).

// Cube root from IEEE mul/fma only: bit-trick seed for x^(-1/3), two Newton steps,
// c = x y^2, one residual-corrected Newton step on c.  Max error 0.76 ulp.
__device__ __forceinline__ float cbrt_repro(float x) {
    uint32_t i = __float_as_uint(x);
    i = 0x54A2FA8Cu - i / 3u;
    float y = __uint_as_float(i);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float t = x * y;
        t = t * y;
        t = t * y;
        y = y * fmaf(-1.0f / 3.0f, t, 4.0f / 3.0f);
    }
    const float y2 = y * y;
    float c = x * y2;
    const float r = fmaf(c * c, c, -x);
    c = fmaf(r, y2 * (-1.0f / 3.0f), c);
    return x > 0.0f ? c : 0.0f;  // branch-free guard (inputs are clamped to >= 0)
}

__device__ __forceinline__ void linear_to_xyb(float r, float g, float b, float& X, float& Y,
                                              float& B) {
    float l = fmaf(kM00, r, fmaf(kM01, g, fmaf(kM02, b, kOpsinBias)));
    float m = fmaf(kM10, r, fmaf(kM11, g, fmaf(kM12, b, kOpsinBias)));
    float s = fmaf(kM20, r, fmaf(kM21, g, fmaf(kM22, b, kOpsinBias)));
    l = fmaxf(l, 0.0f);
    m = fmaxf(m, 0.0f);
    s = fmaxf(s, 0.0f);
    const float cb = c_k.cbrt_bias;
    l = cbrt_repro(l) - cb;
    m = cbrt_repro(m) - cb;
    s = cbrt_repro(s) - cb;
    const float x = 0.5f * (l - m), y = 0.5f * (l + m);
    B = (s - y) + 0.55f;
    X = fmaf(x, 14.0f, 0.42f);
    Y = y + 0.01f;
}

// symmetric 9-tap, same operation order as oracle fir_line(): one mul, four FMAs.
__device__ __forceinline__ float fir9(float c, float s1, float s2, float s3, float s4, float w0,
                                      float w1, float w2, float w3, float w4) {
    float acc = w0 * c;
    acc = fmaf(w1, s1, acc);
    acc = fmaf(w2, s2, acc);
    acc = fmaf(w3, s3, acc);
    acc = fmaf(w4, s4, acc);
    return acc;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ---- 2x2 box downsample in linear light -------------------------------------------------
// out(ox,oy) = ((p00 + p01) + p10 + p11) * 0.25, coordinates clamped to the last row/column
// (published Downsample(in, 2, 2)); same summation order as the oracle.

__global__ __launch_bounds__(256) void k_down_u8(const uint8_t* __restrict__ in0,
                                                 const uint8_t* __restrict__ in1,
                                                 float* __restrict__ out0,
                                                 float* __restrict__ out1, int w, int h, int ow,
                                                 int oh) {
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= ow || oy >= oh) return;
    const uint8_t* in = blockIdx.z ? in1 : in0;
    float* out = blockIdx.z ? out1 : out0;
    const int xa = 2 * ox, xb = min(2 * ox + 1, w - 1);
    const int ya = 2 * oy, yb = min(2 * oy + 1, h - 1);
    const size_t on = (size_t)ow * oh;
    const uint8_t* p00 = in + ((size_t)ya * w + xa) * 3;
    const uint8_t* p01 = in + ((size_t)ya * w + xb) * 3;
    const uint8_t* p10 = in + ((size_t)yb * w + xa) * 3;
    const uint8_t* p11 = in + ((size_t)yb * w + xb) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float sum = c_k.lut[p00[c]];
        sum += c_k.lut[p01[c]];
        sum += c_k.lut[p10[c]];
        sum += c_k.lut[p11[c]];
        out[c * on + (size_t)oy * ow + ox] = sum * 0.25f;
    }
}

__global__ __launch_bounds__(256) void k_down_f32(const float* __restrict__ in0,
                                                  const float* __restrict__ in1,
                                                  float* __restrict__ out0,
                                                  float* __restrict__ out1, int w, int h, int ow,
                                                  int oh) {
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= ow || oy >= oh) return;
    const float* in = blockIdx.z ? in1 : in0;
    float* out = blockIdx.z ? out1 : out0;
    const int xa = 2 * ox, xb = min(2 * ox + 1, w - 1);
    const int ya = 2 * oy, yb = min(2 * oy + 1, h - 1);
    const size_t n = (size_t)w * h, on = (size_t)ow * oh;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* p = in + c * n;
        float sum = p[(size_t)ya * w + xa];
        sum += p[(size_t)ya * w + xb];
        sum += p[(size_t)yb * w + xa];
        sum += p[(size_t)yb * w + xb];
        out[c * on + (size_t)oy * ow + ox] = sum * 0.25f;
    }
}

// ---- per-scale fused XYB + blur + maps + reduction (tile form) -----------------------------
constexpr int TX = 32, TY = 32, RAD = 4;
constexpr int RW = TX + 2 * RAD, RH = TY + 2 * RAD;  // 40 x 40 staged region
constexpr int RP = RW + 1;                          // padded LDS row
constexpr int HP = TX + 1;

template <bool kU8>
__global__ __launch_bounds__(256) void k_scale(const void* __restrict__ ref_in,
                                               const void* __restrict__ dist_in, int w, int h,
                                               double* __restrict__ partials, int nblocks) {
    __shared__ float s_raw[2][3][RH][RP];  // positive XYB of both frames, zero outside image
    __shared__ float s_h[5][RH][HP];       // horizontally blurred {x, y, xx, yy, xy}
    __shared__ double s_red[4][kStats];

    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TX - RAD, y0 = blockIdx.y * TY - RAD;
    const size_t n = (size_t)w * h;

    // stage A: load, sRGB LUT (scale 0), linear -> positive XYB, into LDS
    for (int i = tid; i < RW * RH; i += 256) {
        const int ly = i / RW, lx = i - ly * RW;
        const int gx = x0 + lx, gy = y0 + ly;
        float v[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        if (gx >= 0 && gx < w && gy >= 0 && gy < h) {
#pragma unroll
            for (int img = 0; img < 2; ++img) {
                float r, g, b;
                if (kU8) {
                    const uint8_t* p =
                        (const uint8_t*)(img ? dist_in : ref_in) + ((size_t)gy * w + gx) * 3;
                    r = c_k.lut[p[0]];
                    g = c_k.lut[p[1]];
                    b = c_k.lut[p[2]];
                } else {
                    const float* p = (const float*)(img ? dist_in : ref_in) + (size_t)gy * w + gx;
                    r = p[0];
                    g = p[n];
                    b = p[2 * n];
                }
                linear_to_xyb(r, g, b, v[img][0], v[img][1], v[img][2]);
            }
        }
#pragma unroll
        for (int img = 0; img < 2; ++img)
#pragma unroll
            for (int c = 0; c < 3; ++c) s_raw[img][c][ly][lx] = v[img][c];
    }
    __syncthreads();

    const float w0 = c_k.taps[0], w1 = c_k.taps[1], w2 = c_k.taps[2], w3 = c_k.taps[3],
                w4 = c_k.taps[4];
    float acc[kStats];
#pragma unroll
    for (int i = 0; i < kStats; ++i) acc[i] = 0.f;

#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        // horizontal pass: every staged row, TX output columns
        for (int i = tid; i < RH * TX; i += 256) {
            const int ly = i / TX, ox = i - ly * TX;
            const float* a = &s_raw[0][c][ly][ox];
            const float* b = &s_raw[1][c][ly][ox];
            float av[9], bv[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                av[k] = a[k];
                bv[k] = b[k];
            }
            float xx[9], yy[9], xy[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                xx[k] = av[k] * av[k];
                yy[k] = bv[k] * bv[k];
                xy[k] = av[k] * bv[k];
            }
#define H9(v) fir9(v[4], v[3] + v[5], v[2] + v[6], v[1] + v[7], v[0] + v[8], w0, w1, w2, w3, w4)
            s_h[0][ly][ox] = H9(av);
            s_h[1][ly][ox] = H9(bv);
            s_h[2][ly][ox] = H9(xx);
            s_h[3][ly][ox] = H9(yy);
            s_h[4][ly][ox] = H9(xy);
#undef H9
        }
        __syncthreads();
        // vertical pass + maps
        for (int i = tid; i < TY * TX; i += 256) {
            const int oy = i / TX, ox = i - oy * TX;
            const int gx = x0 + RAD + ox, gy = y0 + RAD + oy;
            if (gx < w && gy < h) {
                float v[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const float* p = &s_h[k][oy][ox];
                    v[k] = fir9(p[4 * HP], p[3 * HP] + p[5 * HP], p[2 * HP] + p[6 * HP],
                                p[1 * HP] + p[7 * HP], p[0] + p[8 * HP], w0, w1, w2, w3, w4);
                }
                const float mu1 = v[0], mu2 = v[1], s11 = v[2], s22 = v[3], s12 = v[4];
                const float r1 = s_raw[0][c][oy + RAD][ox + RAD];
                const float r2 = s_raw[1][c][oy + RAD][ox + RAD];
                // SSIM map
                const float mu11 = mu1 * mu1, mu22 = mu2 * mu2, mu12 = mu1 * mu2;
                const float dm = mu1 - mu2;
                const float num_m = fmaf(-dm, dm, 1.0f);
                const float num_s = fmaf(2.0f, s12 - mu12, kC2);
                const float denom_s = ((s11 - mu11) + (s22 - mu22)) + kC2;
                float d = 1.0f - (num_m * num_s) / denom_s;  // IEEE-rounded division
                d = fmaxf(d, 0.0f);
                float d2 = d * d;
                acc[c * 2] += d;
                acc[c * 2 + 1] += d2 * d2;
                // edge-difference map
                // (1+a)/(1+b) - 1 == (a-b)/(1+b): the published form is evaluated in fp64;
                // this one has no cancellation, so fp32 agrees with it to ~1e-7 relative
                const float ea = fabsf(r2 - mu2), eb = fabsf(r1 - mu1);
                const float e = (ea - eb) / (1.0f + eb);
                const float art = fmaxf(e, 0.0f), det = fmaxf(-e, 0.0f);
                const float a2 = art * art, t2 = det * det;
                acc[6 + c * 4] += art;
                acc[6 + c * 4 + 1] += a2 * a2;
                acc[6 + c * 4 + 2] += det;
                acc[6 + c * 4 + 3] += t2 * t2;
            }
        }
        __syncthreads();
    }

    // reduction: wave shuffle in fp64, then across the 4 waves through LDS
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < kStats; ++i) {
        const double s = wave_sum((double)acc[i]);
        if (lane == 0) s_red[wave][i] = s;
    }
    __syncthreads();
    if (tid < kStats) {
        const double s = ((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid];
        const int blk = blockIdx.y * gridDim.x + blockIdx.x;
        partials[(size_t)tid * nblocks + blk] = s;
    }
}

// ---- per-scale fused kernel, marching form ---------------------------------------------------
// One workgroup (10 waves) owns a strip of MW output columns and marches down `seg_rows`
// output rows, one image row per step.
//   waves 0-3 (converters): lane = one staged column (MW + 8 halo = 128) of one frame.  Each
//     step they convert the next input row (sRGB LUT at scale 0 -> opsin -> cbrt -> positive
//     XYB) into an LDS ring of raw rows; the global loads for the row after that are issued
//     first, so their latency spans a whole step.
//   waves 4-9 (blur + maps): two waves per XYB channel, lane = one output column.  Each step
//     a lane reads its 9-wide window of x (ref) and y (dist) from the ring, forms the
//     products, does the horizontal 9-tap of the five planes {x, y, xx, yy, xy} in
//     registers and pushes the results into a 9-row register window, from which the
//     vertical 9-tap and the SSIM / edge-difference maps of the row four steps back are
//     evaluated and accumulated.  The row loop is unrolled nine times so the window is
//     addressed with compile-time indices (no register moves).
// One output pixel per lane keeps the window at 45 registers (<= 128 VGPRs, 4 waves/SIMD):
// with a lone wave issuing a VALU op only every 4 cycles, occupancy is what fills the SIMDs.
// HBM traffic: each input pixel is read once per strip (+8/MW horizontal, +8/seg_rows
// vertical halo); only 18 partial sums per workgroup are written.
constexpr int MW = 120;        // output columns per strip
constexpr int MRW = MW + 8;    // staged columns (4 px halo each side) = 128 = 2 waves per frame
constexpr int MHALF = MW / 2;  // output columns per blur wave (lanes 0..59 active)
constexpr int RING = 16;       // raw-row ring depth (power of two >= 10: rows t-4 .. t+5)
constexpr int GROUP = 3;       // rows per barrier interval (divides the 9-phase unroll)
constexpr int MARCH_THREADS = 640;
constexpr int CONV_WAVES = 4;
constexpr int PF = 4;  // rows the converters' global loads run ahead of the conversion

// Correctly rounded a / b for operands that need no exponent scaling (here b is in
// [9e-4, 4], |a| < 4): v_rcp_f32 seed, one Newton step on the reciprocal, two fused
// residual corrections of the quotient -- the sequence hipcc emits for `a / b` minus
// v_div_scale / v_div_fixup, which only act on out-of-range exponents.  Same bits as the
// IEEE division the CPU checker performs.
__device__ __forceinline__ float div_rn(float a, float b) {
    float r = __builtin_amdgcn_rcpf(b);
    const float e0 = fmaf(-b, r, 1.0f);
    r = fmaf(e0, r, r);
    float q = a * r;
    const float e1 = fmaf(-b, q, a);
    q = fmaf(e1, r, q);
    const float e2 = fmaf(-b, q, a);
    return fmaf(e2, r, q);
}

// Raw values of one staged pixel of one frame of an input row.
template <bool kU8>
struct MarchRaw {
    typename std::conditional<kU8, uint32_t, float>::type v[3];
    bool ok;
};

// Issue the global loads of input row r, staged column `col` (global x = x0 - 4 + col).
template <bool kU8>
__device__ __forceinline__ void march_load(MarchRaw<kU8>& raw, const void* __restrict__ img, int w,
                                           int h, int x0, int r, int col) {
    const int gx = x0 - RAD + col;
    raw.ok = r >= 0 && r < h && gx >= 0 && gx < w;
    raw.v[0] = raw.v[1] = raw.v[2] = 0;
    if (raw.ok) {
        if constexpr (kU8) {
            const uint8_t* p = (const uint8_t*)img + ((size_t)r * w + gx) * 3;
            raw.v[0] = p[0];
            raw.v[1] = p[1];
            raw.v[2] = p[2];
        } else {
            const size_t n = (size_t)w * h;
            const float* p = (const float*)img + (size_t)r * w + gx;
            raw.v[0] = p[0];
            raw.v[1] = p[n];
            raw.v[2] = p[2 * n];
        }
    }
}

// Convert the loaded pixel to positive XYB and store it into ring slot `slot` of frame k
// (zeros outside the image: the blur is zero padded).
template <bool kU8>
__device__ __forceinline__ void march_convert(float (*ring)[2][3][MRW], const float* lut,
                                              const MarchRaw<kU8>& raw, int slot, int k, int col) {
    float rr, gg, bb, v[3];
    if constexpr (kU8) {
        rr = lut[raw.v[0]];
        gg = lut[raw.v[1]];
        bb = lut[raw.v[2]];
    } else {
        rr = raw.v[0];
        gg = raw.v[1];
        bb = raw.v[2];
    }
    linear_to_xyb(rr, gg, bb, v[0], v[1], v[2]);
#pragma unroll
    for (int c = 0; c < 3; ++c) ring[slot][k][c][col] = raw.ok ? v[c] : 0.0f;
}

template <int P>
__device__ __forceinline__ void march_hv_step(float (*ring)[2][3][MRW], float (&win)[5][9],
                                              float (&acc)[6], int t, int ch, int o, bool ok,
                                              float w0, float w1, float w2, float w3, float w4) {
    const int slot = t & (RING - 1);
    const float* px = &ring[slot][0][ch][o];  // staged columns o .. o+8, centre o+4
    const float* py = &ring[slot][1][ch][o];
    float xv[9], yv[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        xv[q] = px[q];
        yv[q] = py[q];
    }
#define H9(e) \
    fir9(e(4), e(3) + e(5), e(2) + e(6), e(1) + e(7), e(0) + e(8), w0, w1, w2, w3, w4)
#define EX(q) xv[q]
#define EY(q) yv[q]
#define EXX(q) (xv[q] * xv[q])
#define EYY(q) (yv[q] * yv[q])
#define EXY(q) (xv[q] * yv[q])
    win[0][P] = H9(EX);
    win[1][P] = H9(EY);
    win[2][P] = H9(EXX);
    win[3][P] = H9(EYY);
    win[4][P] = H9(EXY);
#undef EX
#undef EY
#undef EXX
#undef EYY
#undef EXY
#undef H9
    if (t >= 8) {  // window full (uniform across the workgroup)
        // vertical 9-tap for the row four steps back: row t-j sits in window slot (P-j) mod 9
        const int cslot = (t - 4) & (RING - 1);
        const float r1 = ring[cslot][0][ch][o + RAD];
        const float r2 = ring[cslot][1][ch][o + RAD];
        float v[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const float* q = win[k];
            v[k] = fir9(q[(P + 5) % 9], q[(P + 4) % 9] + q[(P + 6) % 9],
                        q[(P + 3) % 9] + q[(P + 7) % 9], q[(P + 2) % 9] + q[(P + 8) % 9],
                        q[(P + 1) % 9] + q[P], w0, w1, w2, w3, w4);
        }
        const float mu1 = v[0], mu2 = v[1], s11 = v[2], s22 = v[3], s12 = v[4];
        const float mu11 = mu1 * mu1, mu22 = mu2 * mu2, mu12 = mu1 * mu2;
        const float dm = mu1 - mu2;
        const float num_m = fmaf(-dm, dm, 1.0f);
        const float num_s = fmaf(2.0f, s12 - mu12, kC2);
        const float denom_s = ((s11 - mu11) + (s22 - mu22)) + kC2;
        float d = 1.0f - div_rn(num_m * num_s, denom_s);
        d = fmaxf(d, 0.0f);
        const float ea = fabsf(r2 - mu2), eb = fabsf(r1 - mu1);
        float e = div_rn(ea - eb, 1.0f + eb);  // == (1+ea)/(1+eb) - 1, no cancellation
        d = ok ? d : 0.0f;                      // column inside the image?
        e = ok ? e : 0.0f;
        const float art = fmaxf(e, 0.0f), det = fmaxf(-e, 0.0f);
        const float d2 = d * d, a2 = art * art, t2 = det * det;
        acc[0] += d;
        acc[1] += d2 * d2;
        acc[2] += art;
        acc[3] += a2 * a2;
        acc[4] += det;
        acc[5] += t2 * t2;
    }
}

template <bool kU8>
__global__ __launch_bounds__(MARCH_THREADS) void k_march(const void* __restrict__ ref_in,
                                                            const void* __restrict__ dist_in,
                                                            int w, int h, int seg_rows,
                                                            double* __restrict__ partials,
                                                            int nblocks) {
    __shared__ __attribute__((aligned(16))) float s_ring[RING][2][3][MRW];
    __shared__ float s_lut[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = blockIdx.x * MW;
    const int y0 = blockIdx.y * seg_rows;
    const int rows_out = min(seg_rows, h - y0);
    const int steps = rows_out + 2 * RAD;  // input rows y0-4 .. y0+rows_out+3
    if (kU8 && tid < 256) s_lut[tid] = c_k.lut[tid];
    __syncthreads();

    const float w0 = c_k.taps[0], w1 = c_k.taps[1], w2 = c_k.taps[2], w3 = c_k.taps[3],
                w4 = c_k.taps[4];
    const bool is_conv = wave < CONV_WAVES;
    // converter state: wave -> (frame, half of the staged columns)
    const int frame = wave >> 1;
    const int col = ((wave & 1) << 6) + lane;
    const void* img = frame ? dist_in : ref_in;
    MarchRaw<kU8> q[PF], nxt;  // q[0] = next row to convert, q[PF-1] = newest loaded
    // blur state (fp32 sums: at most seg_rows <= 135 terms per lane before the fp64 reduce)
    float win[5][9];
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int hw = wave - CONV_WAVES;
    const int ch = hw >> 1;
    const bool hv_active = lane < MHALF;
    const int o = (hw & 1) * MHALF + (hv_active ? lane : 0);
    const bool ok = x0 + o < w;

    // Input row j (= image row y0-4+j) lives in ring slot j & (RING-1).  The workgroup
    // synchronises once per GROUP rows: while the blur waves consume rows 3I..3I+2 the
    // converters fill rows 3I+3..3I+5.  Global loads run PF rows ahead of the conversion so
    // HBM latency is off the critical path.
    if (is_conv) {
        MarchRaw<kU8> first[GROUP];
#pragma unroll
        for (int j = 0; j < GROUP; ++j) march_load<kU8>(first[j], img, w, h, x0, y0 - RAD + j, col);
#pragma unroll
        for (int j = 0; j < PF; ++j)
            march_load<kU8>(q[j], img, w, h, x0, y0 - RAD + GROUP + j, col);
#pragma unroll
        for (int j = 0; j < GROUP; ++j) march_convert<kU8>(s_ring, s_lut, first[j], j, frame, col);
    }
    __syncthreads();

#define MARCH_STEP(P)                                                                          \
    {                                                                                          \
        const int t = t0 + P;                                                                  \
        if (t < steps) {                                                                       \
            if (is_conv) {                                                                     \
                if (t + GROUP < steps) {                                                       \
                    march_load<kU8>(nxt, img, w, h, x0, y0 - RAD + t + GROUP + PF, col);       \
                    march_convert<kU8>(s_ring, s_lut, q[0], (t + GROUP) & (RING - 1), frame,   \
                                       col);                                                   \
                    _Pragma("unroll") for (int j = 0; j + 1 < PF; ++j) q[j] = q[j + 1];        \
                    q[PF - 1] = nxt;                                                           \
                }                                                                              \
            } else {                                                                           \
                march_hv_step<P>(s_ring, win, acc, t, ch, o, ok, w0, w1, w2, w3, w4);          \
            }                                                                                  \
        }                                                                                      \
        if ((P % GROUP) == GROUP - 1 && t - (GROUP - 1) < steps) __syncthreads();              \
    }
#pragma unroll 1
    for (int t0 = 0; t0 < steps; t0 += 9) {
        MARCH_STEP(0)
        MARCH_STEP(1)
        MARCH_STEP(2)
        MARCH_STEP(3)
        MARCH_STEP(4)
        MARCH_STEP(5)
        MARCH_STEP(6)
        MARCH_STEP(7)
        MARCH_STEP(8)
    }
#undef MARCH_STEP

    // the two half-strip waves of a channel each publish their sums; combined in fixed order
    __shared__ double s_part[6][6];
    if (!is_conv) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double v = wave_sum(hv_active ? (double)acc[k] : 0.0);
            if (lane == 0) s_part[hw][k] = v;
        }
    }
    __syncthreads();
    if (tid < kStats) {
        // tid = stat index: 0..5 ssim (c*2+n), 6..17 edge (c*4+k)
        const int c = tid < 6 ? tid >> 1 : (tid - 6) >> 2;
        const int k = tid < 6 ? (tid & 1) : 2 + ((tid - 6) & 3);
        const int blk = blockIdx.y * gridDim.x + blockIdx.x;
        partials[(size_t)tid * nblocks + blk] = s_part[2 * c][k] + s_part[2 * c + 1][k];
    }
}

// ---- final reduction ----------------------------------------------------------------------
struct ScaleInfo {
    int nblocks[kNumScales];
    long long offset[kNumScales];  // in doubles, into partials
    double inv_pixels[kNumScales];
    int nscales;
};

// result layout: [0..107] averages [scale][18], [108] score, [109] nscales
__global__ __launch_bounds__(1024) void k_finalize(const double* __restrict__ partials,
                                                   ScaleInfo si, double* __restrict__ result) {
    __shared__ double s_avg[kNumScales * kStats];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int item = wave; item < kNumScales * kStats; item += 16) {
        const int scale = item / kStats, stat = item - scale * kStats;
        double v = 0.0;
        if (scale < si.nscales) {
            const double* p = partials + si.offset[scale] + (size_t)stat * si.nblocks[scale];
            for (int b = lane; b < si.nblocks[scale]; b += 64) v += p[b];
            v = wave_sum(v);
            v *= si.inv_pixels[scale];
            // odd stats are L4 norms: 4th root of the mean of d^4
            if (stat & 1) v = sqrt(sqrt(v));
        }
        if (lane == 0) {
            s_avg[item] = v;
            result[item] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // published Score(): running weight index over the scales actually present
        double ssim = 0.0;
        int i = 0;
        for (int c = 0; c < 3; ++c)
            for (int scale = 0; scale < si.nscales; ++scale) {
                const double* a = s_avg + scale * kStats;
                for (int n = 0; n < 2; ++n) {
                    ssim += c_k.weights[i++] * fabs(a[c * 2 + n]);
                    ssim += c_k.weights[i++] * fabs(a[6 + c * 4 + n]);
                    ssim += c_k.weights[i++] * fabs(a[6 + c * 4 + n + 2]);
                }
            }
        ssim = ssim * 0.9562382616834844;
        ssim = 2.326765642916932 * ssim - 0.020884521182843837 * ssim * ssim +
               6.248496625763138e-05 * ssim * ssim * ssim;
        if (ssim > 0.0) ssim = 100.0 - 10.0 * pow(ssim, 0.6276336467831387);
        else ssim = 100.0;
        result[108] = ssim;
        result[109] = (double)si.nscales;
    }
}

// ---- host side ------------------------------------------------------------------------------

const double kWeightsHost[108] = {
    0.0, 0.0007376606707406586, 0.0, 0.0, 0.0007793481682867309, 0.0,
    0.0, 0.0004371155730107379, 0.0, 1.1041726426657346, 0.00066284834129271,
    0.00015231632783718752, 0.0, 0.0016406437456599754, 0.0, 1.8422455520539298,
    11.441172603757666, 0.0, 0.0007989109436015163, 0.000176816438078653, 0.0,
    1.8787594979546387, 10.949069906051982, 0.0, 0.0007289346991508072,
    0.9677937080626833, 0.0, 0.00014003424285435884, 0.9981766977854967,
    0.00031949755934435053, 0.0004550992113792063, 0.0, 0.0, 0.0013648766163243398,
    0.0, 0.0, 0.0, 0.0, 0.0, 7.466890328078848, 0.0, 17.445833984131262,
    0.0006235601634041466, 0.0, 0.0, 6.683678146179332, 0.00037724407979611296,
    1.027889937768264, 225.20515300849274, 0.0, 0.0, 19.213238186143016,
    0.0011401524586618361, 0.001237755635509985, 176.39317598450694, 0.0, 0.0,
    24.43300999870476, 0.28520802612117757, 0.0004485436923833408, 0.0, 0.0, 0.0,
    34.77906344483772, 44.835625328877896, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0,
    0.0008680556573291698, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0005313191874358747, 0.0,
    0.00016533814161379112, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0004179171803251336,
    0.0017290828234722833, 0.0, 0.0020827005846636437, 0.0, 0.0, 8.826982764996862,
    23.19243343998926, 0.0, 95.1080498811086, 0.9863978034400682, 0.9834382792465353,
    0.0012286405048278493, 171.2667255897307, 0.9807858872435379, 0.0, 0.0, 0.0,
    0.0005130064588990679, 0.0, 0.00010854057858411537};

// FIR taps of the sigma = 1.5 recursive Gaussian (Charalampidis 2016 truncated cosines,
// N = 5): w(d) = sum_k n2_k / sin(w_k) * sin(w_k (d + N)); see DESIGN.md "Blur".
void gaussian_taps(double sigma, float taps[5]) {
    const double kPi = 3.141592653589793238;
    const double radius = round(3.2795 * sigma + 0.2546);
    const double pi_div_2r = kPi / (2.0 * radius);
    const double om[3] = {pi_div_2r, 3.0 * pi_div_2r, 5.0 * pi_div_2r};
    const double p1 = 1.0 / tan(0.5 * om[0]), p3 = -1.0 / tan(0.5 * om[1]),
                 p5 = 1.0 / tan(0.5 * om[2]);
    const double r1 = p1 * p1 / sin(om[0]), r3 = -p3 * p3 / sin(om[1]),
                 r5 = p5 * p5 / sin(om[2]);
    double rho[3];
    for (int i = 0; i < 3; ++i) rho[i] = exp(-0.5 * sigma * sigma * om[i] * om[i]) / radius;
    const double D13 = p1 * r3 - r1 * p3, D35 = p3 * r5 - r3 * p5, D51 = p5 * r1 - r5 * p1;
    const double z15 = D35 / D13, z35 = D51 / D13;
    // solve [p1 p3 p5; r1 r3 r5; z15 z35 1] beta = gamma by Cramer's rule
    const double g[3] = {1.0, radius * radius - sigma * sigma,
                         z15 * rho[0] + z35 * rho[1] + rho[2]};
    const double a = p1, b = p3, c = p5, d = r1, e = r3, f = r5, gg = z15, hh = z35, ii = 1.0;
    const double det = a * (e * ii - f * hh) - b * (d * ii - f * gg) + c * (d * hh - e * gg);
    const double beta0 = (g[0] * (e * ii - f * hh) - b * (g[1] * ii - f * g[2]) +
                          c * (g[1] * hh - e * g[2])) / det;
    const double beta1 = (a * (g[1] * ii - f * g[2]) - g[0] * (d * ii - f * gg) +
                          c * (d * g[2] - g[1] * gg)) / det;
    const double beta2 = (a * (e * g[2] - g[1] * hh) - b * (d * g[2] - g[1] * gg) +
                          g[0] * (d * hh - e * gg)) / det;
    const double beta[3] = {beta0, beta1, beta2};
    for (int t = 0; t < 5; ++t) {
        double wsum = 0.0;
        for (int k = 0; k < 3; ++k) {
            const double n2 = -beta[k] * cos(om[k] * (radius + 1.0));
            wsum += n2 / sin(om[k]) * sin(om[k] * (t + radius));
        }
        taps[t] = (float)wsum;
    }
}

// host twin of the device cbrt_repro (same IEEE sequence; this file is built with
// -ffp-contract=off, fmaf is the correctly rounded libm/hardware fma)
float cbrt_repro_host(float x) {
    if (!(x > 0.0f)) return 0.0f;
    uint32_t i;
    memcpy(&i, &x, 4);
    i = 0x54A2FA8Cu - i / 3u;
    float y;
    memcpy(&y, &i, 4);
    for (int k = 0; k < 2; ++k) {
        float t = x * y;
        t = t * y;
        t = t * y;
        y = y * fmaf(-1.0f / 3.0f, t, 4.0f / 3.0f);
    }
    const float y2 = y * y;
    float c = x * y2;
    const float r = fmaf(c * c, c, -x);
    c = fmaf(r, y2 * (-1.0f / 3.0f), c);
    return c;
}

thread_local std::string g_create_error;

}  // namespace

struct ssimu2_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    // capacity (bytes / floats / doubles currently allocated)
    size_t cap_u8 = 0, cap_lin = 0;
    // device buffers
    uint8_t* d_ref_u8 = nullptr;
    uint8_t* d_dist_u8 = nullptr;
    float* d_lin_ref = nullptr;   // scales 1..5 packed
    float* d_lin_dist = nullptr;
    double* d_partials = nullptr;
    double* d_result = nullptr;   // 110 doubles
    double* h_result = nullptr;   // pinned mirror
    size_t partial_cap = 0;

    // reference state
    bool have_ref = false;
    uint32_t ref_w = 0, ref_h = 0;
    bool pending = false;

    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    bool use_march = true;  // marching kernel (default) vs tile kernel (OAVIF_AMD_KERNEL=tile)
    int seg_rows_override = 0;

    int fail(int code, const char* what, hipError_t e = hipSuccess) {
        char buf[256];
        if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
        else snprintf(buf, sizeof buf, "%s", what);
        err = buf;
        return code;
    }
};

namespace {

#define HIP_TRY(ctx, call)                                                   \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess) return (ctx)->fail(SSIMU2_ERR_HIP, #call, e_); \
    } while (0)

struct Pyramid {
    int w[kNumScales], h[kNumScales];
    size_t lin_off[kNumScales];  // float offset of scale s (s >= 1) in the lin buffers
    size_t lin_total;
    int nscales;
};

// Scale s is scored iff scale s-1 is at least 8x8 (the published loop tests the size
// before downsampling).
Pyramid make_pyramid(uint32_t w, uint32_t h) {
    Pyramid p{};
    int cw = (int)w, ch = (int)h;
    size_t off = 0;
    p.nscales = 0;
    for (int s = 0; s < kNumScales; ++s) {
        if (cw < 8 || ch < 8) break;
        if (s) {
            cw = (cw + 1) / 2;
            ch = (ch + 1) / 2;
            p.lin_off[s] = off;
            off += (size_t)3 * cw * ch;
        }
        p.w[s] = cw;
        p.h[s] = ch;
        ++p.nscales;
    }
    p.lin_total = off;
    return p;
}

void free_buffers(ssimu2_ctx* c) {
    (void)hipFree(c->d_ref_u8);
    (void)hipFree(c->d_dist_u8);
    (void)hipFree(c->d_lin_ref);
    (void)hipFree(c->d_lin_dist);
    (void)hipFree(c->d_partials);
    c->d_ref_u8 = c->d_dist_u8 = nullptr;
    c->d_lin_ref = c->d_lin_dist = nullptr;
    c->d_partials = nullptr;
    c->cap_u8 = c->cap_lin = 0;
    c->partial_cap = 0;
}

// Rows per workgroup of the marching kernel.  Two 10-wave workgroups fit a CU, so 512
// workgroups are one fully balanced resident round of the 256 CUs; aim at that, but keep a
// segment between 8 rows (vertical halo cost 8/seg) and 160 rows (fp32 partial sums).
int march_seg_rows(const ssimu2_ctx* c, int w, int h) {
    if (c->seg_rows_override > 0) return c->seg_rows_override;
    const int nstrips = (w + MW - 1) / MW;
    int nsegs = (512 + nstrips / 2) / nstrips;
    if (nsegs < 1) nsegs = 1;
    int seg = (h + nsegs - 1) / nsegs;
    if (seg < 8) seg = 8;
    if (seg > 160) seg = 160;
    return seg;
}

int scale_blocks(const ssimu2_ctx* c, const Pyramid& p, int s) {
    if (c->use_march) {
        const int seg = march_seg_rows(c, p.w[s], p.h[s]);
        return ((p.w[s] + MW - 1) / MW) * ((p.h[s] + seg - 1) / seg);
    }
    return ((p.w[s] + TX - 1) / TX) * ((p.h[s] + TY - 1) / TY);
}

size_t partial_doubles(const ssimu2_ctx* c, const Pyramid& p) {
    size_t t = 0;
    for (int s = 0; s < p.nscales; ++s) t += (size_t)scale_blocks(c, p, s) * kStats;
    return t;
}

int ensure_capacity(ssimu2_ctx* c, uint32_t w, uint32_t h) {
    const Pyramid p = make_pyramid(w, h);
    const size_t need_u8 = (size_t)w * h * 3, need_lin = p.lin_total + 4,
                 need_part = partial_doubles(c, p) + 8;
    if (c->d_ref_u8 && need_u8 <= c->cap_u8 && need_lin <= c->cap_lin &&
        need_part <= c->partial_cap)
        return SSIMU2_OK;
    // growing frees everything, which also drops a cached reference
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const size_t nu8 = need_u8 > c->cap_u8 ? need_u8 : c->cap_u8;
    const size_t nlin = need_lin > c->cap_lin ? need_lin : c->cap_lin;
    const size_t npart = need_part > c->partial_cap ? need_part : c->partial_cap;
    free_buffers(c);
    c->have_ref = false;
    hipError_t e;
    if ((e = hipMalloc(&c->d_ref_u8, nu8)) != hipSuccess ||
        (e = hipMalloc(&c->d_dist_u8, nu8)) != hipSuccess ||
        (e = hipMalloc(&c->d_lin_ref, nlin * sizeof(float))) != hipSuccess ||
        (e = hipMalloc(&c->d_lin_dist, nlin * sizeof(float))) != hipSuccess ||
        (e = hipMalloc(&c->d_partials, npart * sizeof(double))) != hipSuccess) {
        free_buffers(c);
        return c->fail(SSIMU2_ERR_OOM, "hipMalloc(frame buffers)", e);
    }
    c->cap_u8 = nu8;
    c->cap_lin = nlin;
    c->partial_cap = npart;
    return SSIMU2_OK;
}

void launch_scale(ssimu2_ctx* c, const Pyramid& p, int s, const uint8_t* d_ref,
                  const uint8_t* d_dist, double* part) {
    const void* a = s == 0 ? (const void*)d_ref : (const void*)(c->d_lin_ref + p.lin_off[s]);
    const void* b = s == 0 ? (const void*)d_dist : (const void*)(c->d_lin_dist + p.lin_off[s]);
    const int nb = scale_blocks(c, p, s);
    if (c->use_march) {
        const int seg = march_seg_rows(c, p.w[s], p.h[s]);
        dim3 grid((p.w[s] + MW - 1) / MW, (p.h[s] + seg - 1) / seg), block(MARCH_THREADS);
        if (s == 0)
            hipLaunchKernelGGL(k_march<true>, grid, block, 0, c->stream, a, b, p.w[s], p.h[s], seg,
                               part, nb);
        else
            hipLaunchKernelGGL(k_march<false>, grid, block, 0, c->stream, a, b, p.w[s], p.h[s],
                               seg, part, nb);
        return;
    }
    dim3 grid((p.w[s] + TX - 1) / TX, (p.h[s] + TY - 1) / TY), block(256);
    if (s == 0)
        hipLaunchKernelGGL(k_scale<true>, grid, block, 0, c->stream, a, b, p.w[s], p.h[s], part, nb);
    else
        hipLaunchKernelGGL(k_scale<false>, grid, block, 0, c->stream, a, b, p.w[s], p.h[s], part,
                           nb);
}

// Enqueue the whole score of (d_ref, d_dist) on the ctx stream.  `ref_pyramid_ready`:
// the reference's linear pyramid in d_lin_ref is already valid for this frame size.
int enqueue_score(ssimu2_ctx* c, const uint8_t* d_ref, const uint8_t* d_dist, uint32_t w,
                  uint32_t h, bool ref_pyramid_ready) {
    const Pyramid p = make_pyramid(w, h);
    ScaleInfo si{};
    si.nscales = p.nscales;
    size_t poff = 0;
    // 1. linear-light pyramids (scale s from scale s-1)
    for (int s = 1; s < p.nscales; ++s) {
        const int iw = p.w[s - 1], ih = p.h[s - 1], ow = p.w[s], oh = p.h[s];
        dim3 grid((ow + 63) / 64, (oh + 3) / 4, 2), block(256);
        float* o_ref = c->d_lin_ref + p.lin_off[s];
        float* o_dist = c->d_lin_dist + p.lin_off[s];
        if (ref_pyramid_ready) {
            // only the distorted frame: z = 1 -> run with both slots pointing at dist
            grid.z = 1;
            if (s == 1)
                hipLaunchKernelGGL(k_down_u8, grid, block, 0, c->stream, d_dist, d_dist, o_dist,
                                   o_dist, iw, ih, ow, oh);
            else
                hipLaunchKernelGGL(k_down_f32, grid, block, 0, c->stream,
                                   c->d_lin_dist + p.lin_off[s - 1],
                                   c->d_lin_dist + p.lin_off[s - 1], o_dist, o_dist, iw, ih, ow,
                                   oh);
        } else if (s == 1) {
            hipLaunchKernelGGL(k_down_u8, grid, block, 0, c->stream, d_ref, d_dist, o_ref,
                               o_dist, iw, ih, ow, oh);
        } else {
            hipLaunchKernelGGL(k_down_f32, grid, block, 0, c->stream,
                               c->d_lin_ref + p.lin_off[s - 1],
                               c->d_lin_dist + p.lin_off[s - 1], o_ref, o_dist, iw, ih, ow, oh);
        }
    }
    // 2. per-scale fused kernel
    for (int s = 0; s < p.nscales; ++s) {
        const int nb = scale_blocks(c, p, s);
        si.nblocks[s] = nb;
        si.offset[s] = (long long)poff;
        si.inv_pixels[s] = 1.0 / ((double)p.w[s] * (double)p.h[s]);
        launch_scale(c, p, s, d_ref, d_dist, c->d_partials + poff);
        poff += (size_t)nb * kStats;
    }
    // 3. final reduction + score
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(1024), 0, c->stream,
                       (const double*)c->d_partials, si, c->d_result);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(c->h_result, c->d_result, 110 * sizeof(double),
                              hipMemcpyDeviceToHost, c->stream));
    c->pending = true;
    return SSIMU2_OK;
}

int check_args(ssimu2_ctx* c, const void* a, const void* b, uint32_t w, uint32_t h) {
    if (!c) return SSIMU2_ERR_INVALID_ARG;
    if (!a || !b) return c->fail(SSIMU2_ERR_INVALID_ARG, "null image pointer");
    if (w == 0 || h == 0) return c->fail(SSIMU2_ERR_INVALID_ARG, "zero image dimension");
    if ((uint64_t)w * h > (1ull << 31) / 3)
        return c->fail(SSIMU2_ERR_INVALID_ARG, "image larger than 2^31/3 pixels");
    return SSIMU2_OK;
}

}  // namespace

extern "C" {

const char* ssimu2_version(void) { return "oavif_amd ssimu2 gfx950 v2 (marching kernels)"; }

const char* ssimu2_last_error(const ssimu2_ctx* ctx) {
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int ssimu2_ctx_create(int device, void* hip_stream, ssimu2_ctx** out_ctx) {
    if (!out_ctx) return SSIMU2_ERR_INVALID_ARG;
    *out_ctx = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        g_create_error = "no usable HIP device (hipGetDeviceCount: ";
        g_create_error += hipGetErrorString(e);
        g_create_error += ")";
        return SSIMU2_ERR_NO_DEVICE;
    }
    ssimu2_ctx* c = new (std::nothrow) ssimu2_ctx();
    if (!c) return SSIMU2_ERR_OOM;
    c->device = device;
    if (const char* k = getenv("OAVIF_AMD_KERNEL")) c->use_march = strcmp(k, "tile") != 0;
    if (const char* k = getenv("OAVIF_AMD_SEG_ROWS")) c->seg_rows_override = atoi(k);
#define CREATE_TRY(call)                                   \
    do {                                                   \
        hipError_t e2 = (call);                            \
        if (e2 != hipSuccess) {                            \
            g_create_error = std::string(#call) + ": " + hipGetErrorString(e2); \
            ssimu2_ctx_destroy(c);                         \
            return SSIMU2_ERR_HIP;                         \
        }                                                  \
    } while (0)
    CREATE_TRY(hipSetDevice(device));
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
    } else {
        CREATE_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    CREATE_TRY(hipEventCreate(&c->ev0));
    CREATE_TRY(hipEventCreate(&c->ev1));
    CREATE_TRY(hipMalloc(&c->d_result, 110 * sizeof(double)));
    CREATE_TRY(hipHostMalloc(&c->h_result, 110 * sizeof(double), hipHostMallocDefault));
    // constants
    DevConst* k = new (std::nothrow) DevConst();
    if (!k) {
        ssimu2_ctx_destroy(c);
        return SSIMU2_ERR_OOM;
    }
    for (int i = 0; i < 256; ++i) {
        const double v = (double)i / 255.0;
        k->lut[i] = (float)(v <= 0.04045 ? v / 12.92 : pow((v + 0.055) / 1.055, 2.4));
    }
    gaussian_taps(1.5, k->taps);
    k->cbrt_bias = cbrt_repro_host(kOpsinBias);
    memcpy(k->weights, kWeightsHost, sizeof kWeightsHost);
    hipError_t ec = hipMemcpyToSymbol(HIP_SYMBOL(c_k), k, sizeof(DevConst));
    delete k;
    CREATE_TRY(ec);
#undef CREATE_TRY
    *out_ctx = c;
    return SSIMU2_OK;
}

void ssimu2_ctx_destroy(ssimu2_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_buffers(c);
    (void)hipFree(c->d_result);
    (void)hipHostFree(c->h_result);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int ssimu2_wait(ssimu2_ctx* c, double* out_score) {
    if (!c) return SSIMU2_ERR_INVALID_ARG;
    if (!c->pending) return c->fail(SSIMU2_ERR_INVALID_ARG, "ssimu2_wait: nothing enqueued");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->pending = false;
    if (out_score) *out_score = c->h_result[108];
    return SSIMU2_OK;
}

int ssimu2_enqueue_rgb8_device(ssimu2_ctx* c, const void* d_ref, const void* d_dist, uint32_t w,
                               uint32_t h) {
    int rc = check_args(c, d_ref, d_dist, w, h);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = ensure_capacity(c, w, h))) return rc;
    c->have_ref = false;  // the lin_ref pyramid is overwritten
    return enqueue_score(c, (const uint8_t*)d_ref, (const uint8_t*)d_dist, w, h, false);
}

int ssimu2_score_rgb8_device(ssimu2_ctx* c, const void* d_ref, const void* d_dist, uint32_t w,
                             uint32_t h, double* out_score) {
    if (!out_score) return c ? c->fail(SSIMU2_ERR_INVALID_ARG, "null out_score") : SSIMU2_ERR_INVALID_ARG;
    int rc = ssimu2_enqueue_rgb8_device(c, d_ref, d_dist, w, h);
    if (rc) return rc;
    return ssimu2_wait(c, out_score);
}

int ssimu2_score_rgb8(ssimu2_ctx* c, const uint8_t* ref, const uint8_t* dist, uint32_t w,
                      uint32_t h, uint32_t channels, double* out_score) {
    int rc = check_args(c, ref, dist, w, h);
    if (rc) return rc;
    if (channels != 3) return c->fail(SSIMU2_ERR_UNSUPPORTED, "channels must be 3");
    if (!out_score) return c->fail(SSIMU2_ERR_INVALID_ARG, "null out_score");
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = ensure_capacity(c, w, h))) return rc;
    c->have_ref = false;
    const size_t bytes = (size_t)w * h * 3;
    HIP_TRY(c, hipMemcpyAsync(c->d_ref_u8, ref, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_dist_u8, dist, bytes, hipMemcpyHostToDevice, c->stream));
    if ((rc = enqueue_score(c, c->d_ref_u8, c->d_dist_u8, w, h, false))) return rc;
    return ssimu2_wait(c, out_score);
}

int ssimu2_set_reference(ssimu2_ctx* c, const uint8_t* ref, uint32_t w, uint32_t h) {
    int rc = check_args(c, ref, ref, w, h);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = ensure_capacity(c, w, h))) return rc;
    const size_t bytes = (size_t)w * h * 3;
    HIP_TRY(c, hipMemcpyAsync(c->d_ref_u8, ref, bytes, hipMemcpyHostToDevice, c->stream));
    // build the reference's linear pyramid once
    const Pyramid p = make_pyramid(w, h);
    for (int s = 1; s < p.nscales; ++s) {
        const int iw = p.w[s - 1], ih = p.h[s - 1], ow = p.w[s], oh = p.h[s];
        dim3 grid((ow + 63) / 64, (oh + 3) / 4, 1), block(256);
        float* o_ref = c->d_lin_ref + p.lin_off[s];
        if (s == 1)
            hipLaunchKernelGGL(k_down_u8, grid, block, 0, c->stream, c->d_ref_u8, c->d_ref_u8,
                               o_ref, o_ref, iw, ih, ow, oh);
        else
            hipLaunchKernelGGL(k_down_f32, grid, block, 0, c->stream,
                               c->d_lin_ref + p.lin_off[s - 1], c->d_lin_ref + p.lin_off[s - 1],
                               o_ref, o_ref, iw, ih, ow, oh);
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // caller may free `ref` after return
    c->have_ref = true;
    c->ref_w = w;
    c->ref_h = h;
    return SSIMU2_OK;
}

int ssimu2_score_against_reference(ssimu2_ctx* c, const uint8_t* dist, double* out_score) {
    if (!c) return SSIMU2_ERR_INVALID_ARG;
    if (!c->have_ref) return c->fail(SSIMU2_ERR_NO_REFERENCE, "no reference set");
    if (!dist || !out_score) return c->fail(SSIMU2_ERR_INVALID_ARG, "null pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)c->ref_w * c->ref_h * 3;
    HIP_TRY(c, hipMemcpyAsync(c->d_dist_u8, dist, bytes, hipMemcpyHostToDevice, c->stream));
    int rc = enqueue_score(c, c->d_ref_u8, c->d_dist_u8, c->ref_w, c->ref_h, true);
    if (rc) return rc;
    return ssimu2_wait(c, out_score);
}

int ssimu2_last_averages(ssimu2_ctx* c, double* out, int* out_num_scales) {
    if (!c || !out) return SSIMU2_ERR_INVALID_ARG;
    if (c->pending) {
        int rc = ssimu2_wait(c, nullptr);
        if (rc) return rc;
    }
    memcpy(out, c->h_result, 108 * sizeof(double));
    if (out_num_scales) *out_num_scales = (int)c->h_result[109];
    return SSIMU2_OK;
}

int ssimu2_time_device(ssimu2_ctx* c, const void* d_ref, const void* d_dist, uint32_t w,
                       uint32_t h, int iters, float* out_ms_total, double* out_score) {
    int rc = check_args(c, d_ref, d_dist, w, h);
    if (rc) return rc;
    if (iters <= 0 || !out_ms_total) return c->fail(SSIMU2_ERR_INVALID_ARG, "bad iters/out");
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = ensure_capacity(c, w, h))) return rc;
    c->have_ref = false;
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < iters; ++i)
        if ((rc = enqueue_score(c, (const uint8_t*)d_ref, (const uint8_t*)d_dist, w, h, false)))
            return rc;
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    double score = 0.0;
    if ((rc = ssimu2_wait(c, &score))) return rc;
    HIP_TRY(c, hipEventSynchronize(c->ev1));
    HIP_TRY(c, hipEventElapsedTime(out_ms_total, c->ev0, c->ev1));
    if (out_score) *out_score = score;
    return SSIMU2_OK;
}

int ssimu2_time_scale_kernel(ssimu2_ctx* c, const void* d_ref, const void* d_dist, uint32_t w,
                             uint32_t h, int scale, int iters, float* out_ms_avg) {
    int rc = check_args(c, d_ref, d_dist, w, h);
    if (rc) return rc;
    if (iters <= 0 || !out_ms_avg) return c->fail(SSIMU2_ERR_INVALID_ARG, "bad iters/out");
    double score;
    if ((rc = ssimu2_score_rgb8_device(c, d_ref, d_dist, w, h, &score))) return rc;
    const Pyramid p = make_pyramid(w, h);
    if (scale < 0 || scale >= p.nscales) return c->fail(SSIMU2_ERR_INVALID_ARG, "bad scale");
    size_t poff = 0;
    for (int s = 0; s < scale; ++s) poff += (size_t)scale_blocks(c, p, s) * kStats;
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < iters; ++i)
        launch_scale(c, p, scale, (const uint8_t*)d_ref, (const uint8_t*)d_dist,
                     c->d_partials + poff);
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *out_ms_avg = ms / (float)iters;
    return SSIMU2_OK;
}

}  // extern "C"
