// Device code of the MI355X (gfx950) SSIMULACRA2 scorer: included only by ssimu2_hip.hip.
//
// Kernels (wave64; VALU + LDS work, no MFMA: stencil and pointwise arithmetic):
//   k_pyramid_bands : linear-light 2x2 box pyramid, all five levels in one launch from one read
//                  of the u8 sRGB frames (through the LUT)
//   k_march      : ONE launch for all scales: per workgroup, a strip of 120 output columns of
//                  one scale is marched top to bottom: sRGB LUT -> opsin -> cbrt -> positive
//                  XYB (converter waves, LDS ring of (ref, dist) pair rows), horizontal 9-tap of
//                  {x, y, xx, yy, xy} in registers, vertical 9-tap from a 9-row register
//                  window, SSIM + edge-difference maps, fp64 partial sums
//   k_finalize   : fixed-order fp64 reduction of the partials, 108 averages, weighted sum,
//                  polynomial, score
//
// Arithmetic contract: this translation unit is compiled with -ffp-contract=off;
// every fused multiply-add is an explicit fmaf().  The sequence of IEEE operations per pixel
// is fixed and is the one the CPU checker evaluates, because the SSIM map cancels hard in
// fp32 (a 1-ulp difference upstream moves the score by ~1e-3).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ssimu2_hip.h"

namespace ssimu2 {

constexpr int kNumScales = SSIMU2_NUM_SCALES;
constexpr int kStats = SSIMU2_STATS_PER_SCALE;

// ---- constants of the published algorithm (DESIGN.md "Algorithm") -----------------------------
constexpr float kC2 = 0.0009f;
constexpr float kM00 = 0.30f, kM01 = 0.622f, kM02 = 0.078f;
constexpr float kM10 = 0.23f, kM11 = 0.692f, kM12 = 0.078f;
constexpr float kM20 = 0.24342268924547819f, kM21 = 0.20476744424496821f,
                kM22 = 0.55180986650955360f;
constexpr float kOpsinBias = 0.0037930732552754493f;

struct DevConst {
    float lut[256];     // 8-bit sRGB -> linear, fp32(rounded from fp64)
    float taps[5];      // FIR taps |d| = 0..4 of the sigma-1.5 recursive Gaussian
    float cbrt_bias;    // cbrt_repro(kOpsinBias)
    float rg_n2[3];     // the recursion itself (ssimu2_recursive.h): input gains of the three sections
    float rg_d1[3];     // ... and their feedback coefficients -2 cos(omega_k)
    double weights[108];
};
__constant__ DevConst c_k;

// ---- device helpers ---------------------------------------------------------------------------

// Cube root from IEEE mul/fma only (arguments are in [0, 2)): bit-trick seed for y = x^(-1/3), one third-order step
// y (1 + e/3 + 2e^2/9 + 14e^3/81) with e = 1 - x y^3, c = x y^2, one residual-corrected Newton
// step on c.  17 operations, max error 0.76 ulp.
__device__ __forceinline__ float cbrt_repro(float x) {
    uint32_t i = __float_as_uint(x);
    // i / 3 as one multiply-high: floor(i * 0x55555556 / 2^32) == i / 3 for every i < 2^31
    // (checked exhaustively), and the bits of a float below 2.0 are below 2^30
    i = 0x54A21D2Au - __umulhi(i, 0x55555556u);
    float y = __uint_as_float(i);
    float t = x * y;
    t = t * y;
    t = t * y;
    const float e = 1.0f - t;
    float p = fmaf(e, 14.0f / 81.0f, 2.0f / 9.0f);
    p = fmaf(p, e, 1.0f / 3.0f);
    p = p * e;
    y = fmaf(y, p, y);
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

// Same values as linear_to_xyb for non-negative linear inputs, which is all the marching kernel
// ever sees (LUT entries and their 2x2 averages are >= 0, so every opsin sum is >= the bias):
// the published clamp to zero and cbrt_repro's x > 0 guard can never act and are left out.
__device__ __forceinline__ float cbrt_repro_pos(float x) {
    uint32_t i = __float_as_uint(x);
    // i / 3 as one multiply-high: floor(i * 0x55555556 / 2^32) == i / 3 for every i < 2^31
    // (checked exhaustively), and the bits of a float below 2.0 are below 2^30
    i = 0x54A21D2Au - __umulhi(i, 0x55555556u);
    float y = __uint_as_float(i);
    float t = x * y;
    t = t * y;
    t = t * y;
    const float e = 1.0f - t;
    float p = fmaf(e, 14.0f / 81.0f, 2.0f / 9.0f);
    p = fmaf(p, e, 1.0f / 3.0f);
    p = p * e;
    y = fmaf(y, p, y);
    const float y2 = y * y;
    float c = x * y2;
    const float r = fmaf(c * c, c, -x);
    return fmaf(r, y2 * (-1.0f / 3.0f), c);
}

__device__ __forceinline__ void linear_to_xyb_pos(float r, float g, float b, float& X, float& Y,
                                                  float& B) {
    const float cb = c_k.cbrt_bias;
    const float l = cbrt_repro_pos(fmaf(kM00, r, fmaf(kM01, g, fmaf(kM02, b, kOpsinBias)))) - cb;
    const float m = cbrt_repro_pos(fmaf(kM10, r, fmaf(kM11, g, fmaf(kM12, b, kOpsinBias)))) - cb;
    const float s = cbrt_repro_pos(fmaf(kM20, r, fmaf(kM21, g, fmaf(kM22, b, kOpsinBias)))) - cb;
    const float x = 0.5f * (l - m), y = 0.5f * (l + m);
    B = (s - y) + 0.55f;
    X = fmaf(x, 14.0f, 0.42f);
    Y = y + 0.01f;
}

// symmetric 9-tap in the contract's operation order: one mul, four FMAs.
__device__ __forceinline__ float fir9(float c, float s1, float s2, float s3, float s4, float w0,
                                      float w1, float w2, float w3, float w4) {
    float acc = w0 * c;
    acc = fmaf(w1, s1, acc);
    acc = fmaf(w2, s2, acc);
    acc = fmaf(w3, s3, acc);
    acc = fmaf(w4, s4, acc);
    return acc;
}

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

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ---- linear-light pyramid, all five levels from the 8-bit frame in one pass -----------------------
// A BAND is 256 x 32 pixels of the 8-bit frame (aligned to 256 / 32, so every 2x2 source block
// of every level, clamped or not, lies inside it: 32 = 2^5) and is worked on by 512 threads, one
// 4 x 4 pixel block each:
//   level 1 (2 x 2 per thread) and level 2 (1 per thread) in registers -- per thread four rows of
//     12 contiguous bytes = three dword loads each, twelve loads in flight; a wave covers 768
//     contiguous bytes of each of its rows;
//   levels 3, 4, 5 (32 x 4, 16 x 2, 8 x 1 per band) through three small LDS tiles.
// out(ox,oy) = (((p00 + p01) + p10) + p11) * 0.25 with coordinates clamped to the last row /
// column of the level above (the published Downsample(in, 2, 2); the CPU checker's
// or_downsample2), level by level: the same operations in the same order as round 1's
// three-levels-per-launch kernel, whose planes these reproduce bit for bit.
struct PyrBandArgs {
    const uint8_t* in[2];  // frames (tight RGB8)
    float* out[2][5];      // per frame: planes [3][h_l][w_l] of levels 1..5 (entries >= nlevels unused)
    float* xyb0[2];        // XYB variant only: positive-XYB planes [3][h_0][w_0] of the frame itself
    int w[6], h[6];        // level dimensions, [0] = the 8-bit frame
    int opitch[6];         // floats per row of the OUTPUT planes of each level (= w[] for the linear pyramid; the
                           // recursive modes pad their rows to 128 floats, ssimu2_recursive.h "Row pitch")
    int nlevels;           // levels to produce: 1..5; 0 = nothing to do
    int bands_x, bands_y, nframes;
    unsigned* zero4;       // XYB variant: four job cursors of the recursive passes that follow in the stream, zeroed here
};

constexpr int PYR_THREADS = 512;                      // (256-thread bands of 128 pixels measured the same)
constexpr int PYR_TX = PYR_THREADS / 8;               // threads across a band (8 thread rows of 4 pixel rows)
constexpr int PYR_BAND_W = 4 * PYR_TX, PYR_BAND_H = 32;
constexpr int PYR_BAND_LDS_FLOATS = 3 * 8 * PYR_TX + 3 * 4 * (PYR_TX / 2) + 3 * 2 * (PYR_TX / 4);  // levels 2, 3, 4 tiles

__device__ __forceinline__ float box4(float p00, float p01, float p10, float p11) {
    float sum = p00;
    sum += p01;
    sum += p10;
    sum += p11;
    return sum * 0.25f;
}

// XYB variant (the recursive blur modes, which read nothing but XYB planes): the positive-XYB
// value of a linear-light pixel goes out instead of the pixel; the linear values stay in
// registers / LDS for the next level.  linear_to_xyb is what k_rg_xyb applies to the same values.
__device__ __forceinline__ void pyr_store_xyb(float* out, size_t n, size_t at, const float (&lin)[3]) {
    float X, Y, B;
    linear_to_xyb(lin[0], lin[1], lin[2], X, Y, B);
    out[at] = X;
    out[n + at] = Y;
    out[2 * n + at] = B;
}

// One level from an LDS tile of the level above: `src` is [3][sh][sw] (tile origin = global
// (2*ox0, 2*oy0) of the level above, whose size is wa x ha), outputs ox0+lx, oy0+ly.
template <bool XYB>
__device__ __forceinline__ void pyr_lds_level(const float* src, int sw, int sh, int wa, int ha, float* dst_tile,
                                              int dw_tile, int dh_tile, float* out, int wo, int ho, int po, int ox0,
                                              int oy0, int lx, int ly) {
    const int ox = ox0 + lx, oy = oy0 + ly;
    float v[3] = {0.f, 0.f, 0.f};
    if (ox < wo && oy < ho) {
        const int xa = 2 * lx, xb = min(2 * ox + 1, wa - 1) - 2 * ox0;
        const int ya = 2 * ly, yb = min(2 * oy + 1, ha - 1) - 2 * oy0;
        const size_t n = (size_t)po * ho;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* p = src + c * sw * sh;
            v[c] = box4(p[ya * sw + xa], p[ya * sw + xb], p[yb * sw + xa], p[yb * sw + xb]);
            if (!XYB) out[c * n + (size_t)oy * po + ox] = v[c];
        }
        if (XYB) pyr_store_xyb(out, n, (size_t)oy * po + ox, v);
    }
    if (dst_tile) {
#pragma unroll
        for (int c = 0; c < 3; ++c) dst_tile[(c * dh_tile + ly) * dw_tile + lx] = v[c];
    }
}

// `lut`: the sRGB table in LDS; `lds`: PYR_BAND_LDS_FLOATS floats of scratch; 512 threads, all of
// which must call (barriers inside).  `band` < bands_x * bands_y * nframes.
template <bool XYB>
__device__ __forceinline__ void pyramid_band(const PyrBandArgs& a, int band, const float* lut, float* lds) {
    const int t = threadIdx.x;
    const int per_frame = a.bands_x * a.bands_y;
    const int f = band / per_frame;
    const int r = band - f * per_frame;
    const int by = r / a.bands_x, bx = r - by * a.bands_x;
    const int w0 = a.w[0], h0 = a.h[0], w1 = a.w[1], h1 = a.h[1];
    constexpr int TX = PYR_TX;
    float* s2 = lds;                       // [3][8][TX]
    float* s3 = s2 + 3 * 8 * TX;           // [3][4][TX/2]
    float* s4 = s3 + 3 * 4 * (TX / 2);     // [3][2][TX/4]
    const int tx = t % TX, ty = t / TX;
    const int X0 = bx * PYR_BAND_W + 4 * tx, Y0 = by * PYR_BAND_H + 4 * ty;
    // ---- levels 1 and 2 in registers
    float l1[2][2][3];  // [row][col][channel] of this thread's 2 x 2 level-1 outputs
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) l1[j][i][c] = 0.f;
    if (X0 < w0 && Y0 < h0) {
        const uint8_t* base = a.in[f];
        uint32_t raw[4][3];
        const bool inside = X0 + 3 < w0 && Y0 + 3 < h0;
        if (inside) {  // four rows of 12 contiguous bytes
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint8_t* p = base + ((size_t)(Y0 + j) * w0 + X0) * 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) __builtin_memcpy(&raw[j][k], p + 4 * k, 4);
            }
        } else {  // frame border: clamped coordinates, byte by byte
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int y = min(Y0 + j, h0 - 1);
                uint8_t b[12];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint8_t* p = base + ((size_t)y * w0 + min(X0 + i, w0 - 1)) * 3;
                    b[3 * i] = p[0];
                    b[3 * i + 1] = p[1];
                    b[3 * i + 2] = p[2];
                }
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    raw[j][k] = (uint32_t)b[4 * k] | ((uint32_t)b[4 * k + 1] << 8) | ((uint32_t)b[4 * k + 2] << 16) |
                                ((uint32_t)b[4 * k + 3] << 24);
            }
        }
        // byte 3*i + c of a row = channel c of pixel i
#define PYR_LIN(j, i, c) lut[(raw[j][(3 * (i) + (c)) >> 2] >> (8 * ((3 * (i) + (c)) & 3))) & 255u]
        const int p1 = a.opitch[1];
        const size_t n1 = (size_t)p1 * h1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ox = (X0 >> 1) + i, oy = (Y0 >> 1) + j;
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    l1[j][i][c] = box4(PYR_LIN(2 * j, 2 * i, c), PYR_LIN(2 * j, 2 * i + 1, c),
                                       PYR_LIN(2 * j + 1, 2 * i, c), PYR_LIN(2 * j + 1, 2 * i + 1, c));
                if (ox < w1 && oy < h1 && (!XYB || a.nlevels >= 1)) {
                    float* o = a.out[f][0];
                    if (XYB) {
                        pyr_store_xyb(o, n1, (size_t)oy * p1 + ox, l1[j][i]);
                    } else {
#pragma unroll
                        for (int c = 0; c < 3; ++c) o[c * n1 + (size_t)oy * p1 + ox] = l1[j][i][c];
                    }
                }
            }
        }
        if (XYB) {  // the frame's own XYB planes: this thread's 4 x 4 pixels, a row of four at a time
            const int p0 = a.opitch[0];
            const size_t n0 = (size_t)p0 * h0;
            float* o = a.xyb0[f];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float px[3][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    linear_to_xyb(PYR_LIN(j, i, 0), PYR_LIN(j, i, 1), PYR_LIN(j, i, 2), px[0][i], px[1][i], px[2][i]);
                }
                if (inside) {
                    const size_t at = (size_t)(Y0 + j) * p0 + X0;
#pragma unroll
                    for (int c = 0; c < 3; ++c) __builtin_memcpy(o + c * n0 + at, px[c], 16);
                } else if (Y0 + j < h0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (X0 + i < w0) {
#pragma unroll
                            for (int c = 0; c < 3; ++c) o[c * n0 + (size_t)(Y0 + j) * p0 + X0 + i] = px[c][i];
                        }
                }
            }
        }
#undef PYR_LIN
    }
    if (a.nlevels < 2) return;
    {
        const int w2 = a.w[2], h2 = a.h[2];
        const int ox = X0 >> 2, oy = Y0 >> 2;  // = 64 bx + tx, 8 by + ty
        float v[3] = {0.f, 0.f, 0.f};
        if (ox < w2 && oy < h2) {
            // the level-1 column / row 2*ox+1, 2*oy+1 may not exist: the published clamp repeats the last one
            const int ib = min(2 * ox + 1, w1 - 1) - 2 * ox, jb = min(2 * oy + 1, h1 - 1) - 2 * oy;
            const int p2 = a.opitch[2];
            const size_t n2 = (size_t)p2 * h2;
            float* o = a.out[f][1];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float p01 = ib ? l1[0][1][c] : l1[0][0][c];
                const float p10 = jb ? l1[1][0][c] : l1[0][0][c];
                const float p11 = jb ? (ib ? l1[1][1][c] : l1[1][0][c]) : (ib ? l1[0][1][c] : l1[0][0][c]);
                v[c] = box4(l1[0][0][c], p01, p10, p11);
                if (!XYB) o[c * n2 + (size_t)oy * p2 + ox] = v[c];
            }
            if (XYB) pyr_store_xyb(o, n2, (size_t)oy * p2 + ox, v);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) s2[(c * 8 + ty) * TX + tx] = v[c];
    }
    if (a.nlevels < 3) return;
    __syncthreads();
    if (t < 4 * (TX / 2))
        pyr_lds_level<XYB>(s2, TX, 8, a.w[2], a.h[2], s3, TX / 2, 4, a.out[f][2], a.w[3], a.h[3], a.opitch[3], bx * (TX / 2), by * 4,
                      t % (TX / 2), t / (TX / 2));
    if (a.nlevels < 4) return;
    __syncthreads();
    if (t < 2 * (TX / 4))
        pyr_lds_level<XYB>(s3, TX / 2, 4, a.w[3], a.h[3], s4, TX / 4, 2, a.out[f][3], a.w[4], a.h[4], a.opitch[4], bx * (TX / 4), by * 2,
                      t % (TX / 4), t / (TX / 4));
    if (a.nlevels < 5) return;
    __syncthreads();
    if (t < TX / 8)
        pyr_lds_level<XYB>(s4, TX / 4, 2, a.w[4], a.h[4], nullptr, 0, 0, a.out[f][4], a.w[5], a.h[5], a.opitch[5], bx * (TX / 8), by, t, 0);
}

__global__ __launch_bounds__(PYR_THREADS) void k_pyramid_bands(PyrBandArgs a) {
    __shared__ float s_lut[256];
    __shared__ float s_tiles[PYR_BAND_LDS_FLOATS];
    if (threadIdx.x < 256) s_lut[threadIdx.x] = c_k.lut[threadIdx.x];
    __syncthreads();
    pyramid_band<false>(a, (int)blockIdx.x, s_lut, s_tiles);
}

// The same bands with positive-XYB planes of every level -- the frame's own included -- as the
// output instead of the linear-light levels: the one conversion launch of the recursive blur modes.
__global__ __launch_bounds__(PYR_THREADS) void k_pyramid_bands_xyb(PyrBandArgs a) {
    __shared__ float s_lut[256];
    __shared__ float s_tiles[PYR_BAND_LDS_FLOATS];
    if (threadIdx.x < 256) s_lut[threadIdx.x] = c_k.lut[threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x < 4 && a.zero4) a.zero4[threadIdx.x] = 0u;  // k_rg_h / k_rg_v start after this launch has ended
    __syncthreads();
    pyramid_band<true>(a, (int)blockIdx.x, s_lut, s_tiles);
}

// ---- fused per-scale kernel, marching form, all scales in one launch ----------------------------
// One workgroup (8 waves) owns a strip of MW output columns of ONE scale and marches down
// `seg` output rows, one image row per step.
//   waves 0-1 (converters): lane = one staged column (MW + 8 halo = 128), both frames.  Each
//     step they convert one input row (sRGB LUT at scale 0 -> opsin -> cbrt -> positive XYB)
//     into an LDS ring of raw rows, one barrier group ahead of the blur waves; their global
//     loads run one more group ahead, so HBM latency is off the critical path.  At scale 0 a
//     lane fetches its pixel with ONE dword load (the three bytes of the pixel plus one of the
//     next), the wave's loads covering a contiguous 193-byte run of the row.
//   waves 2-7 (blur + maps): two waves per XYB channel, lane = one output column.  Each step
//     a lane reads its 9-wide window of (x, y) = (ref, dist) pairs from the ring, forms the
//     products, does the horizontal 9-tap of the five planes {x, y, xx, yy, xy} in
//     registers and pushes the results into a 9-row register window, from which the
//     vertical 9-tap and the SSIM / edge-difference maps of the row four steps back are
//     evaluated and accumulated.  The row loop is unrolled nine times so the window is
//     addressed with compile-time indices (no register moves).
//     LDS latency is taken off the critical path without extra registers: the five inner
//     window pairs of the NEXT row are requested right after the horizontal pass of this row
//     (when the tap registers are dead) and land under the vertical pass and the maps; the
//     four outer pairs are requested at the top of the step and land under the partial sums
//     of the inner ones.
// One output pixel per lane keeps the window at 45 registers, and the two roles run separate
// loops (own register allocation): <= 80 VGPRs, 3 workgroups = 24 waves per CU.  A lone wave
// issues a VALU op only every ~4 cycles, so occupancy is what fills the SIMDs.
// The workgroup synchronises once per GROUP rows; in the blur waves the barrier sits between
// the horizontal pass of the group's last row and the prefetch for the next group's first.
// HBM traffic: each input pixel is read once per strip (+8/MW horizontal, +8/seg vertical
// halo); only 18 partial sums per workgroup are written.
// Scales are laid out largest first in the grid, so the short workgroups of the small scales
// fill the tail of the big ones instead of running as five latency-bound launches.
constexpr int RAD = 4;
constexpr int MW = 120;        // output columns per strip
constexpr int MRW = MW + 8;    // staged columns (4 px halo each side) = 128 = 2 waves per frame
constexpr int MHALF = MW / 2;  // output columns per blur wave (lanes 0..59 active)
constexpr int RING = 16;       // raw-row ring depth (power of two >= 13: rows t-4 .. t+8)
constexpr int AHEAD = 3;       // rows the converters run ahead of the blur waves (1 group)
constexpr int GROUP = 3;       // rows per barrier interval (divides the 9-phase unroll)
constexpr int MARCH_THREADS = 512;
constexpr int CONV_WAVES = 2;  // each converter lane handles one staged column of BOTH frames

// One ring element: the positive-XYB value of one channel of one staged pixel in both frames,
// .x = reference, .y = distorted.  Keeping the pair together makes every tap of the blur
// waves one naturally aligned ds_read_b64 (256 B/clk/CU; the planar layout needed
// ds_read2_b32, 128 B/clk/CU) and every converter store one ds_write_b64.
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int RING_ROW = 3 * MRW;  // f2 elements per ring row: [channel][column]

struct MarchPlan {
    int nscales;
    int blk_end[kNumScales];   // exclusive end of each scale's block range in the grid
    int w[kNumScales], h[kNumScales], seg[kNumScales], nstrips[kNumScales], nblocks[kNumScales];
    const void* ref[kNumScales];   // scale 0: u8 interleaved; others: fp32 planes
    const void* dist[kNumScales];
    const float* ref_xyb[kNumScales];  // cached positive-XYB planes of the reference, or null
    // blur(ref*ref) planes [3][h][w] of the scale: read by MARCH_REFBLUR (cached once per
    // search), written by MARCH_EMIT
    float* ref_s11[kNumScales];
    double* part[kNumScales];      // [18][nblocks] partial sums of the scale
};

// Kernel modes of the marching body (one __global__ entry each, so every mode has its own
// register allocation and the pair-score kernel is untouched by the others):
//   MARCH_PAIR     both frames blurred in flight (any pair; what `value` measures)
//   MARCH_REFBLUR  s11 = blur(ref*ref), which depends on the reference alone, comes from planes
//                  cached once per search (tq.zig:37 passes the same e.rgb on every pass); the blur
//                  waves keep four planes instead of five.  Same operations on the same operands,
//                  so the score's bits do not move.  mu1 = blur(ref) depends on the reference
//                  alone too, but caching it as well makes the kernel HBM-bound and slower
//                  (measured): one cached plane is the balance point.
//   MARCH_EMIT     writes that plane (run once by ssimu2_set_reference)
enum { MARCH_PAIR = 0, MARCH_REFBLUR = 1, MARCH_EMIT = 2 };

// Per-lane cursor of a blur wave over the cached / emitted reference blur planes of its channel:
// the pixel of this lane's column in the next output row, plus a prefetch queue (the values are
// loaded RB_AHEAD steps before the step that consumes them; HBM latency under load is several
// row steps).
#ifndef MARCH_RB_AHEAD
#define MARCH_RB_AHEAD 6   // round 4: 4 / 6 / 8 = 0.1621 / 0.1597 / 0.1636 ms per cached 4K pass (profiles/r04_refblur_ab.log)
#endif
#ifndef MARCH_NT_CACHED
#define MARCH_NT_CACHED 0   // 1: streaming (nontemporal) loads of the cached reference planes, read once per pass
#endif
constexpr int RB_AHEAD = MARCH_RB_AHEAD;
struct MarchRefBlur {
    float* s11;
    int pitch;       // elements per row
    int rows_left;   // rows of this segment not loaded yet
    bool active;     // lane owns an output column (lanes >= MHALF of a blur wave only shadow lane 0)
    float ps11[9];  // slot = phase of the consuming step; RB_AHEAD of them are live
};

// ---- converter waves ------------------------------------------------------------------------------
// opsin mix and the rest of linear_to_xyb_pos as two halves (same operations in the same order),
// so that the sRGB LUT reads of the next row can be issued between them.
__device__ __forceinline__ void opsin_mix(float r, float g, float b, float (&lms)[3]) {
    lms[0] = fmaf(kM00, r, fmaf(kM01, g, fmaf(kM02, b, kOpsinBias)));
    lms[1] = fmaf(kM10, r, fmaf(kM11, g, fmaf(kM12, b, kOpsinBias)));
    lms[2] = fmaf(kM20, r, fmaf(kM21, g, fmaf(kM22, b, kOpsinBias)));
}
__device__ __forceinline__ void mixed_to_xyb_pos(const float (&lms)[3], float (&v)[3]) {
    const float cb = c_k.cbrt_bias;
    const float l = cbrt_repro_pos(lms[0]) - cb;
    const float m = cbrt_repro_pos(lms[1]) - cb;
    const float s = cbrt_repro_pos(lms[2]) - cb;
    const float x = 0.5f * (l - m), y = 0.5f * (l + m);
    v[2] = (s - y) + 0.55f;
    v[0] = fmaf(x, 14.0f, 0.42f);
    v[1] = y + 0.01f;
}

// Per-lane read cursor of a converter lane over one frame: address of this lane's
// (column-clamped) pixel in the next row to load, advanced by one row pitch per load.
//   u8 frames (scale 0): ONE unaligned dword load per pixel -- R | G << 8 | B << 16 | the next
//     pixel's R << 24 -- so a wave's loads cover one contiguous 193-byte run of the row.  The
//     lane of the image's last column reads one byte earlier and shifts (`shift` = 8), so no
//     load ever reaches behind the frame's last byte.
//   fp32 planes (scales >= 1, cached reference XYB): three dword loads, `plane` elements apart.
struct MarchCursor {
    const uint8_t* base;  // uniform: first byte of the frame / of the first plane
    uint32_t off;         // per lane: byte offset of this lane's pixel inside a row
    size_t plane;         // bytes between planes (fp32 sources)
    int pitch;            // bytes per row
    uint32_t shift;       // u8 only
};

template <bool U8>
__device__ __forceinline__ MarchCursor march_cursor(const void* base, int w, int h, int gxc) {
    MarchCursor c;
    c.plane = (size_t)w * h * 4;
    c.pitch = U8 ? w * 3 : w * 4;
    const bool last_col = U8 && gxc == w - 1;
    c.shift = last_col ? 8u : 0u;
    c.base = (const uint8_t*)base;
    c.off = (uint32_t)gxc * (U8 ? 3u : 4u) - (last_col ? 1u : 0u);
    return c;
}

// Load this lane's pixel of image row `row` (uniform; clamped into the image by the caller, so
// the load is unconditional: every row issues the same number of loads and the compiler's
// s_waitcnt vmcnt(N) can count them -- a conditional load made it fall back to vmcnt(0), which
// shortened the prefetch distance to one row).
template <bool U8, bool NT = false>
__device__ __forceinline__ void march_load(uint32_t (&raw)[3], const MarchCursor& c, int row) {
    const uint8_t* p = c.base + (size_t)(uint32_t)row * (uint32_t)c.pitch;
    if (U8) {
        uint32_t d;
        __builtin_memcpy(&d, p + c.off, 4);
        raw[0] = d;
    } else if (NT) {
        raw[0] = __builtin_nontemporal_load((const uint32_t*)(p + c.off));
        raw[1] = __builtin_nontemporal_load((const uint32_t*)(p + c.plane + c.off));
        raw[2] = __builtin_nontemporal_load((const uint32_t*)(p + 2 * c.plane + c.off));
    } else {
        raw[0] = *(const uint32_t*)(p + c.off);
        raw[1] = *(const uint32_t*)(p + c.plane + c.off);
        raw[2] = *(const uint32_t*)(p + 2 * c.plane + c.off);
    }
}

// MARCH_LUT_COPIES (experiment): 4 = four interleaved copies of the table,
// lane & 3 selects the copy.  Measured: no fewer bank conflicts and no faster (a ds_read_b32 serves
// 32 lanes from 32 banks whichever copy a lane reads); the product uses 1.
#ifndef MARCH_LUT_COPIES
#define MARCH_LUT_COPIES 1
#endif
__device__ __forceinline__ void march_lut(const float* lut, uint32_t d, uint32_t shift, float (&lin)[3]) {
    d >>= shift;
    if (MARCH_LUT_COPIES == 1) {
        lin[0] = lut[d & 255u];
        lin[1] = lut[(d >> 8) & 255u];
        lin[2] = lut[(d >> 16) & 255u];
    } else {
        const uint32_t k = threadIdx.x & (MARCH_LUT_COPIES - 1);
        lin[0] = lut[(d & 255u) * MARCH_LUT_COPIES + k];
        lin[1] = lut[((d >> 8) & 255u) * MARCH_LUT_COPIES + k];
        lin[2] = lut[((d >> 16) & 255u) * MARCH_LUT_COPIES + k];
    }
}

// The converter role of one workgroup: rows 0 .. steps-1 of the segment into the ring, GROUP
// rows per barrier interval, global loads GROUP rows ahead of their use.
//   U8    the frames are 8-bit sRGB (scale 0); otherwise fp32 linear-light planes
//   MODE  MARCH_PAIR: both frames converted.  MARCH_REFBLUR: frame 0 is the cached positive-XYB
//         planes of the reference (no arithmetic).  MARCH_EMIT: the same, and frame 1 is not needed.
// Per row and lane: [LUT values of this row, requested one row earlier] -> opsin mix of both
// frames -> request the LUT values of the next row -> cube roots, XYB, one ds_write_b64 per
// channel.  The LUT reads therefore land under ~100 arithmetic instructions.
template <bool U8, int MODE>
__device__ __forceinline__ void march_convert_rows(f2 (*ring)[3][MRW], const float* lut,
                                                   const MarchPlan& plan, int sc, int w, int h, int x0,
                                                   int y0, int steps, int ngroups, int col) {
    constexpr bool CACHED = MODE != MARCH_PAIR;
    constexpr bool TWO = MODE != MARCH_EMIT;
    constexpr bool LUT0 = U8 && !CACHED;  // frame 0 goes through the sRGB LUT
    const int gx = x0 - RAD + col;
    const bool col_ok = gx >= 0 && gx < w;
    const bool all_cols = x0 - RAD >= 0 && x0 - RAD + MRW <= w;  // uniform: no lane outside the image
    const int gxc = min(max(gx, 0), w - 1);  // clamped: the value is discarded when gx is outside
    const MarchCursor c0 = CACHED ? march_cursor<false>(plan.ref_xyb[sc], w, h, gxc)
                                  : march_cursor<U8>(plan.ref[sc], w, h, gxc);
    const MarchCursor c1 = march_cursor<U8>(plan.dist[sc], w, h, gxc);
    int load_row = y0 - RAD;  // image row of the next load
    // raw[f][j]: loaded, not yet converted row of frame f.  The queue is QD = 2 * GROUP rows deep
    // (the LUT reads of a row are requested one row before it is converted, so its pixels must
    // have landed by then: 4.5 rows = several microseconds of flight, enough for an HBM miss
    // under load); slot j is refilled every QD rows, the loop body is unrolled QD times, so the
    // queue rotates without register moves.
    constexpr int NG = 2;  // groups in flight (1, 3 and 4 measured the same: what mattered was the counted wait)
    constexpr int QD = NG * GROUP;
    uint32_t raw0[QD][3], raw1[QD][3];
    float lin0[3] = {0.f, 0.f, 0.f}, lin1[3] = {0.f, 0.f, 0.f};  // LUT values of the next row to convert
    // A row outside the image (only in the halo of the image's first and last segments) loads the
    // nearest image row instead and is converted like any other; its values are then replaced by
    // the blur's zero padding together with the columns outside the image.  So the row body is
    // straight-line code: no per-row branch around the loads or the arithmetic.
#define MARCH_LOAD(J)                                                      \
    {                                                                      \
        const int lrow_ = min(max(load_row, 0), h - 1); /* uniform */      \
        if (CACHED) march_load<false, MARCH_NT_CACHED != 0>(raw0[J], c0, lrow_); \
        else march_load<U8>(raw0[J], c0, lrow_);                           \
        if (TWO) march_load<U8>(raw1[J], c1, lrow_);                       \
        ++load_row;                                                        \
    }
#define MARCH_LUT(J)                                                             \
    if (U8) {                                                                    \
        if (LUT0) march_lut(lut, raw0[J][0], c0.shift, lin0);                    \
        if (TWO) march_lut(lut, raw1[J][0], c1.shift, lin1);                     \
    }
#define MARCH_PUT(J, R)                                                                        \
    {                                                                                          \
        /* is ring row R inside the image?  (uniform, from scalars: keeps the test on the */   \
        /* scalar unit instead of a flag carried in a vector register) */                      \
        const bool rok_ = (unsigned)(y0 - RAD + (R)) < (unsigned)h;                            \
        float a_[3] = {0.f, 0.f, 0.f}, b_[3] = {0.f, 0.f, 0.f};                                \
        float m0_[3], m1_[3];                                                                  \
        if (!CACHED) {                                                                         \
            if (U8) opsin_mix(lin0[0], lin0[1], lin0[2], m0_);                                 \
            else opsin_mix(__uint_as_float(raw0[J][0]), __uint_as_float(raw0[J][1]),           \
                           __uint_as_float(raw0[J][2]), m0_);                                  \
        }                                                                                      \
        if (TWO) {                                                                             \
            if (U8) opsin_mix(lin1[0], lin1[1], lin1[2], m1_);                                 \
            else opsin_mix(__uint_as_float(raw1[J][0]), __uint_as_float(raw1[J][1]),           \
                           __uint_as_float(raw1[J][2]), m1_);                                  \
        }                                                                                      \
        if (U8) {                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            MARCH_LUT((J + 1) % QD)                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                 \
        }                                                                                      \
        if (CACHED) {                                                                          \
            _Pragma("unroll") for (int c_ = 0; c_ < 3; ++c_) a_[c_] = __uint_as_float(raw0[J][c_]); \
        } else {                                                                               \
            mixed_to_xyb_pos(m0_, a_);                                                         \
        }                                                                                      \
        if (TWO) mixed_to_xyb_pos(m1_, b_);                                                    \
        if (!(all_cols && rok_)) { /* uniform; the asm keeps it a branch, not 12 selects */    \
            asm volatile("; zero padding");                                                    \
            const bool keep_ = col_ok && rok_;                                                 \
            _Pragma("unroll") for (int c_ = 0; c_ < 3; ++c_) {                                 \
                a_[c_] = keep_ ? a_[c_] : 0.0f;                                                \
                b_[c_] = keep_ ? b_[c_] : 0.0f;                                                \
            }                                                                                  \
        }                                                                                      \
        _Pragma("unroll") for (int c_ = 0; c_ < 3; ++c_)                                       \
            ring[(R) & (RING - 1)][c_][col] = f2{a_[c_], b_[c_]};                              \
    }
#pragma unroll
    for (int j = 0; j < QD; ++j) MARCH_LOAD(j)  // rows 0 .. QD-1
    MARCH_LUT(0)
    // group g produces ring rows 3g .. 3g+2 (the blur waves consume them in their group g, one
    // barrier later) and loads rows 3g+6 .. 3g+8; ngroups + 1 barriers in all, the last group
    // only joins the barrier.  A partial last group converts up to two rows behind the segment
    // into ring slots nobody reads any more.
#define MARCH_GROUP(G, J0)                                   \
    if ((G) * GROUP < steps) {                               \
        MARCH_PUT(J0 + 0, (G) * GROUP + 0)                   \
        MARCH_LOAD(J0 + 0)                                   \
        MARCH_PUT(J0 + 1, (G) * GROUP + 1)                   \
        MARCH_LOAD(J0 + 1)                                   \
        MARCH_PUT(J0 + 2, (G) * GROUP + 2)                   \
        MARCH_LOAD(J0 + 2)                                   \
    }
    static_assert(GROUP == 3, "MARCH_GROUP is written for three rows per barrier interval");
#pragma unroll 1
    for (int g = 0; g <= ngroups; g += 2) {
        MARCH_GROUP(g, 0)
        __syncthreads();
        if (g + 1 <= ngroups) {
            MARCH_GROUP(g + 1, GROUP)
            __syncthreads();
        }
    }
#undef MARCH_GROUP
#undef MARCH_LOAD
#undef MARCH_LUT
#undef MARCH_PUT
}

// Positive-XYB planes of one frame at one scale (run once per search for the reference, whose
// pixels are the same on every pass: tq.zig:37 passes the same e.rgb, main.zig:86).
__global__ __launch_bounds__(256) void k_ref_xyb(const void* __restrict__ in, bool u8, int w, int h,
                                                 float* __restrict__ out) {
    const size_t n = (size_t)w * h;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float r, g, b;
    if (u8) {
        const uint8_t* p = (const uint8_t*)in + i * 3;
        r = c_k.lut[p[0]];
        g = c_k.lut[p[1]];
        b = c_k.lut[p[2]];
    } else {
        const float* p = (const float*)in + i;
        r = p[0];
        g = p[n];
        b = p[2 * n];
    }
    float X, Y, B;
    linear_to_xyb(r, g, b, X, Y, B);
    out[i] = X;
    out[n + i] = Y;
    out[2 * n + i] = B;
}

// One (ref, dist) pair as it sits in the ring, fetched with one ds_read_b64 and split into its
// two 32-bit halves (sub-registers: no instruction).  Kept as a 64-bit integer rather than a
// float vector so that the compiler does not pair up the x / y arithmetic into v_pk_*_f32,
// which ties values to even-aligned register pairs the 80-VGPR budget has no room for.
struct MarchPair {
    unsigned long long u;
    __device__ __forceinline__ float x() const { return __uint_as_float((uint32_t)u); }
    __device__ __forceinline__ float y() const { return __uint_as_float((uint32_t)(u >> 32)); }
};
// volatile: one ds_read_b64 per tap.  The compiler would otherwise fuse neighbouring taps into
// ds_read2_b64, which the LDS serves at half the rate of two ds_read_b64 (measured: +3 %).
typedef __attribute__((address_space(3))) volatile unsigned long long lds_vu64;

__device__ __forceinline__ const lds_vu64* march_row(const lds_vu64* rp, int t) {
    return rp + (t & (RING - 1)) * RING_ROW;
}

// Horizontal pass of ring row t into window slot P: the lane's 9-wide window of (x, y) pairs
// (staged columns o .. o+8, centre o+4), nine ds_read_b64.
// plain planes: pair sums of the taps; product planes: the products are formed inside the
// pair sums as fma(a-, b-, a+ * b+) -- one rounding and one operation fewer than two
// products and an add (arithmetic contract, mirrored by the CPU checker)
template <int P, int MODE>
__device__ __forceinline__ void march_h(const lds_vu64* rp, float (&win)[5][9], int t, float w0,
                                        float w1, float w2, float w3, float w4) {
    const lds_vu64* row = march_row(rp, t);
    MarchPair v[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) v[q].u = row[q];
#define H9(f) \
    fir9(v[4].f(), v[3].f() + v[5].f(), v[2].f() + v[6].f(), v[1].f() + v[7].f(), v[0].f() + v[8].f(), \
         w0, w1, w2, w3, w4)
#define H9P(a, b)                                                                                      \
    fir9(v[4].a() * v[4].b(), fmaf(v[3].a(), v[3].b(), v[5].a() * v[5].b()),                           \
         fmaf(v[2].a(), v[2].b(), v[6].a() * v[6].b()), fmaf(v[1].a(), v[1].b(), v[7].a() * v[7].b()), \
         fmaf(v[0].a(), v[0].b(), v[8].a() * v[8].b()), w0, w1, w2, w3, w4)
    win[0][P] = H9(x);
    if (MODE != MARCH_REFBLUR) win[2][P] = H9P(x, x);
    if (MODE != MARCH_EMIT) {
        win[1][P] = H9(y);
        win[3][P] = H9P(y, y);
        win[4][P] = H9P(x, y);
    }
#undef H9
#undef H9P
}

// Second half of step t: vertical pass and maps of output row t - 4 (window slot of row t - j
// is (P - j) mod 9).  `edge`: the strip reaches past the image's right edge, so lanes whose
// column is outside (`!ok`) must not contribute (uniform; interior strips skip the selects).
template <int P, int MODE>
__device__ __forceinline__ void march_v(const lds_vu64* rp, float (&win)[5][9], float (&acc)[6], int t,
                                        bool ok, bool edge, float w0, float w1, float w2, float w3,
                                        float w4, MarchRefBlur& rb) {
    float c_s11 = 0.f;
    if (MODE == MARCH_REFBLUR) {
        // consume the value loaded RB_AHEAD steps ago, then load the row RB_AHEAD steps ahead
        c_s11 = rb.ps11[P];
        if (t >= 8 - RB_AHEAD && rb.rows_left > 0) {  // uniform: that output row exists
            rb.ps11[(P + RB_AHEAD) % 9] = ok ? (MARCH_NT_CACHED ? __builtin_nontemporal_load(rb.s11) : *rb.s11) : 0.0f;
            rb.s11 += rb.pitch;
            --rb.rows_left;
        }
    }
    if (t < 8) return;  // window not full yet (uniform across the workgroup)
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        if (MODE == MARCH_REFBLUR && k == 2) continue;
        if (MODE == MARCH_EMIT && k != 2) continue;
        const float* q = win[k];
        v[k] = fir9(q[(P + 5) % 9], q[(P + 4) % 9] + q[(P + 6) % 9],
                    q[(P + 3) % 9] + q[(P + 7) % 9], q[(P + 2) % 9] + q[(P + 8) % 9],
                    q[(P + 1) % 9] + q[P], w0, w1, w2, w3, w4);
    }
    if (MODE == MARCH_EMIT) {  // the reference's blur(ref*ref) plane, one row per step
        if (ok && rb.active) *rb.s11 = v[2];
        rb.s11 += rb.pitch;
        return;
    }
    MarchPair ctr;
    ctr.u = march_row(rp, t - 4)[RAD];  // centre pixel of the output row
    const float r1 = ctr.x(), r2 = ctr.y();
    const float mu1 = v[0], mu2 = v[1];
    const float s11 = MODE == MARCH_REFBLUR ? c_s11 : v[2], s22 = v[3], s12 = v[4];
    const float mu11 = mu1 * mu1, mu22 = mu2 * mu2, mu12 = mu1 * mu2;
    const float dm = mu1 - mu2;
    const float num_m = fmaf(-dm, dm, 1.0f);
    const float num_s = fmaf(2.0f, s12 - mu12, kC2);
    const float denom_s = ((s11 - mu11) + (s22 - mu22)) + kC2;
    float d = 1.0f - div_rn(num_m * num_s, denom_s);
    d = fmaxf(d, 0.0f);
    const float ea = fabsf(r2 - mu2), eb = fabsf(r1 - mu1);
    float e = div_rn(ea - eb, 1.0f + eb);  // == (1+ea)/(1+eb) - 1, no cancellation
    if (edge) {  // uniform; the asm keeps it a branch
        asm volatile("; right image edge");
        d = ok ? d : 0.0f;  // column inside the image?
        e = ok ? e : 0.0f;
    }
    const float art = fmaxf(e, 0.0f), det = fmaxf(-e, 0.0f);
    const float d2 = d * d, a2 = art * art, t2 = det * det;
    acc[0] += d;
    acc[1] += d2 * d2;
    acc[2] += art;
    acc[3] += a2 * a2;
    acc[4] += det;
    acc[5] += t2 * t2;
}

// XCD-aware tile order.  Workgroups are dealt to the eight XCDs round-robin by blockIdx (each XCD has its own
// 4 MiB L2), while neighbouring strips of one segment share their 8 halo columns and -- strips being 120 columns,
// not a multiple of a 128-byte line -- the cache lines their edges straddle.  With tile = blockIdx those neighbours
// sit on DIFFERENT XCDs and every XCD fetches its own copy.  MARCH_XCD_ORDER = 1 hands each XCD a contiguous run of
// tiles instead: among the workgroups [first, end) of a scale, the ones with blockIdx % 8 == x take the x-th run,
// in blockIdx order.  A bijection on [0, end - first); which workgroup computes a tile does not enter the tile's
// arithmetic, and the partial sums are stored under the TILE index: scores keep their bits.
#ifndef MARCH_XCD_ORDER
#define MARCH_XCD_ORDER 1
#endif
__device__ __forceinline__ int march_tile_of_block(int b, int first, int end) {
    if (!MARCH_XCD_ORDER) return b - first;
    // cnt(n, r) = how many k in [0, n) have k % 8 == r
    const int x = b & 7;
    int before = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int n_r = (end + 7 - r) / 8 - (first + 7 - r) / 8;  // workgroups of this scale on XCD phase r
        before += r < x ? n_r : 0;
    }
    return before + (b >> 3) - ((first + 7 - x) >> 3);  // + its rank among those of phase x
}

template <int MODE>
__device__ __forceinline__ void march_body(const MarchPlan& plan) {
    // [row slot][channel][column] of (ref, dist) pairs
    __shared__ __attribute__((aligned(16))) f2 s_ring[RING][3][MRW];
    __shared__ float s_lut[256 * MARCH_LUT_COPIES];
    __shared__ double s_part[6][6];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // which scale / strip / segment is this workgroup?
    int sc = 0, first = 0;
#pragma unroll
    for (int s = 0; s < kNumScales - 1; ++s)
        if (s + 1 < plan.nscales && (int)blockIdx.x >= plan.blk_end[s]) {
            sc = s + 1;
            first = plan.blk_end[s];
        }
    const int blk = march_tile_of_block((int)blockIdx.x, first, plan.blk_end[sc]);
    const int w = plan.w[sc], h = plan.h[sc], seg_rows = plan.seg[sc];
    const int nstrips = plan.nstrips[sc];
    const int by = blk / nstrips, bx = blk - by * nstrips;
    const bool u8 = sc == 0;
    const int x0 = bx * MW;
    const int y0 = by * seg_rows;
    const int rows_out = min(seg_rows, h - y0);
    const int steps = rows_out + 2 * RAD;  // input rows y0-4 .. y0+rows_out+3
    if (u8) {
        for (int i = tid; i < 256 * MARCH_LUT_COPIES; i += MARCH_THREADS) s_lut[i] = c_k.lut[i / MARCH_LUT_COPIES];
    }
    __syncthreads();

    const float w0 = c_k.taps[0], w1 = c_k.taps[1], w2 = c_k.taps[2], w3 = c_k.taps[3],
                w4 = c_k.taps[4];
    const bool is_conv = wave < CONV_WAVES;
    // blur state (fp32 sums: at most seg <= 160 terms per lane before the fp64 reduce)
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int hw = wave - CONV_WAVES;
    const int ch = hw >> 1;
    const bool hv_active = lane < MHALF;
    const int o = (hw & 1) * MHALF + (hv_active ? lane : 0);
    const bool ok = x0 + o < w;
    MarchRefBlur rb;
    rb.pitch = w;
    rb.rows_left = rows_out;
    rb.active = hv_active;
#pragma unroll
    for (int k = 0; k < 9; ++k) rb.ps11[k] = 0.f;
    {
        // this lane's pixel in output row y0 of its channel's planes (column clamped: lanes
        // outside the image never dereference it)
        const size_t at = ((size_t)max(ch, 0) * h + y0) * (size_t)w + (size_t)min(x0 + o, w - 1);
        rb.s11 = MODE == MARCH_PAIR ? nullptr : plan.ref_s11[sc] + at;
    }

    // Input row j (= image row y0-4+j) lives in ring slot j & (RING-1).  While the blur waves
    // consume rows 3I..3I+2 the converters fill rows 3I+AHEAD..3I+AHEAD+2.  The two roles run
    // separate loops with the same number of barriers (one per group of GROUP rows), so each
    // gets its own register allocation instead of carrying the other role's state.
    const int ngroups = (steps + GROUP - 1) / GROUP;
    if (is_conv) {
        // The converter waves carry the heavier per-row stream on their SIMDs; a static
        // issue-priority bump lets them keep pace (measured: -7 %).
        __builtin_amdgcn_s_setprio(1);
        const int col = (wave << 6) + lane;  // staged column; this lane converts both frames
        if (u8) march_convert_rows<true, MODE>(s_ring, s_lut, plan, sc, w, h, x0, y0, steps, ngroups, col);
        else march_convert_rows<false, MODE>(s_ring, s_lut, plan, sc, w, h, x0, y0, steps, ngroups, col);
    } else {
        float win[5][9];
        const lds_vu64* rp = (const lds_vu64*)&s_ring[0][ch][o];  // staged columns o .. o+8, centre o+4
        const bool edge = x0 + MW > w;  // uniform: some lanes of this strip are outside the image
        __syncthreads();
        // step t: horizontal pass of ring row t; [barrier after the group's last row]; vertical
        // pass + maps of output row t - 4
#define MARCH_STEP(P)                                                                          \
    {                                                                                          \
        const int t = t0 + P;                                                                  \
        if (t < steps) march_h<P, MODE>(rp, win, t, w0, w1, w2, w3, w4);                       \
        if ((P % GROUP) == GROUP - 1 && t - (GROUP - 1) < steps) __syncthreads();              \
        if (t < steps) march_v<P, MODE>(rp, win, acc, t, ok, edge, w0, w1, w2, w3, w4, rb);    \
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
    }

    if (MODE == MARCH_EMIT) return;  // planes written, nothing to reduce
    // the two half-strip waves of a channel each publish their sums; combined in fixed order
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
        plan.part[sc][(size_t)tid * plan.nblocks[sc] + blk] = s_part[2 * c][k] + s_part[2 * c + 1][k];
    }
}

// launch bound: 3 workgroups of 8 waves per CU = 6 waves per SIMD (<= 80 VGPRs)
__global__ __launch_bounds__(MARCH_THREADS, 6) void k_march(MarchPlan plan) {
    march_body<MARCH_PAIR>(plan);
}

// the per-pass kernel of a search: reference XYB and blur(ref*ref) planes cached
__global__ __launch_bounds__(MARCH_THREADS, 6) void k_march_refblur(MarchPlan plan) {
    march_body<MARCH_REFBLUR>(plan);
}

// once per search: blur(ref*ref) of every scale into plan.ref_s11
__global__ __launch_bounds__(MARCH_THREADS, 6) void k_ref_blur(MarchPlan plan) {
    march_body<MARCH_EMIT>(plan);
}

// ---- final reduction ------------------------------------------------------------------------------
struct FinalizeArgs {
    const double* part[kNumScales];
    int nblocks[kNumScales];
    double inv_pixels[kNumScales];
    int nscales;
};

// result layout: [0..107] averages [scale][18], [108] score, [109] nscales.  `result` is the context's page-locked
// host mirror: the kernel writes the 880 bytes over the bus itself (round 5: no device-side copy of them and no D2H
// copy command per score; -1.7 us per pair score, bits equal).
// 8 lanes per (scale, stat) item stride over the workgroup partials; lane-local sums, then a
// fixed-order 8-lane shuffle tree: deterministic, and 128 items run in parallel instead of 16.
// A launch of its own by measurement (round 5, profiles/r05_finalize_ab.log): folded into the last workgroup of the
// marching launch (an atomic done-counter behind a device-scope release) it made k_march 20 us SLOWER at 4K -- every
// one of its ~1000 workgroups pays an L2 write-back for the release -- and the recursive pass 2.5 us slower.
__global__ __launch_bounds__(1024) void k_finalize(FinalizeArgs fa, double* __restrict__ result) {
    __shared__ double s_avg[kNumScales * kStats];
    const int item = threadIdx.x >> 3, sub = threadIdx.x & 7;
    double v = 0.0;
    const int scale = item / kStats, stat = item - scale * kStats;
    const bool live = item < kNumScales * kStats && scale < fa.nscales;
    if (live) {
        // each lane sums runs of 8 consecutive partials: the 8 loads of a run are independent,
        // so the loop is 8x shorter than one dependent load + add per partial
        const double* p = fa.part[scale] + (size_t)stat * fa.nblocks[scale];
        const int nb = fa.nblocks[scale];
        for (int b = sub * 8; b < nb; b += 64) {
            double t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = b + k < nb ? p[b + k] : 0.0;
            v += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        }
    }
    v += __shfl_down(v, 4, 8);
    v += __shfl_down(v, 2, 8);
    v += __shfl_down(v, 1, 8);
    if (sub == 0 && item < kNumScales * kStats) {
        if (live) {
            v *= fa.inv_pixels[scale];
            if (stat & 1) v = sqrt(sqrt(v));  // odd stats are L4 norms
        }
        s_avg[item] = v;
    }
    __syncthreads();
    // published Score(): weights are consumed with a running index over (channel, scale present,
    // norm, {ssim, artifact, detail}); term j of that walk is evaluated by thread j and the
    // terms are summed with a fixed shuffle tree (two waves), then by thread 0.
    __shared__ double s_red[2];
    if (threadIdx.x < 128) {
        const int j = threadIdx.x;
        const int nterms = 3 * fa.nscales * 2 * 3;
        double term = 0.0;
        if (j < nterms) {
            const int k = j % 3, n = (j / 3) & 1, cs = j / 6;
            const int sc = cs % fa.nscales, c = cs / fa.nscales;
            const double* a = s_avg + sc * kStats;
            const double val = k == 0 ? a[c * 2 + n] : a[6 + c * 4 + n + (k == 2 ? 2 : 0)];
            term = c_k.weights[j] * fabs(val);
        }
        term = wave_sum(term);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = term;
    }
    __syncthreads();
    // the averages leave as two full-wave stores of consecutive doubles (they cross PCIe: not 108 scattered ones)
    if (threadIdx.x < kNumScales * kStats) result[threadIdx.x] = s_avg[threadIdx.x];
    if (threadIdx.x == 0) {
        double ssim = s_red[0] + s_red[1];
        ssim = ssim * 0.9562382616834844;
        ssim = 2.326765642916932 * ssim - 0.020884521182843837 * ssim * ssim +
               6.248496625763138e-05 * ssim * ssim * ssim;
        if (ssim > 0.0) ssim = 100.0 - 10.0 * pow(ssim, 0.6276336467831387);
        else ssim = 100.0;
        result[108] = ssim;
        result[109] = (double)fa.nscales;
    }
}

// ---------------------------------------------------------------------------------------------
// Decoded-frame hand-off (io.zig:654-663): libavif's RGB / RGBA rows `pitch` bytes apart ->
// tight RGB8.  kFast: four RGBA pixels per lane, four dword loads and three dword stores
// (needs w % 4 == 0, pitch % 4 == 0 and dword-aligned buffers); otherwise one pixel per lane
// with byte accesses.  Pure byte movement, HBM-bound: (ch + 3) bytes per pixel.
// ---------------------------------------------------------------------------------------------
template <bool kFast>
__global__ __launch_bounds__(256) void k_unpack_rgb(const uint8_t* __restrict__ src, uint32_t pitch,
                                                    uint32_t ch, uint32_t w, uint32_t h,
                                                    uint8_t* __restrict__ dst) {
    const uint32_t y = blockIdx.y;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (y >= h) return;
    if (kFast) {
        if (i * 4u >= w) return;
        const uint32_t* s = reinterpret_cast<const uint32_t*>(src + (size_t)y * pitch) + i * 4u;
        const uint32_t p0 = s[0], p1 = s[1], p2 = s[2], p3 = s[3];  // LE: R | G<<8 | B<<16 | A<<24
        uint32_t* d = reinterpret_cast<uint32_t*>(dst + ((size_t)y * w + i * 4u) * 3u);
        d[0] = (p0 & 0x00FFFFFFu) | (p1 << 24);
        d[1] = ((p1 >> 8) & 0x0000FFFFu) | (p2 << 16);
        d[2] = ((p2 >> 16) & 0x000000FFu) | (p3 << 8);
    } else {
        if (i >= w) return;
        const uint8_t* s = src + (size_t)y * pitch + (size_t)i * ch;
        uint8_t* d = dst + ((size_t)y * w + i) * 3u;
        d[0] = s[0];
        d[1] = s[1];
        d[2] = s[2];
    }
}

}  // namespace ssimu2
