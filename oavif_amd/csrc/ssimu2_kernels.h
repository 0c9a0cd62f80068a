// Device code of the MI355X (gfx950) SSIMULACRA2 scorer: included only by ssimu2_hip.hip.
//
// Kernels (wave64; VALU + LDS work, no MFMA: stencil and pointwise arithmetic):
//   k_pyramid    : linear-light 2x2 box pyramid, up to three levels per launch from one read
//                  of the input level (u8 sRGB frames through the LUT, or fp32 planes)
//   k_march      : ONE launch for all scales: per workgroup, a strip of 120 output columns of
//                  one scale is marched top to bottom: sRGB LUT -> opsin -> cbrt -> positive
//                  XYB (converter waves, LDS ring of raw rows), horizontal 9-tap of
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
    double weights[108];
};
__constant__ DevConst c_k;

// ---- device helpers ---------------------------------------------------------------------------

// Cube root from IEEE mul/fma only: bit-trick seed for y = x^(-1/3), one third-order step
// y (1 + e/3 + 2e^2/9 + 14e^3/81) with e = 1 - x y^3, c = x y^2, one residual-corrected Newton
// step on c.  17 operations, max error 0.76 ulp.
__device__ __forceinline__ float cbrt_repro(float x) {
    uint32_t i = __float_as_uint(x);
    i = 0x54A21D2Au - i / 3u;
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
    i = 0x54A21D2Au - i / 3u;
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

// ---- linear-light pyramid ----------------------------------------------------------------------
// out(ox,oy) = (((p00 + p01) + p10) + p11) * 0.25 with coordinates clamped to the last
// row/column of the level above (the published Downsample(in, 2, 2)).  One workgroup reads a
// 64x32 tile of the input level once and emits the 32x16, 16x8 and 8x4 tiles of the next
// three levels (tiles are aligned to powers of two, so every 2x2 source block, clamped or
// not, lies inside the tile).  blockIdx.z selects the frame.
// A workgroup covers 32 x 16 outputs of the first produced level (9 KB of LDS, <= 32 VGPRs):
// small enough to be co-resident with three k_march workgroups on a CU (3 x 49.3 KB of the
// 160 KB LDS, 480 of 512 VGPRs per SIMD), so that with two streams the HBM-bound pyramid of one
// score really runs under the VALU-bound marching kernel of another.
struct PyramidArgs {
    const void* in[2];   // per frame: u8 interleaved RGB (level 0) or fp32 planes [3][h][w]
    float* out[2][3];    // per frame, per produced level: fp32 planes; null = not produced
    int w[4], h[4];      // w[0],h[0] = input level; w[k],h[k] = k-th produced level
    int nlevels;         // 1..3 levels to produce
};

constexpr int PYR_TILE_H = 16;  // rows of the first produced level per workgroup (multiple of 8)

template <bool kU8>
__global__ __launch_bounds__(256) void k_pyramid(PyramidArgs a) {
    __shared__ float s1[3][PYR_TILE_H][33];
    __shared__ float s2[3][PYR_TILE_H / 2][17];
    __shared__ float s_lut[256];
    const int tid = threadIdx.x;
    const int f = blockIdx.z;
    if (kU8) s_lut[tid] = c_k.lut[tid];
    if (kU8) __syncthreads();
    const int w0 = a.w[0], h0 = a.h[0], w1 = a.w[1], h1 = a.h[1];
    const int tx0 = blockIdx.x * 32, ty0 = blockIdx.y * PYR_TILE_H;  // tile origin at level +1
    const size_t n0 = (size_t)w0 * h0, n1 = (size_t)w1 * h1;
    // level +1: 32 x PYR_TILE_H outputs, PYR_TILE_H / 8 per thread
#pragma unroll
    for (int j = 0; j < PYR_TILE_H / 8; ++j) {
        const int lx = tid & 31, ly = (tid >> 5) + 8 * j;
        const int ox = tx0 + lx, oy = ty0 + ly;
        float v[3] = {0.f, 0.f, 0.f};
        if (ox < w1 && oy < h1) {
            const int xa = 2 * ox, xb = min(2 * ox + 1, w0 - 1);
            const int ya = 2 * oy, yb = min(2 * oy + 1, h0 - 1);
            if (kU8) {
                const uint8_t* base = (const uint8_t*)a.in[f];
                const uint8_t* p00 = base + ((size_t)ya * w0 + xa) * 3;
                const uint8_t* p01 = base + ((size_t)ya * w0 + xb) * 3;
                const uint8_t* p10 = base + ((size_t)yb * w0 + xa) * 3;
                const uint8_t* p11 = base + ((size_t)yb * w0 + xb) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float sum = s_lut[p00[c]];
                    sum += s_lut[p01[c]];
                    sum += s_lut[p10[c]];
                    sum += s_lut[p11[c]];
                    v[c] = sum * 0.25f;
                }
            } else {
                const float* base = (const float*)a.in[f];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float* p = base + c * n0;
                    float sum = p[(size_t)ya * w0 + xa];
                    sum += p[(size_t)ya * w0 + xb];
                    sum += p[(size_t)yb * w0 + xa];
                    sum += p[(size_t)yb * w0 + xb];
                    v[c] = sum * 0.25f;
                }
            }
            float* o = a.out[f][0];
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c * n1 + (size_t)oy * w1 + ox] = v[c];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) s1[c][ly][lx] = v[c];
    }
    if (a.nlevels < 2) return;
    __syncthreads();
    // level +2: 16 x PYR_TILE_H/2 outputs, one per thread
    const int w2 = a.w[2], h2 = a.h[2];
    if (tid < 16 * (PYR_TILE_H / 2)) {
        const int lx = tid & 15, ly = tid >> 4;
        const int ox = (tx0 >> 1) + lx, oy = (ty0 >> 1) + ly;
        float v[3] = {0.f, 0.f, 0.f};
        if (ox < w2 && oy < h2) {
            // local coordinates inside s1; the clamp is against the level +1 image size
            const int xa = 2 * lx, xb = min(2 * ox + 1, w1 - 1) - tx0;
            const int ya = 2 * ly, yb = min(2 * oy + 1, h1 - 1) - ty0;
            const size_t n2 = (size_t)w2 * h2;
            float* o = a.out[f][1];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float sum = s1[c][ya][xa];
                sum += s1[c][ya][xb];
                sum += s1[c][yb][xa];
                sum += s1[c][yb][xb];
                v[c] = sum * 0.25f;
                o[c * n2 + (size_t)oy * w2 + ox] = v[c];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) s2[c][ly][lx] = v[c];
    }
    if (a.nlevels < 3) return;
    __syncthreads();
    // level +3: 8 x PYR_TILE_H/4 outputs
    if (tid < 8 * (PYR_TILE_H / 4)) {
        const int w3 = a.w[3], h3 = a.h[3];
        const int lx = tid & 7, ly = tid >> 3;
        const int ox = (tx0 >> 2) + lx, oy = (ty0 >> 2) + ly;
        if (ox < w3 && oy < h3) {
            const int xa = 2 * lx, xb = min(2 * ox + 1, w2 - 1) - (tx0 >> 1);
            const int ya = 2 * ly, yb = min(2 * oy + 1, h2 - 1) - (ty0 >> 1);
            const size_t n3 = (size_t)w3 * h3;
            float* o = a.out[f][2];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float sum = s2[c][ya][xa];
                sum += s2[c][ya][xb];
                sum += s2[c][yb][xa];
                sum += s2[c][yb][xb];
                o[c * n3 + (size_t)oy * w3 + ox] = sum * 0.25f;
            }
        }
    }
}

// ---- fused per-scale kernel, marching form, all scales in one launch ----------------------------
// One workgroup (8 waves) owns a strip of MW output columns of ONE scale and marches down
// `seg` output rows, one image row per step.
//   waves 0-1 (converters): lane = one staged column (MW + 8 halo = 128), both frames.  Each
//     step they convert one input row (sRGB LUT at scale 0 -> opsin -> cbrt -> positive XYB)
//     into an LDS ring of raw rows, one barrier group ahead of the blur waves; their global
//     loads run one more group ahead, so HBM latency is off the critical path.
//   waves 2-7 (blur + maps): two waves per XYB channel, lane = one output column.  Each step
//     a lane reads its 9-wide window of x (ref) and y (dist) from the ring, forms the
//     products, does the horizontal 9-tap of the five planes {x, y, xx, yy, xy} in
//     registers and pushes the results into a 9-row register window, from which the
//     vertical 9-tap and the SSIM / edge-difference maps of the row four steps back are
//     evaluated and accumulated.  The row loop is unrolled nine times so the window is
//     addressed with compile-time indices (no register moves).
// One output pixel per lane keeps the window at 45 registers, and the two roles run separate
// loops (own register allocation): <= 80 VGPRs, 3 workgroups = 24 waves per CU.  A lone wave
// issues a VALU op only every ~4 cycles, so occupancy is what fills the SIMDs.
// The workgroup synchronises once per GROUP rows.
// HBM traffic: each input pixel is read once per strip (+8/MW horizontal, +8/seg vertical
// halo); only 18 partial sums per workgroup are written.
// Scales are laid out largest first in the grid, so the short workgroups of the small scales
// fill the tail of the big ones instead of running as five latency-bound launches.
constexpr int RAD = 4;
constexpr int MW = 120;        // output columns per strip
constexpr int MRW = MW + 8;    // staged columns (4 px halo each side) = 128 = 2 waves per frame
constexpr int MHALF = MW / 2;  // output columns per blur wave (lanes 0..59 active)
constexpr int RING = 16;       // raw-row ring depth (power of two >= 13: rows t-4 .. t+8)
#ifdef EXP_AHEAD
constexpr int AHEAD = EXP_AHEAD;
#else
constexpr int AHEAD = 3;       // rows the converters run ahead of the blur waves (1 group)
#endif
constexpr int GROUP = 3;       // rows per barrier interval (divides the 9-phase unroll)
constexpr int MARCH_THREADS = 512;
constexpr int CONV_WAVES = 2;  // each converter lane handles one staged column of BOTH frames

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
#ifndef RB_AHEAD
#define RB_AHEAD 4
#endif
struct MarchRefBlur {
    float* s11;
    int pitch;       // elements per row
    int rows_left;   // rows of this segment not loaded yet
    bool active;     // lane owns an output column (lanes >= MHALF of a blur wave only shadow lane 0)
    float ps11[9];  // slot = phase of the consuming step; RB_AHEAD of them are live
};

// Raw values of one staged pixel of one frame of an input row (u8 codes or fp32 bits).
struct MarchRaw {
    uint32_t v[3];
    bool ok;
};

// Per-lane read cursor over one frame: address of this lane's (column-clamped) pixel in the
// next row to load; advanced by one row pitch per load, so no per-row 64-bit address math.
struct MarchSrc {
    const uint8_t* p;  // u8: interleaved RGB; otherwise fp32 planes
    size_t plane;      // plane stride in elements (fp32 sources)
    int pitch;         // bytes per row
    bool u8;
};

__device__ __forceinline__ MarchSrc march_src(const void* base, bool u8, int w, int h, int gx,
                                              int first_row) {
    MarchSrc s;
    const int gxc = min(max(gx, 0), w - 1);  // clamped: the value is discarded when gx is outside
    s.u8 = u8;
    s.plane = (size_t)w * h;
    s.pitch = u8 ? w * 3 : w * 4;
    s.p = (const uint8_t*)base + (ptrdiff_t)first_row * s.pitch + (size_t)gxc * (u8 ? 3 : 4);
    return s;
}

// Issue the global loads of the cursor's row and advance it.  `row_ok` (the row is inside the
// image) is uniform over the workgroup; rows outside are never dereferenced.
__device__ __forceinline__ void march_load(MarchRaw& raw, MarchSrc& s, bool row_ok, bool col_ok) {
    raw.ok = row_ok && col_ok;
    if (row_ok) {
        if (s.u8) {
            raw.v[0] = s.p[0];
            raw.v[1] = s.p[1];
            raw.v[2] = s.p[2];
        } else {
            const uint32_t* q = (const uint32_t*)s.p;
            raw.v[0] = q[0];
            raw.v[1] = q[s.plane];
            raw.v[2] = q[2 * s.plane];
        }
    } else {
        raw.v[0] = raw.v[1] = raw.v[2] = 0;
    }
    s.p += s.pitch;
}

// Convert the loaded pixel to positive XYB and store it into ring slot `slot` of frame k
// (zeros outside the image: the blur is zero padded).
__device__ __forceinline__ void march_convert(float (*ring)[3][2][MRW], const float* lut, bool u8,
                                              const MarchRaw& raw, int slot, int k, int col) {
    float rr, gg, bb, v[3];
    if (u8) {
        rr = lut[raw.v[0]];
        gg = lut[raw.v[1]];
        bb = lut[raw.v[2]];
    } else {
        rr = __uint_as_float(raw.v[0]);
        gg = __uint_as_float(raw.v[1]);
        bb = __uint_as_float(raw.v[2]);
    }
#ifdef ABL_NOCONV
    v[0] = rr; v[1] = gg; v[2] = bb;
#else
    linear_to_xyb_pos(rr, gg, bb, v[0], v[1], v[2]);
#endif
#pragma unroll
    for (int c = 0; c < 3; ++c) ring[slot][c][k][col] = raw.ok ? v[c] : 0.0f;
}

// Cached reference: the loaded values already are positive XYB (k_ref_xyb); store them.
__device__ __forceinline__ void march_store_xyb(float (*ring)[3][2][MRW], const MarchRaw& raw, int slot,
                                                int k, int col) {
#pragma unroll
    for (int c = 0; c < 3; ++c) ring[slot][c][k][col] = raw.ok ? __uint_as_float(raw.v[c]) : 0.0f;
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

// LDS reads of one blur-wave step: the 9-wide x / y windows of ring row t and the centre pixel
// of the output row (t - 4).  Issued one step ahead of their use (see march_hv_step).
struct MarchTaps {
    float x[9], y[9], r1, r2;
};

__device__ __forceinline__ void march_hv_fetch(MarchTaps& m, float (*ring)[3][2][MRW], int t, int ch,
                                               int o) {
    const int slot = t & (RING - 1);
    const float* px = &ring[slot][ch][0][o];  // staged columns o .. o+8, centre o+4
    const float* py = &ring[slot][ch][1][o];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        m.x[q] = px[q];
        m.y[q] = py[q];
    }
    const int cslot = (t - 4) & (RING - 1);
    m.r1 = ring[cslot][ch][0][o + RAD];
    m.r2 = ring[cslot][ch][1][o + RAD];
}

template <int P, int MODE>
__device__ __forceinline__ void march_hv_step(float (*ring)[3][2][MRW], float (&win)[5][9],
                                              float (&acc)[6], int t, int ch, int o, bool ok,
                                              float w0, float w1, float w2, float w3, float w4,
                                              MarchRefBlur& rb) {
    MarchTaps cur;
    march_hv_fetch(cur, ring, t, ch, o);
    // plain planes: pair sums of the taps; product planes: the products are formed inside the
    // pair sums as fma(a-, b-, a+ * b+) -- one rounding and one operation fewer than two
    // products and an add (arithmetic contract, mirrored by the CPU checker)
#define H9(v) \
    fir9(v[4], v[3] + v[5], v[2] + v[6], v[1] + v[7], v[0] + v[8], w0, w1, w2, w3, w4)
#define H9P(a, b)                                                                              \
    fir9(a[4] * b[4], fmaf(a[3], b[3], a[5] * b[5]), fmaf(a[2], b[2], a[6] * b[6]),            \
         fmaf(a[1], b[1], a[7] * b[7]), fmaf(a[0], b[0], a[8] * b[8]), w0, w1, w2, w3, w4)
    win[0][P] = H9(cur.x);
    if (MODE != MARCH_REFBLUR) win[2][P] = H9P(cur.x, cur.x);
    if (MODE != MARCH_EMIT) {
        win[1][P] = H9(cur.y);
        win[3][P] = H9P(cur.y, cur.y);
        win[4][P] = H9P(cur.x, cur.y);
    }
#undef H9
#undef H9P
    const float r1 = cur.r1, r2 = cur.r2;
    float c_s11 = 0.f;
    if (MODE == MARCH_REFBLUR) {
        // consume the value loaded RB_AHEAD steps ago, then load the row RB_AHEAD steps ahead
        c_s11 = rb.ps11[P];
        if (t >= 8 - RB_AHEAD && rb.rows_left > 0) {  // uniform: that output row exists
            rb.ps11[(P + RB_AHEAD) % 9] = ok ? *rb.s11 : 0.0f;
            rb.s11 += rb.pitch;
            --rb.rows_left;
        }
    }
#ifdef ABL_NOVMAPS
    if (t >= 8) { acc[0] += win[0][(P + 5) % 9] + win[1][P] + win[2][P] + win[3][P] + win[4][P]; }
    if (t < 0)
#else
    if (t >= 8)
#endif
    {  // window full (uniform across the workgroup)
        // vertical 9-tap for the row four steps back: row t-j sits in window slot (P-j) mod 9
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

// Timing-only ablation switches (scripts/ablate.sh); never defined in the product build.
#ifdef ABL_NOBARRIER
#define MARCH_BARRIER() ((void)0)
#else
#define MARCH_BARRIER() __syncthreads()
#endif

template <int MODE>
__device__ __forceinline__ void march_body(const MarchPlan& plan) {
    // [row slot][channel][frame][column]: the x (ref) and y (dist) windows of a channel are 512 B
    // apart, inside the 8-bit dword offset of one ds_read2 base register
    __shared__ __attribute__((aligned(16))) float s_ring[RING][3][2][MRW];
    __shared__ float s_lut[256];
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
    const int blk = (int)blockIdx.x - first;
    const int w = plan.w[sc], h = plan.h[sc], seg_rows = plan.seg[sc];
    const int nstrips = plan.nstrips[sc];
    const int by = blk / nstrips, bx = blk - by * nstrips;
    const bool u8 = sc == 0;
    const int x0 = bx * MW;
    const int y0 = by * seg_rows;
    const int rows_out = min(seg_rows, h - y0);
    const int steps = rows_out + 2 * RAD;  // input rows y0-4 .. y0+rows_out+3
    if (u8 && tid < 256) s_lut[tid] = c_k.lut[tid];
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
    // consume rows 3I..3I+2 the converters fill rows 3I+AHEAD..3I+AHEAD+2.  The two roles run separate loops with the same
    // number of barriers (one per group of GROUP rows), so each gets its own register
    // allocation instead of carrying the other role's state.
    const int ngroups = (steps + GROUP - 1) / GROUP;
    if (is_conv) {
        // The converter waves carry the heavier per-row stream on their SIMDs (~225 vs ~180
        // instructions); a static issue-priority bump lets them keep pace (measured: -7 %).
#ifndef EXP_CONV_PRIO
#define EXP_CONV_PRIO 1
#endif
        __builtin_amdgcn_s_setprio(EXP_CONV_PRIO);
        const int col = (wave << 6) + lane;  // staged column; this lane converts both frames
        const int gx = x0 - RAD + col;
        const bool col_ok = gx >= 0 && gx < w;
        // reference frame: cached positive-XYB planes (fp32, no conversion) when available
        const bool ref_cached = plan.ref_xyb[sc] != nullptr;
        MarchSrc src0 = march_src(ref_cached ? (const void*)plan.ref_xyb[sc] : plan.ref[sc],
                                  u8 && !ref_cached, w, h, gx, y0 - RAD);
        MarchSrc src1 = march_src(plan.dist[sc], u8, w, h, gx, y0 - RAD);
        int load_row = y0 - RAD;  // image row the cursors point at
#define MARCH_LOAD2(Q0, Q1)                                          \
    {                                                                \
        const bool row_ok = load_row >= 0 && load_row < h;           \
        march_load(Q0, src0, row_ok, col_ok);                        \
        march_load(Q1, src1, row_ok, col_ok);                        \
        ++load_row;                                                  \
    }
#define MARCH_PUT2(Q0, Q1, SLOT)                                            \
    {                                                                       \
        if (ref_cached) march_store_xyb(s_ring, Q0, SLOT, 0, col);          \
        else march_convert(s_ring, s_lut, src0.u8, Q0, SLOT, 0, col);       \
        march_convert(s_ring, s_lut, u8, Q1, SLOT, 1, col);                 \
    }
        // q[f][j]: loaded, not yet converted row of frame f; slot j is refilled every GROUP rows,
        // so the GROUP-deep prefetch queue rotates with the unrolled group (no register moves)
        MarchRaw q[2][GROUP];
#pragma unroll
        for (int j0 = 0; j0 < AHEAD; j0 += GROUP) {  // prologue: ring rows 0 .. AHEAD-1
#pragma unroll
            for (int j = 0; j < GROUP; ++j) MARCH_LOAD2(q[0][j], q[1][j])
#pragma unroll
            for (int j = 0; j < GROUP; ++j) MARCH_PUT2(q[0][j], q[1][j], j0 + j)
        }
#pragma unroll
        for (int j = 0; j < GROUP; ++j) MARCH_LOAD2(q[0][j], q[1][j])  // rows AHEAD .. AHEAD+2
        MARCH_BARRIER();
#pragma unroll 1
        for (int g = 0; g < ngroups; ++g) {
#pragma unroll
            for (int j = 0; j < GROUP; ++j) {
                const int r = g * GROUP + j + AHEAD;  // ring row to produce (uniform)
                if (r < steps) {
                    MARCH_PUT2(q[0][j], q[1][j], r & (RING - 1))
                    MARCH_LOAD2(q[0][j], q[1][j])  // row r + GROUP, consumed next iteration
                }
            }
            MARCH_BARRIER();
        }
#undef MARCH_LOAD2
#undef MARCH_PUT2
    } else {
        float win[5][9];
        MARCH_BARRIER();
#define MARCH_STEP(P)                                                                     \
    {                                                                                     \
        const int t = t0 + P;                                                             \
        if (t < steps)                                                                    \
            march_hv_step<P, MODE>(s_ring, win, acc, t, ch, o, ok, w0, w1, w2, w3, w4, rb); \
        if ((P % GROUP) == GROUP - 1 && t - (GROUP - 1) < steps) MARCH_BARRIER();         \
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

// result layout: [0..107] averages [scale][18], [108] score, [109] nscales
// 8 lanes per (scale, stat) item stride over the workgroup partials; lane-local sums, then a
// fixed-order 8-lane shuffle tree: deterministic, and 128 items run in parallel instead of 16.
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
        result[item] = v;
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

// ---------------------------------------------------------------------------------------------
// Read-stream probe (ssimu2_measure_read_stream): every lane reads 16 bytes per step, four steps
// in flight; the xor of everything read is stored only if it is a value the zeroed
// buffer cannot produce, so the loads are kept and nothing is written.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_read_stream(const uint4* __restrict__ src, size_t n16,
                                                     uint32_t* __restrict__ sink) {
    // one contiguous chunk per workgroup (whole DRAM pages per workgroup), lanes 16 B apart
    const size_t chunk = (n16 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * chunk;
    const size_t hi = lo + chunk < n16 ? lo + chunk : n16;
    size_t i = lo + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (; i + 768 < hi; i += 1024) {
        const uint4 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
        acc.x ^= a.x ^ b.x ^ c.x ^ d.x;
        acc.y ^= a.y ^ b.y ^ c.y ^ d.y;
        acc.z ^= a.z ^ b.z ^ c.z ^ d.z;
        acc.w ^= a.w ^ b.w ^ c.w ^ d.w;
    }
    for (; i < hi; i += 256) {
        const uint4 a = src[i];
        acc.x ^= a.x; acc.y ^= a.y; acc.z ^= a.z; acc.w ^= a.w;
    }
    const uint32_t v = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (v == 0x9E3779B9u) *sink = v;
}

}  // namespace ssimu2
