// SSIMULACRA2 with the PUBLISHED blur: the recursive Gaussian of libjxl's ssimulacra2 (three
// second-order sections driven by in[n-N-1] + in[n+N-1], N = 5 for sigma 1.5; Charalampidis 2016),
// applied to the materialised product planes, horizontally then vertically -- the optional
// SSIMU2_BLUR_RECURSIVE mode of a scorer context (include/ssimu2_hip.h).
//
// Why it exists: fssimu2's source is not available, the default kernels
// evaluate the recursion's 9-tap impulse response instead (k_march), and the two differ by the
// recursion's own rounding noise (median 0.5, up to 2.4 points at 4K).  This mode follows the
// published operation order exactly -- the CPU checker's OR_BLUR_IIR planes are reproduced bit for
// bit -- so a maintainer who can run fssimu2 can see which of the two it agrees with.
//
// A recursion cannot be cut into strips or segments: every output depends on the whole line
// before it.  Parallelism is therefore lines x planes only (15 planes per scale: x, y, xx, yy, xy
// of three channels), ~1 wave per SIMD at 4K, each lane a chain of 2,160-3,840 dependent steps:
// this mode is latency-bound by construction (about 9x the time of the default kernels at 4K)
// and is not what `bench.py` measures.
//
//   k_rg_h     lane = image row, streaming its own row 32 columns at a time as 16-byte loads and
//              stores (no transposition, no LDS); the products are formed on load.
//   k_rg_v     lane = image column: coalesced as it is; loads issued a batch of 10 rows ahead.
//   k_rg_maps  SSIM and edge-difference maps from the 15 blurred planes + the two XYB frames,
//              partial sums in the layout k_finalize reduces.
#pragma once

namespace ssimu2 {

constexpr int RG_N = 5;                // radius of the sigma-1.5 recursion: round(3.2795 sigma + 0.2546)
constexpr int RG_TILE = 32;            // columns per register tile of the horizontal pass
constexpr int RG_MAPS_BLOCKS = 256;    // partial-sum blocks per scale and channel triple

struct RgArgs {
    const float* xa;  // positive-XYB planes of the reference  [3][h][w]
    const float* xb;  // ... of the distorted frame
    float* hout;      // horizontal pass of the 15 planes     [15][h][w], plane = 5 * channel + kind
    float* vout;      // vertical pass of those
    int w, h;
};

// kind: 0 = x, 1 = y, 2 = x*x, 3 = y*y, 4 = x*y (the product rounded to fp32 first, as published)
__device__ __forceinline__ float rg_source(const float* a, const float* b, int kind, size_t i) {
    if (kind == 0) return a[i];
    if (kind == 1) return b[i];
    if (kind == 2) {
        const float v = a[i];
        return v * v;
    }
    if (kind == 3) {
        const float v = b[i];
        return v * v;
    }
    return a[i] * b[i];
}

// state of the three sections of one line
struct RgState {
    float p1[3], p2[3];  // previous and second-previous output of each section
};

// one step of the recursion (FastGaussian1D, scalar form):
//   o_k = n2_k * (left + right) - prev2_k - d1_k * prev_k;   out = (o_1 + o_3) + o_5
// FMA = false: every operation rounded by itself (the published scalar order);
// FMA = true:  the last multiply-subtract fused, fma(-d1_k, prev_k, .), as a compiler targeting
//              an FMA unit contracts it (SSIMU2_BLUR_RECURSIVE_FMA; the checker's OR_BLUR_IIR_FMA)
template <bool FMA>
__device__ __forceinline__ float rg_step(RgState& s, float left, float right, const float (&n2)[3],
                                         const float (&d1)[3]) {
    const float sum = left + right;
    float o[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v = sum * n2[k];
        v = v - s.p2[k];
        if (FMA) v = fmaf(-d1[k], s.p1[k], v);
        else v = v - d1[k] * s.p1[k];
        o[k] = v;
        s.p2[k] = s.p1[k];
        s.p1[k] = v;
    }
    return (o[0] + o[1]) + o[2];
}

// Horizontal pass.  One wave per (block of 64 rows, plane); grid = (ceil(h / 64), 15).
// lane = image row, and the lane streams ITS OWN row: 32 columns per tile as eight 16-byte loads
// (a lane's eight loads share one 128-byte line, the L1 serves the repeats) into registers, 32
// recursion steps, eight 16-byte stores -- no transposition, no LDS, no barrier, one pointer per
// lane.  The loads of tile t + 1 are issued before the steps of tile t (a lone wave per SIMD hides
// nothing by itself).  The last, partial tile of a row goes element by element.
typedef float rg_f4 __attribute__((ext_vector_type(4)));

template <bool FMA>
__global__ __launch_bounds__(64) void k_rg_h(RgArgs a) {
    const int lane = threadIdx.x;
    const int plane = blockIdx.y, ch = plane / 5, kind = plane - 5 * ch;
    const int w = a.w, h = a.h;
    const int row = blockIdx.x * 64 + lane;
    const bool live = row < h;
    const size_t n = (size_t)w * h;
    const size_t base = (size_t)min(row, h - 1) * w;  // idle lanes shadow the last row, never store
    const float* xa = a.xa + ch * n;
    const float* xb = a.xb + ch * n;
    // plane kinds as a pointer pair and a uniform flag: x, y, x*x, y*y, x*y
    const float* srcp = ((kind == 1 || kind == 3) ? xb : xa) + base;
    const float* srcq = (kind == 2 ? xa : xb) + base;
    const bool prod = kind >= 2;
    float* out = a.hout + plane * n + base;
    const float n2[3] = {c_k.rg_n2[0], c_k.rg_n2[1], c_k.rg_n2[2]};
    const float d1[3] = {c_k.rg_d1[0], c_k.rg_d1[1], c_k.rg_d1[2]};
    RgState st;
#pragma unroll
    for (int k = 0; k < 3; ++k) st.p1[k] = st.p2[k] = 0.f;
    float prev[2 * RG_N];  // the last ten inputs of the previous tile (zeros: the published padding)
#pragma unroll
    for (int k = 0; k < 2 * RG_N; ++k) prev[k] = 0.f;

    const int nfull = w / RG_TILE;                        // tiles that lie wholly inside the row
    const int ntiles = (w + (RG_N - 1) + RG_TILE - 1) / RG_TILE;  // m = n + 4 runs to w + 3
    float nxt[RG_TILE];
    // tile T of this lane's row into nxt[]: whole tiles as 16-byte loads, the rest element-wise
#define RG_LOAD_TILE(T)                                                                   \
    if ((T) < nfull) {                                                                    \
        const rg_f4* p4_ = reinterpret_cast<const rg_f4*>(srcp + (size_t)(T) * RG_TILE);  \
        const rg_f4* q4_ = reinterpret_cast<const rg_f4*>(srcq + (size_t)(T) * RG_TILE);  \
        _Pragma("unroll") for (int v = 0; v < RG_TILE / 4; ++v) {                         \
            rg_f4 pv_, qv_;                                                               \
            __builtin_memcpy(&pv_, p4_ + v, 16);                                          \
            __builtin_memcpy(&qv_, q4_ + v, 16);                                          \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                 \
                nxt[4 * v + e] = prod ? pv_[e] * qv_[e] : pv_[e];                         \
        }                                                                                 \
    } else {                                                                              \
        _Pragma("unroll") for (int mm = 0; mm < RG_TILE; ++mm) {                          \
            const int col_ = (T) * RG_TILE + mm;                                          \
            const int colc_ = min(col_, w - 1);                                           \
            const float pv_ = srcp[colc_], qv_ = srcq[colc_];                             \
            /* x * 1 and x * 0 are exact for the finite, non-negative XYB values; a select */ \
            /* here is turned into a branch around the load, one wait per load */        \
            nxt[mm] = (prod ? pv_ * qv_ : pv_) * (col_ < w ? 1.0f : 0.0f);               \
        }                                                                                 \
    }
    RG_LOAD_TILE(0)
    for (int t = 0; t < ntiles; ++t) {
        float r[RG_TILE];
#pragma unroll
        for (int mm = 0; mm < RG_TILE; ++mm) r[mm] = nxt[mm];
        if (t + 1 < ntiles) RG_LOAD_TILE(t + 1)
        // 32 steps: m = 32 t + mm is the right-hand input column, the output column is m - 4, so
        // the tile's outputs are columns 32 t - 4 .. 32 t + 27.
        float o[RG_TILE];
#pragma unroll
        for (int mm = 0; mm < RG_TILE; ++mm) {
            const float left = mm >= 2 * RG_N ? r[mm - 2 * RG_N] : prev[mm];
            o[mm] = rg_step<FMA>(st, left, r[mm], n2, d1);
        }
#pragma unroll
        for (int k = 0; k < 2 * RG_N; ++k) prev[k] = r[RG_TILE - 2 * RG_N + k];
        // o[] = output columns c0 .. c0 + 31 with c0 = 32 t - 4 (a multiple of 4: 16-byte stores)
        if (live) {
            const int c0 = t * RG_TILE - (RG_N - 1);  // first output column of this tile's o[]
            if (c0 >= 0 && c0 + RG_TILE <= w) {
                rg_f4* o4 = reinterpret_cast<rg_f4*>(out + c0);
#pragma unroll
                for (int v = 0; v < RG_TILE / 4; ++v) {
                    const rg_f4 val = {o[4 * v], o[4 * v + 1], o[4 * v + 2], o[4 * v + 3]};
                    __builtin_memcpy(o4 + v, &val, 16);
                }
            } else {
#pragma unroll
                for (int mm = 0; mm < RG_TILE; ++mm) {
                    const int col = c0 + mm;
                    if (col >= 0 && col < w) out[col] = o[mm];
                }
            }
        }
    }
#undef RG_LOAD_TILE
}

// Vertical pass of the 15 horizontally blurred planes.  lane = column; grid = (ceil(w / 64), 15).
template <bool FMA>
__global__ __launch_bounds__(64) void k_rg_v(RgArgs a) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int w = a.w, h = a.h;
    if (x >= w) return;
    const size_t n = (size_t)w * h;
    const float* in = a.hout + blockIdx.y * n + x;
    float* out = a.vout + blockIdx.y * n + x;
    const float n2[3] = {c_k.rg_n2[0], c_k.rg_n2[1], c_k.rg_n2[2]};
    const float d1[3] = {c_k.rg_d1[0], c_k.rg_d1[1], c_k.rg_d1[2]};
    RgState st;
#pragma unroll
    for (int k = 0; k < 3; ++k) st.p1[k] = st.p2[k] = 0.f;
    // Ten rows per batch: the left-hand input in[m - 10] of a step is the right-hand one of the
    // same slot of the previous batch, so every row is loaded once; the next batch's loads are in
    // flight under this one's steps.
    constexpr int U = 2 * RG_N;
    float right[U], left[U], nright[U];
#define RG_LOAD_ROWS(M0, R)                                          \
    _Pragma("unroll") for (int j = 0; j < U; ++j) {                  \
        const int m_ = (M0) + j; /* uniform */                       \
        R[j] = m_ < h ? in[(size_t)m_ * w] : 0.f;                    \
    }
#pragma unroll
    for (int j = 0; j < U; ++j) right[j] = 0.f;  // rows -10 .. -1: the published zero padding
    RG_LOAD_ROWS(0, nright)
    for (int m0 = 0; m0 < h + RG_N - 1; m0 += U) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            left[j] = right[j];
            right[j] = nright[j];
        }
        if (m0 + U < h + RG_N - 1) RG_LOAD_ROWS(m0 + U, nright)
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int m = m0 + j;
            if (m < h + RG_N - 1) {
                const float o = rg_step<FMA>(st, left[j], right[j], n2, d1);
                if (m >= RG_N - 1) out[(size_t)(m - (RG_N - 1)) * w] = o;
            }
        }
    }
#undef RG_LOAD_ROWS
}

// Maps + partial sums.  grid = (RG_MAPS_BLOCKS, 3 channels), 256 threads; part[stat][block].
__global__ __launch_bounds__(256) void k_rg_maps(RgArgs a, double* __restrict__ part) {
    __shared__ double s_part[4][6];
    const int ch = blockIdx.y, blk = blockIdx.x;
    const size_t n = (size_t)a.w * a.h;
    const size_t chunk = (n + RG_MAPS_BLOCKS - 1) / RG_MAPS_BLOCKS;
    const size_t lo = (size_t)blk * chunk, hi = lo + chunk < n ? lo + chunk : n;
    const float* v = a.vout + (size_t)ch * 5 * n;
    const float* xa = a.xa + ch * n;
    const float* xb = a.xb + ch * n;
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
        const float mu1 = v[i], mu2 = v[n + i], s11 = v[2 * n + i], s22 = v[3 * n + i], s12 = v[4 * n + i];
        const float r1 = xa[i], r2 = xb[i];
        const float mu11 = mu1 * mu1, mu22 = mu2 * mu2, mu12 = mu1 * mu2;
        const float dm = mu1 - mu2;
        const float num_m = fmaf(-dm, dm, 1.0f);
        const float num_s = fmaf(2.0f, s12 - mu12, kC2);
        const float denom_s = ((s11 - mu11) + (s22 - mu22)) + kC2;
        float d = 1.0f - div_rn(num_m * num_s, denom_s);
        d = fmaxf(d, 0.0f);
        const float ea = fabsf(r2 - mu2), eb = fabsf(r1 - mu1);
        const float e = div_rn(ea - eb, 1.0f + eb);  // == (1+ea)/(1+eb) - 1, no cancellation
        const float art = fmaxf(e, 0.0f), det = fmaxf(-e, 0.0f);
        const float d2 = d * d, a2 = art * art, t2 = det * det;
        acc[0] += (double)d;
        acc[1] += (double)(d2 * d2);
        acc[2] += (double)art;
        acc[3] += (double)(a2 * a2);
        acc[4] += (double)det;
        acc[5] += (double)(t2 * t2);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double s = wave_sum(acc[k]);
        if (lane == 0) s_part[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        const double s = ((s_part[0][k] + s_part[1][k]) + s_part[2][k]) + s_part[3][k];
        // stat index as k_finalize reads it: 0..5 ssim (c*2 + n), 6..17 edge (c*4 + j)
        const int stat = k < 2 ? ch * 2 + k : 6 + ch * 4 + (k - 2);
        part[(size_t)stat * RG_MAPS_BLOCKS + blk] = s;
    }
}

}  // namespace ssimu2
