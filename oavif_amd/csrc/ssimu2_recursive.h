// SSIMULACRA2 with the PUBLISHED blur: the recursive Gaussian of libjxl's ssimulacra2 (three
// second-order sections driven by in[n-N-1] + in[n+N-1], N = 5 for sigma 1.5; Charalampidis 2016),
// applied to the materialised product planes, horizontally then vertically -- the
// SSIMU2_BLUR_RECURSIVE / _FMA modes of a scorer context (include/ssimu2_hip.h).
//
// Why it exists: fssimu2's source is not available, the default kernels
// evaluate the recursion's 9-tap impulse response instead (k_march), and the two differ by the
// recursion's own rounding noise (median 0.5, up to 2.4 points at 4K).  This
// mode follows the published operation order exactly -- the CPU checker's OR_BLUR_IIR planes are
// reproduced bit for bit -- so whichever form fssimu2 follows can be had on the device.
//
// A recursion cannot be cut into strips or segments: every output depends on the whole line
// before it, so the parallelism is lines x planes x sections and nothing else.  Layout since round 3
// (all six scales in ONE launch per stage, largest scale first; round 4: rows of every plane padded to
// 128 floats, products formed once in k_rg_h's staging, k_rg_v persistent with one workgroup per CU):
//
//   k_pyramid_bands_xyb (ssimu2_kernels.h)  positive-XYB planes of one frame at every scale,
//              straight from its bytes: the band pyramid with XYB outputs, no linear level stored.
//   k_rg_h     horizontal pass.  One LINE = THREE LANES, one second-order section each; the
//              published sum (o1 + o3) + o5 is two DPP adds (row_shr:5) per step.  A wave holds
//              4 DPP rows x 5 lines = 20 image rows (15 of 16 lanes busy); a workgroup is the
//              planes of one channel over the same 20 rows, one wave per plane ({y, yy, xy} of
//              a pass, {x, xx} of the reference), so the shared inputs are fetched once.  Each
//              wave stages its 20 rows 64 columns at a time through wave-private LDS with coalesced
//              16-byte loads and stores (see "Staging" below); products are formed once per element
//              while the tile is staged, rounded to fp32 first as published.  7 VALU instructions per
//              step instead of 15.
//   k_rg_v     vertical pass + maps.  lane = image column (coalesced rows), the three sections in
//              the lane; batches of ten rows, so that the left-hand inputs of a batch are the
//              right-hand ones of the previous batch; loads three batches ahead, streaming
//              (nontemporal: every plane is read once per pass).  The blurred
//              rows go to a double-buffered LDS tile, where five more waves of the workgroup
//              turn them -- with the cached reference planes -- into the SSIM / edge-difference
//              sums (the expressions of k_march): the nine per-pass planes never reach HBM after
//              the vertical pass.  Persistent: one workgroup per CU pulls (scale, channel, column
//              group) jobs longest first (see the kernel).
//   reference  mu1 = blur(x) and s11 = blur(x*x) depend on the reference alone (tq.zig:37 passes
//              the same e.rgb on every pass): ssimu2_set_reference runs both passes over {x, xx}
//              once and keeps the planes, so a pass of a search recurses 9 planes, not 15.
#pragma once

namespace ssimu2 {

constexpr int RG_N = 5;         // radius of the sigma-1.5 recursion: round(3.2795 sigma + 0.2546)
constexpr int RG_HL = 20;       // image rows per wave of the horizontal pass (4 DPP rows x 5 lines)
constexpr int RG_HT = 16;       // columns per register tile of the horizontal pass
constexpr int RG_VB = 2 * RG_N; // rows per batch of the vertical pass
constexpr int RG_VW = 64;       // columns per workgroup of the vertical pass
constexpr int RG_MAPS_WAVES = RG_VB / 2;  // each maps wave owns two rows of every batch

// Row pitch.  Every plane of the recursive modes -- XYB, cached reference, horizontal pass -- keeps its
// rows RG_PITCH_ALIGN floats (512 bytes) apart-aligned: pitch = w rounded up.  Measured (round 4, same
// pixels per frame): rows that do not start on 512 bytes cost k_rg_v 30-45 % and k_rg_h up to 35 %
// (3856, 3904, 3776 wide against 3840 / 4096) -- each lane quad's 64-byte segment then straddles lines,
// and the aligned 64-column stores of k_rg_h are aligned in no row but the first.  3 % more memory at
// worst on a 4K-class frame; nothing changes for widths that are multiples of 128.
constexpr int RG_PITCH_ALIGN = 128;
__host__ __device__ __forceinline__ int rg_pitch(int w) { return (w + RG_PITCH_ALIGN - 1) / RG_PITCH_ALIGN * RG_PITCH_ALIGN; }

struct RgPlan {
    int nscales;
    int w[kNumScales], h[kNumScales];
    int pitch[kNumScales];        // floats per row of every plane of that scale (rg_pitch(w)); a plane is pitch * h floats
    int hblk_end[kNumScales];     // exclusive end of each scale's workgroups in the k_rg_h grid
    int vblk_end[kNumScales];     // ... in the k_rg_v grid
    int vgroups[kNumScales];      // column groups of a scale = partial sums per statistic
    float* xout[kNumScales];      // where the conversion of this plan's frame writes its XYB planes [3][n]
    const float* xa[kNumScales];  // positive-XYB planes of the reference [3][n]
    const float* xb[kNumScales];  // ... of the distorted frame
    float* hbuf[kNumScales];      // horizontal pass: [channel][plane of the pass][n]
    float* cache[kNumScales];     // reference: [channel][{mu1, s11}][n]
    double* part[kNumScales];     // [18 statistics][vgroups]
    float* emit[kNumScales];      // k_rg_v_emit target [channel][plane of the pass][n]; null = skip the scale
    float* dump;                  // one row of floats that absorbs the stores of rows outside the image
    // persistent scheduling (k_rg_h, k_rg_v): jobs in decreasing length (largest scale first), handed
    // out through cursors in device memory that the conversion launch in front of them zeroes
    int hjob_end[kNumScales];     // exclusive end of each scale's jobs of the horizontal pass
    int hjobs, hlong;             // all of them / the first `hlong` form the queue of the long class
    int vjobs;                    // jobs of the vertical pass = vblk_end[nscales - 1]
    unsigned* q;                  // [0] long-queue cursor, [1] filler-queue cursor (k_rg_h), [2] k_rg_v's
};

// ---- the recursion --------------------------------------------------------------------------------
// one step (FastGaussian1D, scalar form):
//   o_k = n2_k * (left + right) - prev2_k - d1_k * prev_k;   out = (o_1 + o_3) + o_5
// FMA = false: every operation rounded by itself (the published scalar order);
// FMA = true:  the last multiply-subtract fused, fma(-d1_k, prev_k, .), as a compiler targeting
//              an FMA unit contracts it (SSIMU2_BLUR_RECURSIVE_FMA; the checker's OR_BLUR_IIR_FMA)

// all three sections in one lane (vertical pass)
struct RgState {
    float p1[3], p2[3];  // previous and second-previous output of each section
};
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

// one section per lane, the sections of a line five lanes apart inside a 16-lane DPP row:
// lanes 0-4 section 1, 5-9 section 3, 10-14 section 5 of lines 0-4; lane 15 idles.
__device__ __forceinline__ float rg_shr5(float v) {  // value of the lane five below (same DPP row)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x115, 0xf, 0xf, true));
}
// returns the line's output in the lanes of section 5 (10-14): (o1 + o3) + o5 in that order
template <bool FMA>
__device__ __forceinline__ float rg_step_lane(float& p1, float& p2, float left, float right, float n2, float d1) {
    const float sum = left + right;
    float v = sum * n2;
    v = v - p2;
    if (FMA) v = fmaf(-d1, p1, v);
    else v = v - d1 * p1;
    p2 = p1;
    p1 = v;
    const float t = rg_shr5(v) + v;  // lanes 5-9: o1 + o3
    return rg_shr5(t) + v;           // lanes 10-14: (o1 + o3) + o5
}

typedef float rg_f4 __attribute__((ext_vector_type(4)));
typedef float rg_f4u __attribute__((ext_vector_type(4), aligned(4)));  // a row need not start on 16 bytes
// Streaming (nontemporal) access for planes that are written once and read once per pass
// (measured on MI355X, 4K cached pass: loads of the vertical pass 233 -> 192 us).
#ifndef RG_NT_ST
#define RG_NT_ST 1   // horizontal pass: stores of its planes (0.437 -> 0.425 ms)
#endif
#ifndef RG_NT_LD
#define RG_NT_LD 1   // vertical pass: loads of those planes, of the cached reference planes and of the XYB planes
#endif
#ifndef RG_NT_HLD
#define RG_NT_HLD 0  // horizontal pass: loads of the XYB planes
#endif
// Diagnosis builds only (scripts/gpu_rg_exp.sh; results are garbage, timings are the point):
// what bounds a stage when its HBM stream is taken away.
#ifndef RG_EXP_H_NOSTORE
#define RG_EXP_H_NOSTORE 0   // horizontal pass without its global stores
#endif
#ifndef RG_EXP_H_SAMETILE
#define RG_EXP_H_SAMETILE 0  // horizontal pass loading tile 0 over and over (cache-fed)
#endif
#ifndef RG_EXP_H_CLASS
#define RG_EXP_H_CLASS 0     // 1: only the long class works (filler waves exit), 2: only the filler class
#endif
#ifndef RG_EXP_V_SAMEROWS
#define RG_EXP_V_SAMEROWS 0  // vertical pass loading rows 0-9 of every plane over and over (cache-fed)
#endif
#ifndef RG_EXP_V_REVERSE
#define RG_EXP_V_REVERSE 0   // vertical pass: column groups right to left (most recently written first)
#endif

// Staging of the horizontal pass.  A lane that streamed its own row (round 2, and the first
// round-3 form: 16-byte loads, 20 rows per wave) hands the memory pipeline one request PER LANE --
// 64 tag look-ups per load instruction for 20 distinct 64-byte segments -- and the texture
// addresser, not the recursion, set the time (k_rg_h 0.50 ms per 4K pass).  So a wave stages its 20
// rows 64 columns at a time through wave-private LDS: five 16-byte loads per lane cover 20 rows x
// 256 contiguous bytes (16 lanes per row: whole 64-byte segments per lane quad), one tile ahead of
// its use; five ds_write_b128 put the tile into [row][column] order; each lane then reads its own
// line 16 columns at a time (the three section lanes of a line read the same address: a
// broadcast).  Outputs go back the same way: 16 columns per line into an LDS ring of 128 columns,
// from which five coalesced 16-byte stores per lane write the 64-column segment that is ALIGNED
// with the input tiles (a tile's own outputs are columns 64 T - 4 .. 64 T + 59; stored as they
// fall they straddle the 64-byte lines: WRITE_SIZE showed 1.27x the plane bytes), one tile late.
// Wave-private LDS needs no barrier: the LDS executes one wave's
// instructions in order; __builtin_amdgcn_wave_barrier() only stops the COMPILER from moving an
// access across it (it cannot see that lanes read what other lanes wrote).
constexpr int RG_TW = 64;          // columns per staged tile
constexpr int RG_TP = RG_TW + 4;   // LDS row pitch in floats (16-byte multiple; spreads the rows over the banks)
constexpr int RG_TR = RG_HL / 4;   // 16-byte accesses per lane and tile (4 rows of 16 lanes each)
constexpr int RG_OW = 2 * RG_TW;   // columns of the output ring
typedef float RgTile[RG_HL][RG_TP];
typedef float RgRing[RG_HL][RG_OW + 4];

struct RgLine {
    const float* ga;        // plane(s) read (uniform)
    const float* gb;
    float* gout;            // plane written (uniform)
    uint32_t goff[RG_TR];   // per lane: float offset of (row 4 i + lane / 16, column 4 (lane % 16)) in a plane
    int w;
    int line;               // this lane's line (0..19) and whether it holds the line's output (section 5)
    bool holds_out;
    float n2, d1;           // this lane's section
    float p1, p2;           // ... and its state
    float cur[2][RG_HT];    // inputs of the current and the previous 16-column group (by parity)
};

template <int OPK>
__device__ __forceinline__ void rg_h_fetch(const RgLine& L, rg_f4 (&ra)[RG_TR], rg_f4 (&rb)[RG_TR], int T) {
    // unconditional and all alike (countable by s_waitcnt vmcnt): a tile that reaches past the end
    // of a row reads on into the next row / plane of the same allocation -- every plane read here
    // is followed by other planes of the context's buffer -- and the EDGE tiles zero what lies
    // outside the row
#pragma unroll
    for (int i = 0; i < RG_TR; ++i) {
        const size_t toff = RG_EXP_H_SAMETILE ? 0 : (size_t)T * RG_TW;
        const float* pa = L.ga + ((size_t)L.goff[i] + toff);
        const float* pb = L.gb + ((size_t)L.goff[i] + toff);
        if (RG_NT_HLD) {
            ra[i] = __builtin_nontemporal_load(reinterpret_cast<const rg_f4u*>(pa));
            if (OPK == 2) rb[i] = __builtin_nontemporal_load(reinterpret_cast<const rg_f4u*>(pb));
        } else {
            __builtin_memcpy(&ra[i], pa, 16);
            if (OPK == 2) __builtin_memcpy(&rb[i], pb, 16);
        }
    }
}

// Tile T of the line: 64 steps.  Step m has the right-hand input column m and the left-hand one
// m - 10 and produces output column m - 4 (zeros outside the row: the published padding).
// OPK: 0 = the plane itself, 1 = its square, 2 = the product of two planes (rounded to fp32 before
// the blur, as published).  EDGE: the tile touches the row's first or last columns.
// Segment S = output columns 64 S .. 64 S + 63 of the 20 rows, from the ring to the plane.
template <bool EDGE>
__device__ __forceinline__ void rg_h_store(const RgLine& L, const RgRing& tout, int S) {
    const int rr = threadIdx.x >> 4 & 3, cc = threadIdx.x & 15;
    if (RG_EXP_H_NOSTORE) return;
#pragma unroll
    for (int i = 0; i < RG_TR; ++i) {
        const rg_f4 v = *reinterpret_cast<const rg_f4*>(&tout[4 * i + rr][((S & 1) * RG_TW) + 4 * cc]);
        float* dst = L.gout + ((size_t)L.goff[i] + (size_t)S * RG_TW);
        if (!EDGE) {
            // written once here, read once by the vertical pass: streaming stores
            if (RG_NT_ST) __builtin_nontemporal_store((rg_f4u)v, reinterpret_cast<rg_f4u*>(dst));
            else __builtin_memcpy(dst, &v, 16);
        } else {
            const int col = S * RG_TW + 4 * cc;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (col + e < L.w) dst[e] = v[e];
        }
    }
}

// AHEAD: tiles the global loads run ahead of their use (1, or 2 with two register sets).  Measured (round 4,
// profiles/r04_rg_ahead_ab.log): two tiles ahead take the REFERENCE's horizontal pass (planes x, x*x; two waves per
// workgroup; 238 VGPRs, no spill) from 134 to 124 us.  The pass kernel gets SLOWER with it: its x*y plane needs 80
// registers for two sets and spills under the 256-register cap of two waves per SIMD (171 us against 163), and with
// only the one-plane jobs two ahead it is still 171 -- three waves per workgroup already keep the memory system
// at 0.85 of what a mixed stream reaches, more loads in flight only lengthen the queue.  So: 2 for the reference, 1 for a pass.
#ifndef RG_H_AHEAD_REF
#define RG_H_AHEAD_REF 2
#endif
#ifndef RG_H_AHEAD_PASS
#define RG_H_AHEAD_PASS 1
#endif
template <bool FMA, int OPK, bool EDGE, int AHEAD>
__device__ __forceinline__ void rg_h_tile(RgLine& L, RgTile& tin, RgRing& tout, float* dump,
                                          rg_f4 (&ra)[RG_TR], rg_f4 (&rb)[RG_TR], int T) {
    const int rr = threadIdx.x >> 4 & 3, cc = threadIdx.x & 15;
    __builtin_amdgcn_wave_barrier();
    // the plane the recursion runs over is formed HERE, once per element, by the lane that staged
    // it (a product is one fp32 multiply, rounded before the blur as published -- the same value
    // whichever lane forms it); the three section lanes of a line then read it ready-made
#pragma unroll
    for (int i = 0; i < RG_TR; ++i) {
        rg_f4 v = OPK == 0 ? ra[i] : OPK == 1 ? ra[i] * ra[i] : ra[i] * rb[i];
        if (EDGE) {  // zeros outside the row: the published padding
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = T * RG_TW + 4 * cc + e < L.w ? v[e] : 0.0f;
        }
        *reinterpret_cast<rg_f4*>(&tin[4 * i + rr][4 * cc]) = v;
    }
    rg_h_fetch<OPK>(L, ra, rb, T + AHEAD);  // lands under the steps of this tile (and of the next, AHEAD = 2)
    __builtin_amdgcn_wave_barrier();
    // 16 columns at a time; the LDS reads of the next 16 are issued before the steps of these 16
    // (an LDS read takes ~100+ cycles to land)
    rg_f4 va[2][RG_HT / 4];
#define RG_H_READ(S)                                                                           \
    _Pragma("unroll") for (int v = 0; v < RG_HT / 4; ++v)                                      \
        va[(S) & 1][v] = *reinterpret_cast<const rg_f4*>(&tin[L.line][RG_HT * (S) + 4 * v]);
    RG_H_READ(0)
#pragma unroll
    for (int s = 0; s < RG_TW / RG_HT; ++s) {
        float (&c)[RG_HT] = L.cur[s & 1];
        const float (&pv)[RG_HT] = L.cur[(s & 1) ^ 1];
        if (s + 1 < RG_TW / RG_HT) {
            RG_H_READ(s + 1)
            __builtin_amdgcn_sched_barrier(0);  // keep them up here
        }
#pragma unroll
        for (int v = 0; v < RG_HT / 4; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) c[4 * v + e] = va[s & 1][v][e];
        float o[RG_HT];
#pragma unroll
        for (int mm = 0; mm < RG_HT; ++mm) {
            const float left = mm >= 2 * RG_N ? c[mm - 2 * RG_N] : pv[mm + RG_HT - 2 * RG_N];
            o[mm] = rg_step_lane<FMA>(L.p1, L.p2, left, c[mm], L.n2, L.d1);
        }
        // output columns 64 T - 4 + 16 s ..: ring position = column mod 128.  Every lane stores (no
        // branch around LDS traffic in the middle of the tile): the lanes that do not hold the
        // line's output write their 16 bytes into a dump row nobody reads.
#pragma unroll
        for (int v = 0; v < RG_HT / 4; ++v) {
            const int col = (T * RG_TW - (RG_N - 1) + RG_HT * s + 4 * v) & (RG_OW - 1);
            float* dst = L.holds_out ? &tout[L.line][col] : dump + 4 * (threadIdx.x & 31);
            *reinterpret_cast<rg_f4*>(dst) = rg_f4{o[4 * v], o[4 * v + 1], o[4 * v + 2], o[4 * v + 3]};
        }
    }
#undef RG_H_READ
    __builtin_amdgcn_wave_barrier();
    if (T > 0) rg_h_store<EDGE>(L, tout, T - 1);  // uniform; segment T - 1 is complete now
}

template <bool FMA, int OPK, int AHEAD>
__device__ __forceinline__ void rg_h_line(RgLine& L, RgTile& tin, RgRing& tout, float* dump) {
    const int w = L.w;
    const int ntiles = (w + (RG_N - 1) + RG_TW - 1) / RG_TW;  // steps run to m = w + 3
    // tiles 1 .. nmain - 1 lie inside the row and complete a segment that does (64 (T + 1) <= w)
    const int nmain = max(1, w / RG_TW);
#pragma unroll
    for (int k = 0; k < RG_HT; ++k) L.cur[1][k] = 0.0f;  // columns -16 .. -1
    L.p1 = L.p2 = 0.0f;
    if constexpr (AHEAD == 1) {
        rg_f4 ra[RG_TR], rb[RG_TR];
        rg_h_fetch<OPK>(L, ra, rb, 0);
        rg_h_tile<FMA, OPK, true, 1>(L, tin, tout, dump, ra, rb, 0);
        int T = 1;
#pragma unroll 1
        for (; T < nmain; ++T) rg_h_tile<FMA, OPK, false, 1>(L, tin, tout, dump, ra, rb, T);
#pragma unroll 1
        for (; T < ntiles; ++T) rg_h_tile<FMA, OPK, true, 1>(L, tin, tout, dump, ra, rb, T);
    } else {
        // two register sets: tile T is staged from set T & 1, which is then refilled with tile T + 2
        rg_f4 ra0[RG_TR], rb0[RG_TR], ra1[RG_TR], rb1[RG_TR];
        rg_h_fetch<OPK>(L, ra0, rb0, 0);
        rg_h_fetch<OPK>(L, ra1, rb1, 1);
        rg_h_tile<FMA, OPK, true, 2>(L, tin, tout, dump, ra0, rb0, 0);
        int T = 1;
#pragma unroll 1
        for (; T + 1 < nmain; T += 2) {  // T odd here
            rg_h_tile<FMA, OPK, false, 2>(L, tin, tout, dump, ra1, rb1, T);
            rg_h_tile<FMA, OPK, false, 2>(L, tin, tout, dump, ra0, rb0, T + 1);
        }
        if (T < nmain) {  // uniform
            rg_h_tile<FMA, OPK, false, 2>(L, tin, tout, dump, ra1, rb1, T);
            ++T;
        }
#pragma unroll 1
        for (; T < ntiles; ++T) {
            if (T & 1) rg_h_tile<FMA, OPK, true, 2>(L, tin, tout, dump, ra1, rb1, T);
            else rg_h_tile<FMA, OPK, true, 2>(L, tin, tout, dump, ra0, rb0, T);
        }
    }
    __builtin_amdgcn_wave_barrier();
    rg_h_store<true>(L, tout, ntiles - 1);  // the row's last columns (w - 1 <= 64 (ntiles - 1) + 59)
}

// plane index among the 15 of a scale (5 * channel + {x, y, xx, yy, xy}) of plane `kind` of a pass
__host__ __device__ __forceinline__ int rg_plane15(bool ref, int ch, int kind) {
    return 5 * ch + (ref ? 2 * kind : (kind == 0 ? 1 : kind + 2));
}

// Horizontal pass.  REF: the planes {x, x*x} of the reference (once per search); otherwise
// {y, y*y, x*y}.  grid = sum over scales of 3 channels x ceil(h / 20) workgroups of NK waves, one wave
// per plane, each staging and recursing its own tiles.
//
// What bounds it (round 4, profiles/r04_rg_chain_vs_bytes.log, 4K pass): with every HBM access taken
// away the launch takes 139 us (a wave alone on a SIMD needs 26 ns per step, 101 us for a 3,840-step
// row; some SIMDs hold two full-resolution waves), with only its loads or only its stores coming from
// HBM 144-152 us, with both 159-177 us: 0.70 GB of interleaved reads and writes at 4.0-4.4 TB/s, 0.85 of
// the 5.1 TB/s a clean 2 : 3 read + write stream reaches on this chip (profiles/r04_rw_mix.txt) -- and
// at that ceiling the bytes alone would take 137 us, the same as the chains alone: both floors coincide.
// A persistent form that gives every full-resolution chain a SIMD of its own (RG_H_PERSISTENT = 1:
// one workgroup of eight waves per CU, jobs handed out longest first through two cursors, the class
// of a wave decided by the SIMD it finds itself on) has the same 137 us without HBM and 206-222 us
// with it -- more row streams in flight, the long chains exposed to the memory latency once their
// partner wave has finished -- so it stays an A/B build; the bytes of the h -> v round trip are the
// bound, not the placement.
#ifndef RG_H_PERSISTENT
#define RG_H_PERSISTENT 0
#endif
constexpr int RG_HW = 8;  // waves per workgroup of the persistent form
template <bool REF>
__device__ __forceinline__ int rg_h_pull(const RgPlan& p, bool long_first) {
    int job = -1;
    if ((threadIdx.x & 63) == 0) {
        const int nq[2] = {p.hlong, p.hjobs - p.hlong};
        const int a = long_first ? 0 : 1;
        int idx = (int)atomicAdd(p.q + a, 1u);
        if (idx < nq[a]) job = a == 0 ? idx : p.hlong + idx;
        else if (!RG_EXP_H_CLASS) {
            idx = (int)atomicAdd(p.q + (1 - a), 1u);
            if (idx < nq[1 - a]) job = a == 0 ? p.hlong + idx : idx;
        }
    }
    return __builtin_amdgcn_readfirstlane(job);
}

template <bool FMA, bool REF>
__global__ __launch_bounds__(64 * RG_HW) void k_rg_h_persistent(RgPlan p) {
    constexpr int NK = REF ? 2 : 3;
    __shared__ __attribute__((aligned(16))) RgTile s_in[RG_HW];
    __shared__ __attribute__((aligned(16))) RgRing s_o[RG_HW];
    __shared__ __attribute__((aligned(16))) float s_dump[RG_OW];
    __shared__ int s_on_simd[4];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // class of this wave: the FIRST wave of the workgroup on each SIMD is of the long class, whatever
    // order the dispatcher placed the eight waves in (HW_ID bits 5:4 = the SIMD the wave runs on)
    if (threadIdx.x < 4) s_on_simd[threadIdx.x] = 0;
    __syncthreads();
    unsigned hw_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    int arrival = 0;
    if (lane == 0) arrival = atomicAdd(&s_on_simd[(hw_id >> 4) & 3u], 1);
    const bool long_class = __builtin_amdgcn_readfirstlane(arrival) == 0;
    (void)wave;
    if (RG_EXP_H_CLASS == 1 && !long_class) return;
    if (RG_EXP_H_CLASS == 2 && long_class) return;
    RgLine L;
    // lane -> (DPP row, line, section); lane 15 of a row shadows line 4 / section 5 and holds nothing
    const int l16 = lane & 15;
    const int sec = l16 < 15 ? l16 / 5 : 2;
    L.line = (lane >> 4) * 5 + (l16 < 15 ? l16 - 5 * sec : 4);
    L.holds_out = l16 >= 10 && l16 < 15;
    L.n2 = c_k.rg_n2[sec];
    L.d1 = c_k.rg_d1[sec];
#pragma unroll 1
    for (;;) {
        const int job = rg_h_pull<REF>(p, long_class);
        if (job < 0) break;
        int sc = 0, first = 0;
#pragma unroll
        for (int s = 0; s < kNumScales - 1; ++s)
            if (s + 1 < p.nscales && job >= p.hjob_end[s]) {
                sc = s + 1;
                first = p.hjob_end[s];
            }
        // jobs of a scale: row group, then channel, then plane -- neighbours share their inputs
        const int local = job - first;
        const int rgrp = local / (3 * NK), ch = local % (3 * NK) / NK, kind = local % NK;
        const int w = p.w[sc], h = p.h[sc], pitch = p.pitch[sc];
        const size_t n = (size_t)pitch * h;
        L.w = w;
        // rows of the cooperative 16-byte accesses; rows behind the image shadow its last row (their
        // lines compute and store that row's values once more)
#pragma unroll
        for (int i = 0; i < RG_TR; ++i)
            L.goff[i] = (uint32_t)min(rgrp * RG_HL + 4 * i + (lane >> 4), h - 1) * (uint32_t)pitch + 4u * (uint32_t)(lane & 15);
        const float* xa = p.xa[sc] + ch * n;
        const float* xb = REF ? xa : p.xb[sc] + ch * n;
        L.gout = p.hbuf[sc] + (size_t)(ch * NK + kind) * n;
        // REF: x, x*x.  pass: y, y*y, x*y
        L.ga = kind == 2 ? xa : xb;
        L.gb = xb;
        constexpr int AHEAD = REF ? RG_H_AHEAD_REF : RG_H_AHEAD_PASS;
        if (kind == 0) rg_h_line<FMA, 0, AHEAD>(L, s_in[wave], s_o[wave], s_dump);
        else if (kind == 1) rg_h_line<FMA, 1, AHEAD>(L, s_in[wave], s_o[wave], s_dump);
        else rg_h_line<FMA, 2, AHEAD>(L, s_in[wave], s_o[wave], s_dump);
    }
}

template <bool FMA, bool REF>
__global__ __launch_bounds__(REF ? 128 : 192) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_rg_h(RgPlan p) {  // (capped at 168 VGPRs for 3 waves per SIMD it spills and is slower)
    constexpr int NK = REF ? 2 : 3;
    __shared__ __attribute__((aligned(16))) RgTile s_in[NK];
    __shared__ __attribute__((aligned(16))) RgRing s_o[NK];
    __shared__ __attribute__((aligned(16))) float s_dump[RG_OW];
    int sc = 0, first = 0;
#pragma unroll
    for (int s = 0; s < kNumScales - 1; ++s)
        if (s + 1 < p.nscales && (int)blockIdx.x >= p.hblk_end[s]) {
            sc = s + 1;
            first = p.hblk_end[s];
        }
    const int blk = (int)blockIdx.x - first;
    const int ch = blk % 3, rgrp = blk / 3;
    const int kind = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int w = p.w[sc], h = p.h[sc], pitch = p.pitch[sc];
    const size_t n = (size_t)pitch * h;
    RgLine L;
    // lane -> (DPP row, line, section); lane 15 of a row shadows line 4 / section 5 and holds nothing
    const int l16 = lane & 15;
    const int sec = l16 < 15 ? l16 / 5 : 2;
    L.line = (lane >> 4) * 5 + (l16 < 15 ? l16 - 5 * sec : 4);
    L.holds_out = l16 >= 10 && l16 < 15;
    L.n2 = c_k.rg_n2[sec];
    L.d1 = c_k.rg_d1[sec];
    L.w = w;
    // rows of the cooperative 16-byte accesses; rows behind the image shadow its last row (their
    // lines compute and store that row's values once more)
#pragma unroll
    for (int i = 0; i < RG_TR; ++i)
        L.goff[i] = (uint32_t)min(rgrp * RG_HL + 4 * i + (lane >> 4), h - 1) * (uint32_t)pitch + 4u * (uint32_t)(lane & 15);
    const float* xa = p.xa[sc] + ch * n;
    const float* xb = REF ? xa : p.xb[sc] + ch * n;
    L.gout = p.hbuf[sc] + (size_t)(ch * NK + kind) * n;
    // REF: x, x*x.  pass: y, y*y, x*y
    L.ga = kind == 2 ? xa : xb;
    L.gb = xb;
    constexpr int AHEAD = REF ? RG_H_AHEAD_REF : RG_H_AHEAD_PASS;
    if (kind == 0) rg_h_line<FMA, 0, AHEAD>(L, s_in[0], s_o[0], s_dump);
    else if (kind == 1) rg_h_line<FMA, 1, AHEAD>(L, s_in[1], s_o[1], s_dump);
    else rg_h_line<FMA, 2, AHEAD>(L, s_in[NK - 1], s_o[NK - 1], s_dump);
}

// ---- vertical pass (+ maps) -----------------------------------------------------------------------
// The SSIM and edge-difference terms of one pixel from its five blurred values and the two frames
// (the expressions of march_v; products rounded as there).
__device__ __forceinline__ void rg_maps_pixel(float mu1, float mu2, float s11, float s22, float s12, float r1,
                                              float r2, double (&acc)[6]) {
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

// The recursion of one plane down this lane's column, PF batches of ten rows in a register queue
// (three in flight under the one consumed).  Everything that counts in vmcnt is unconditional:
// rows behind the image load the last row (and are replaced by the published zero padding in the
// steps), lanes right of the image shadow its last column, and the batch count is rounded up to
// the queue depth (the batches behind the image produce nothing that is kept).  `emit(b, o)`
// receives the ten outputs of batch b: rows 10 b - 4 .. 10 b + 5.
#ifndef RG_PF_DEPTH
#define RG_PF_DEPTH 6   // 4 -> 6 with one workgroup per CU (round 4): 0.388 -> 0.375 ms per cached 4K pass; 8: 0.378
#endif
constexpr int RG_PF = RG_PF_DEPTH;  // queue slots: RG_PF - 1 batches in flight under the one consumed
__device__ __forceinline__ int rg_v_batches(int h) {
    return ((h + (RG_N - 1) + RG_VB - 1) / RG_VB + RG_PF - 1) / RG_PF * RG_PF;  // step m = right-hand row, to h + 3
}

template <bool FMA, typename Emit>
__device__ __forceinline__ void rg_v_column(const float* __restrict__ in, int pitch, int h, Emit emit) {
    const float n2[3] = {c_k.rg_n2[0], c_k.rg_n2[1], c_k.rg_n2[2]};
    const float d1[3] = {c_k.rg_d1[0], c_k.rg_d1[1], c_k.rg_d1[2]};
    const int nb = rg_v_batches(h);
    RgState st;
#pragma unroll
    for (int k = 0; k < 3; ++k) st.p1[k] = st.p2[k] = 0.f;
    float q[RG_PF][RG_VB];  // queue slot b % PF holds rows 10 b .. 10 b + 9, as loaded
#define RG_V_LOAD(B, SLOT)                                          \
    _Pragma("unroll") for (int j = 0; j < RG_VB; ++j)               \
        q[SLOT][j] = RG_NT_LD ? __builtin_nontemporal_load(in + (size_t)min((RG_EXP_V_SAMEROWS ? 0 : (B) * RG_VB) + j, h - 1) * pitch) \
                           : in[(size_t)min((RG_EXP_V_SAMEROWS ? 0 : (B) * RG_VB) + j, h - 1) * pitch];
#pragma unroll
    for (int j = 0; j < RG_VB; ++j) q[RG_PF - 1][j] = 0.f;  // batch -1: rows -10 .. -1
#pragma unroll
    for (int k = 0; k < RG_PF - 1; ++k) { RG_V_LOAD(k, k) }
#pragma unroll 1
    for (int b0 = 0; b0 < nb; b0 += RG_PF) {
#pragma unroll
        for (int u = 0; u < RG_PF; ++u) {
            const int b = b0 + u;
            // the slot of batch b + RG_PF - 1 is the one batch b - 1 (this batch's left-hand inputs)
            // sits in, so its loads are issued AFTER the steps
            float (&right)[RG_VB] = q[u];
            const float (&left)[RG_VB] = q[(u + RG_PF - 1) % RG_PF];
            if (b * RG_VB + RG_VB > h) {  // uniform: the image ends inside or before this batch
#pragma unroll
                for (int j = 0; j < RG_VB; ++j) right[j] = b * RG_VB + j < h ? right[j] : 0.0f;
            }
            float o[RG_VB];
#pragma unroll
            for (int j = 0; j < RG_VB; ++j) o[j] = rg_step<FMA>(st, left[j], right[j], n2, d1);
            RG_V_LOAD(b + RG_PF - 1, (u + RG_PF - 1) % RG_PF)
            emit(b, o);
        }
    }
#undef RG_V_LOAD
}

// Vertical pass that writes its planes: the reference's mu1 / s11 (NK = 2: {x, x*x}, once per
// search) -- and, in instrumented builds, the three per-pass planes for the parity tests (NK = 3).
// One wave per plane and 64 columns; grid = sum over scales of 3 channels x ceil(w / 64).
// Stores are unconditional too: rows outside the image go to a dump row.
template <bool FMA, int NK>
__global__ __launch_bounds__(64 * NK) void k_rg_v_emit(RgPlan p) {
    int sc = 0, first = 0;
#pragma unroll
    for (int s = 0; s < kNumScales - 1; ++s)
        if (s + 1 < p.nscales && (int)blockIdx.x >= p.vblk_end[s]) {
            sc = s + 1;
            first = p.vblk_end[s];
        }
    if (p.emit[sc] == nullptr) return;  // uniform (the instrumented builds' one-scale runs)
    const int blk = (int)blockIdx.x - first;
    const int ch = blk % 3, cg = blk / 3;
    const int kind = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int w = p.w[sc], h = p.h[sc], pitch = p.pitch[sc];
    const size_t n = (size_t)pitch * h;
    const int xc = min(cg * RG_VW + (int)(threadIdx.x & 63), w - 1);  // lanes right of the image shadow its last column
    const float* in = p.hbuf[sc] + (size_t)(ch * NK + kind) * n + xc;
    float* out = p.emit[sc] + (size_t)(ch * NK + kind) * n;
    float* dump = p.dump;
    rg_v_column<FMA>(in, pitch, h, [&](int b, const float (&o)[RG_VB]) {
#pragma unroll
        for (int j = 0; j < RG_VB; ++j) {
            const int r = b * RG_VB + j - (RG_N - 1);  // output row (uniform)
            float* row = r >= 0 && r < h ? out + (size_t)r * pitch : dump;
            row[xc] = o[j];
        }
    });
}

// Vertical pass of a pass's planes + maps.  Waves 0-2 recurse {y, y*y, x*y} into a double-buffered
// LDS tile of ten rows, waves 3-7 turn two of those rows each, with the cached mu1 / s11 and the two
// frames' XYB values, into the six sums.  A JOB is one channel of 64 columns of one scale (a chain of
// h steps); round 4: PERSISTENT like k_rg_h -- one workgroup per CU (launched with rg_v_pad_bytes()
// of unused dynamic LDS, past half a CU's), the jobs of all scales handed out longest first through a cursor.  Measured before the change
// (profiles/r04_rg_chain_vs_bytes.log): capping the round-3 launch at one workgroup per CU took it from
// 193 to 170 us -- with two per CU the dispatcher doubles up full-resolution column groups on some CUs
// while others run the small scales, and a CU with two of them issues at half the rate per chain.
// Dynamic LDS of the launch, never touched: with the kernel's own tiles it pushes the workgroup past half of the CU's
// LDS, so the hardware cannot place a second workgroup on the CU.  Derived from the LDS size ssimu2_ctx_create read
// off the device (160 KB on gfx950: 81 KB - 15 KB of tiles = 66 KB); a device with less than 160 KB never gets here
// (SSIMU2_ERR_NO_DEVICE at context creation).
constexpr unsigned RG_V_STATIC_LDS = 2u * 3u * RG_VB * RG_VW * (unsigned)sizeof(float);  // s_out; the rest is < 1 KB
constexpr unsigned rg_v_pad_bytes(unsigned lds_bytes_per_cu) {
    return lds_bytes_per_cu / 2u + 1024u > RG_V_STATIC_LDS ? lds_bytes_per_cu / 2u + 1024u - RG_V_STATIC_LDS : 0u;
}
static_assert(rg_v_pad_bytes(160u * 1024u) + RG_V_STATIC_LDS > 80u * 1024u && rg_v_pad_bytes(160u * 1024u) + RG_V_STATIC_LDS + 1024u <= 160u * 1024u,
              "k_rg_v: one workgroup per 160 KB CU, and the request fits");

template <bool FMA>
__global__ __launch_bounds__(512) void k_rg_v(RgPlan p) {
    constexpr int NK = 3;
    __shared__ float s_out[2][NK][RG_VB][RG_VW];
    __shared__ double s_part[RG_MAPS_WAVES][6];
    __shared__ int s_job;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (;;) {
        if (threadIdx.x == 0) {
            const int j = (int)atomicAdd(p.q + 2, 1u);
            s_job = j < p.vjobs ? j : -1;
        }
        __syncthreads();
        const int job = __builtin_amdgcn_readfirstlane(s_job);
        if (job < 0) break;
        int sc = 0, first = 0;
#pragma unroll
        for (int s = 0; s < kNumScales - 1; ++s)
            if (s + 1 < p.nscales && job >= p.vblk_end[s]) {
                sc = s + 1;
                first = p.vblk_end[s];
            }
        const int blk = job - first;
        const int ch = blk % 3, cg = RG_EXP_V_REVERSE ? p.vgroups[sc] - 1 - blk / 3 : blk / 3;
        const int w = p.w[sc], h = p.h[sc], pitch = p.pitch[sc];
        const size_t n = (size_t)pitch * h;
        const int x = cg * RG_VW + lane;
        const bool ok = x < w;
        const int xc = min(x, w - 1);
        const int nb = rg_v_batches(h);

        if (wave < NK) {
            const int kind = wave;
            const float* in = p.hbuf[sc] + (size_t)(ch * NK + kind) * n + xc;
            rg_v_column<FMA>(in, pitch, h, [&](int b, const float (&o)[RG_VB]) {
#pragma unroll
                for (int j = 0; j < RG_VB; ++j) s_out[b & 1][kind][j][lane] = o[j];
                __syncthreads();  // batch b is in the tile
            });
        } else {
            // maps: this wave's two rows of every batch; the same barriers as the recursion waves
            const int j0 = 2 * (wave - NK);
            const float* g_mu1 = p.cache[sc] + (size_t)(2 * ch) * n + xc;
            const float* g_s11 = g_mu1 + n;
            const float* g_r1 = p.xa[sc] + (size_t)ch * n + xc;
            const float* g_r2 = p.xb[sc] + (size_t)ch * n + xc;
            double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            float g[RG_PF][2][4];
#define RG_M_LOAD(B, SLOT)                                                     \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                         \
        const int r_ = (RG_EXP_V_SAMEROWS ? 0 : (B) * RG_VB) + j0 + jj - (RG_N - 1); \
        const size_t o_ = (size_t)min(max(r_, 0), h - 1) * pitch;              \
        g[SLOT][jj][0] = RG_NT_LD ? __builtin_nontemporal_load(g_mu1 + o_) : g_mu1[o_]; \
        g[SLOT][jj][1] = RG_NT_LD ? __builtin_nontemporal_load(g_s11 + o_) : g_s11[o_]; \
        g[SLOT][jj][2] = RG_NT_LD ? __builtin_nontemporal_load(g_r1 + o_) : g_r1[o_];   \
        g[SLOT][jj][3] = RG_NT_LD ? __builtin_nontemporal_load(g_r2 + o_) : g_r2[o_];   \
    }
#pragma unroll
            for (int k = 0; k < RG_PF - 1; ++k) { RG_M_LOAD(k, k) }
#pragma unroll 1
            for (int b0 = 0; b0 < nb; b0 += RG_PF) {
#pragma unroll
                for (int u = 0; u < RG_PF; ++u) {
                    const int b = b0 + u;
                    RG_M_LOAD(b + RG_PF - 1, (u + RG_PF - 1) % RG_PF)
                    __syncthreads();  // batch b is in the tile
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int r = b * RG_VB + j0 + jj - (RG_N - 1);
                        if (r >= 0 && r < h) {  // uniform
                            const float mu2 = s_out[b & 1][0][j0 + jj][lane];
                            const float s22 = s_out[b & 1][1][j0 + jj][lane];
                            const float s12 = s_out[b & 1][2][j0 + jj][lane];
                            double z[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                            rg_maps_pixel(g[u][jj][0], mu2, g[u][jj][1], s22, s12, g[u][jj][2], g[u][jj][3], z);
                            if (ok) {
#pragma unroll
                                for (int k = 0; k < 6; ++k) acc[k] += z[k];
                            }
                        }
                    }
                }
            }
#undef RG_M_LOAD
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const double sum = wave_sum(acc[k]);
                if (lane == 0) s_part[wave - NK][k] = sum;
            }
        }
        __syncthreads();
        if (threadIdx.x < 6) {
            const int k = threadIdx.x;
            double sum = s_part[0][k];
#pragma unroll
            for (int m = 1; m < RG_MAPS_WAVES; ++m) sum += s_part[m][k];
            // statistic index as k_finalize reads it: 0..5 ssim (c*2 + n), 6..17 edge (c*4 + j)
            const int stat = k < 2 ? ch * 2 + k : 6 + ch * 4 + (k - 2);
            p.part[sc][(size_t)stat * p.vgroups[sc] + cg] = sum;
        }
        // the next job's first barrier (behind its pull) orders these reads of s_part / s_job before
        // anything overwrites them
    }
}

}  // namespace ssimu2
