// SSIMULACRA2 scorer for MI355X (gfx950): host side + the C ABI of include/ssimu2_hip.h.
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
//   partials    : fp64 [scale][18 stats][workgroups] partial sums
//   result      : fp64 [108 averages][score][nscales], written by k_finalize straight into pinned host memory
//
//   SSIMU2_BLUR_RECURSIVE modes only (ssimu2_recursive.h), every scale packed: XYB planes of both
//   frames, the reference's cached blur(x) / blur(x*x) planes, the horizontal pass of a pass's planes
//
// One score = ONE k_pyramid_bands launch (all five levels), ONE k_march launch covering all six scales, one
// k_finalize launch that writes its 880 bytes into the host mirror itself (recursive modes: conversion, horizontal
// pass, vertical pass + maps, k_finalize); everything on the ctx stream, no host sync inside (enqueue / wait
// split).  Streams the library creates are placed on distinct hardware queues ("stream placement").
#include <hip/hip_runtime.h>
#ifdef SSIMU2_INSTRUMENTED_BUILD
#include <hip/hip_ext.h>
#endif

#include <ctype.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <future>
#include <mutex>
#include <new>
#include <string>

#include "../../include/ssimu2_hip.h"
#include "ssimu2_kernels.h"
#include "ssimu2_recursive.h"

using namespace ssimu2;

namespace {

constexpr uint32_t kMinLdsPerCu = 160u * 1024u;  // gfx950's; what ssimu2_ctx_create requires (include/ssimu2_hip.h)

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
void gaussian_taps(double sigma, float taps[5], float n2_out[3], float d1_out[3]) {
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
    // the recursion itself (SSIMU2_BLUR_RECURSIVE): input gain and feedback of each section
    for (int k = 0; k < 3; ++k) {
        n2_out[k] = (float)(-beta[k] * cos(om[k] * (radius + 1.0)));
        d1_out[k] = (float)(-2.0 * cos(om[k]));
    }
}

// host twin of the device cbrt_repro (same IEEE sequence; this file is built with
// -ffp-contract=off, fmaf is the correctly rounded libm/hardware fma)
float cbrt_repro_host(float x) {
    if (!(x > 0.0f)) return 0.0f;
    uint32_t i;
    memcpy(&i, &x, 4);
    i = 0x54A21D2Au - i / 3u;
    float y;
    memcpy(&y, &i, 4);
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
    return c;
}

thread_local std::string g_create_error;

}  // namespace

struct ssimu2_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;   // created here, destroyed with the ctx
    bool pool_stream = false;  // borrowed from the process-wide set of streams on distinct hardware queues
    std::string err;

    // capacity (bytes / floats / doubles currently allocated)
    size_t cap_u8 = 0, cap_lin = 0, cap_part = 0;
    uint8_t* d_ref_u8 = nullptr;
    uint8_t* d_dist_u8 = nullptr;
    float* d_lin_ref = nullptr;   // scales 1..5 packed
    float* d_lin_dist = nullptr;
    float* d_xyb_ref = nullptr;   // cached positive-XYB planes of the reference, all scales
    float* d_ref_blur = nullptr;  // cached blur(ref*ref) planes, all scales
    size_t cap_blur = 0;          // floats allocated (0 = not cached)
    uint8_t* d_stage = nullptr;   // decoded avifRGBImage as uploaded (RGBA / padded rows)
    size_t cap_stage = 0;
    size_t cap_xyb = 0;
    double* d_partials = nullptr;
    // SSIMU2_BLUR_RECURSIVE only (ssimu2_recursive.h), every scale packed: XYB planes of both frames
    // [3], the reference's cached mu1 / s11 planes [6], the horizontal pass of a pass's planes [9]
    int blur_mode = SSIMU2_BLUR_FIR;
    float* d_rg = nullptr;
    size_t cap_rg = 0;            // floats
    double* d_rg_part = nullptr;  // [scale][18][column groups]
    unsigned* d_rg_q = nullptr;   // job cursors of the persistent recursive-mode kernels (4 words)
    int num_cus = 0;              // workgroups of those kernels: one per CU
    ssimu2_device_info dev{};     // what ctx_create saw of the device (and checked: gfx950, 160 KB of LDS per CU)
    unsigned rg_v_pad = 0;        // unused dynamic LDS of k_rg_v's launch: rg_v_pad_bytes(dev.lds_bytes_per_cu)
    size_t cap_rg_part = 0;       // doubles
    float* d_rg_dbg = nullptr;    // instrumented builds: 15 + 15 raw planes of scale rg_dbg_scale
    size_t cap_rg_dbg = 0;
    double* d_result = nullptr;   // 110 doubles in device memory: the stage timing of the instrumented build only
    double* h_result = nullptr;   // page-locked host memory k_finalize writes the result into (110 doubles)

    // reference state
    bool have_ref = false;
    uint32_t ref_w = 0, ref_h = 0;
    bool pending = false;

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // measurement builds only (ssimu2_instrument.hip); always 0 / true in the product library
    int seg_rows_override = 0;
    int seg_rows_tail_override = 0;
    bool cache_ref_blur = true;
    int rg_dbg_scale = -1;  // recursive mode: keep that scale's 15 raw planes (after each pass) downloadable
#ifdef SSIMU2_INSTRUMENTED_BUILD
    // the hipGraph experiment of the instrumented build (ssimu2_instr_use_graph): one instantiated chain of kernel nodes
    bool use_graph = false;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int graph_n = 0;
    hipGraphNode_t graph_node[8];
    void* graph_func[8];
    dim3 graph_grid[8], graph_block[8];
    unsigned graph_lds[8];
    unsigned long long graph_builds = 0, graph_launches = 0;
#endif

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

// Every kernel of a score goes through launch().  Product build: hipLaunchKernelGGL, nothing else.  Instrumented build
// (ssimu2_instrument.hip): while a timing scope is open on the calling thread the same launch is made with
// hipExtLaunchKernelGGL and a start / stop event pair, so the duration of each kernel comes from its own dispatch packet
// (what rocprofv3's kernel trace reads) with no marker or barrier packet added between the launches of a score.
template <class T>
struct arg_of { using type = T; };
#ifdef SSIMU2_INSTRUMENTED_BUILD
struct LaunchTimer {
    hipEvent_t* ev;  // cap events: launch k of the scope uses ev[2k] (start) and ev[2k + 1] (stop)
    int n, cap;
};
thread_local LaunchTimer* g_launch_timer = nullptr;

// The hipGraph experiment (ssimu2_instr_use_graph): while a recorder is open the launches of one score
// are not made but noted -- function, geometry, a copy of the arguments -- and enqueue_score() hands the chain to
// graph_flush(), which keeps one instantiated graph of kernel nodes per context, rewrites the nodes' parameters with
// hipGraphExecKernelNodeSetParams when the chain has the shape of the last one (same functions, grids, blocks) and
// launches the graph: one submission per score instead of one per kernel.
struct LaunchRecorder {
    static constexpr int kMax = 8, kArgBytes = 1024, kMaxArgs = 12;
    int n = 0;
    void* func[kMax];
    dim3 grid[kMax], block[kMax];
    unsigned lds[kMax];
    alignas(16) unsigned char blob[kMax][kArgBytes];
    void* argv[kMax][kMaxArgs];
    int nargs[kMax];
    bool overflow = false;
};
thread_local LaunchRecorder* g_launch_recorder = nullptr;

template <typename T>
inline void record_arg(LaunchRecorder* r, int k, size_t* off, const T& v) {
    const size_t a = alignof(T) > 16 ? 16 : alignof(T);
    *off = (*off + a - 1) / a * a;
    if (*off + sizeof(T) > LaunchRecorder::kArgBytes || r->nargs[k] >= LaunchRecorder::kMaxArgs) {
        r->overflow = true;
        return;
    }
    memcpy(r->blob[k] + *off, &v, sizeof(T));
    r->argv[k][r->nargs[k]++] = r->blob[k] + *off;
    *off += sizeof(T);
}
#endif
template <typename... KArgs>
inline void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, unsigned lds, hipStream_t stream,
                   typename arg_of<KArgs>::type... args) {
#ifdef SSIMU2_INSTRUMENTED_BUILD
    if (LaunchRecorder* r = g_launch_recorder) {
        if (r->n < LaunchRecorder::kMax) {
            const int k = r->n++;
            r->func[k] = (void*)kernel;
            r->grid[k] = grid;
            r->block[k] = block;
            r->lds[k] = lds;
            r->nargs[k] = 0;
            size_t off = 0;
            (record_arg<KArgs>(r, k, &off, args), ...);
        } else {
            r->overflow = true;
        }
        return;
    }
    if (LaunchTimer* t = g_launch_timer) {
        if (t->n + 2 <= t->cap) {
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, t->ev[t->n], t->ev[t->n + 1], 0, args...);
            t->n += 2;
            return;
        }
    }
#endif
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...);
}

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

// Rows per workgroup (segment length) of the marching kernel.  A segment of R rows costs R + 8
// rows of conversion and horizontal blur, so longer is cheaper; shorter gives more workgroups
// to balance.  Measured on MI355X at 4K (scripts/gpu_sweep2.sh): the full-resolution scale is
// best at ~512 workgroups (two thirds of the 768 resident slots, 6 % halo), the smaller scales
// at ~48 rows -- their few, long workgroups overlap the tail of scale 0 and, with two
// streams, the next score.  Bounds: >= 8 rows, <= 160 rows (fp32 partial sums per lane).
int march_seg_rows(const ssimu2_ctx* c, const Pyramid& p, int scale) {
    if (scale > 0 && c->seg_rows_tail_override > 0) return c->seg_rows_tail_override;
    if (scale == 0 && c->seg_rows_override > 0) return c->seg_rows_override;
    const int nstrips = (p.w[0] + MW - 1) / MW;
    int nsegs = (512 + nstrips / 2) / nstrips;
    if (nsegs < 1) nsegs = 1;
    int seg = (p.h[0] + nsegs - 1) / nsegs;  // full-resolution scale: ~512 workgroups
    if (seg < 8) seg = 8;
    if (seg > 160) seg = 160;
    // smaller scales: 48 rows, but never longer than the full-resolution segments, or their
    // workgroups would outlast scale 0's on small frames
    if (scale > 0 && seg > 48) seg = 48;
    return seg;
}

int scale_blocks(const ssimu2_ctx* c, const Pyramid& p, int s) {
    const int seg = march_seg_rows(c, p, s);
    return ((p.w[s] + MW - 1) / MW) * ((p.h[s] + seg - 1) / seg);
}

size_t partial_doubles(const ssimu2_ctx* c, const Pyramid& p) {
    size_t t = 0;
    for (int s = 0; s < p.nscales; ++s) t += (size_t)scale_blocks(c, p, s) * kStats;
    return t;
}

void free_recursive(ssimu2_ctx* c) {
    (void)hipFree(c->d_rg);
    (void)hipFree(c->d_rg_part);
    (void)hipFree(c->d_rg_dbg);
    (void)hipFree(c->d_rg_q);
    c->d_rg_q = nullptr;
    c->d_rg = c->d_rg_dbg = nullptr;
    c->d_rg_part = nullptr;
    c->cap_rg = c->cap_rg_part = c->cap_rg_dbg = 0;
}

void free_buffers(ssimu2_ctx* c) {
    (void)hipFree(c->d_ref_u8);
    (void)hipFree(c->d_dist_u8);
    (void)hipFree(c->d_lin_ref);
    (void)hipFree(c->d_lin_dist);
    (void)hipFree(c->d_partials);
    (void)hipFree(c->d_xyb_ref);
    c->d_xyb_ref = nullptr;
    c->cap_xyb = 0;
    (void)hipFree(c->d_stage);
    c->d_stage = nullptr;
    c->cap_stage = 0;
    (void)hipFree(c->d_ref_blur);
    c->d_ref_blur = nullptr;
    c->cap_blur = 0;
    free_recursive(c);
    c->d_ref_u8 = c->d_dist_u8 = nullptr;
    c->d_lin_ref = c->d_lin_dist = nullptr;
    c->d_partials = nullptr;
    c->cap_u8 = c->cap_lin = c->cap_part = 0;
}

int ensure_capacity(ssimu2_ctx* c, uint32_t w, uint32_t h) {
    const Pyramid p = make_pyramid(w, h);
    const size_t need_u8 = (size_t)w * h * 3, need_lin = p.lin_total + 4,
                 need_part = partial_doubles(c, p) + 8;
    if (c->d_ref_u8 && need_u8 <= c->cap_u8 && need_lin <= c->cap_lin && need_part <= c->cap_part)
        return SSIMU2_OK;
    // growing frees everything, which also drops a cached reference
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const size_t nu8 = need_u8 > c->cap_u8 ? need_u8 : c->cap_u8;
    const size_t nlin = need_lin > c->cap_lin ? need_lin : c->cap_lin;
    const size_t npart = need_part > c->cap_part ? need_part : c->cap_part;
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
    c->cap_part = npart;
    return SSIMU2_OK;
}

// Band-pyramid arguments of up to two frames: levels 1..nscales-1 of frames[i] into lin[i].
PyrBandArgs pyramid_args(const Pyramid& p, int nframes, const uint8_t* const* frames, float* const* lin) {
    PyrBandArgs a{};
    a.nlevels = p.nscales > 1 ? p.nscales - 1 : 0;
    for (int l = 0; l < p.nscales; ++l) {
        a.w[l] = p.w[l];
        a.h[l] = p.h[l];
        a.opitch[l] = p.w[l];  // tight rows (the linear-light pyramid); rg_launch_convert pads them
    }
    a.nframes = nframes;
    a.bands_x = (p.w[0] + PYR_BAND_W - 1) / PYR_BAND_W;
    a.bands_y = (p.h[0] + PYR_BAND_H - 1) / PYR_BAND_H;
    for (int f = 0; f < nframes; ++f) {
        a.in[f] = frames[f];
        for (int l = 0; l < a.nlevels; ++l) a.out[f][l] = lin[f] ? lin[f] + p.lin_off[l + 1] : nullptr;
    }
    return a;
}

// Linear-light pyramids of up to two frames as a launch of their own (all levels, one launch).
void launch_pyramid(ssimu2_ctx* c, const Pyramid& p, int nframes, const uint8_t* const* frames,
                    float* const* lin) {
    if (p.nscales < 2) return;
    const PyrBandArgs a = pyramid_args(p, nframes, frames, lin);
    launch(k_pyramid_bands, dim3(a.bands_x * a.bands_y * nframes), dim3(PYR_THREADS), 0, c->stream, a);
}

// float offset of scale s in the cached reference XYB buffer (scale 0 first)
size_t xyb_off(const Pyramid& p, int s) {
    size_t off = 0;
    for (int k = 0; k < s; ++k) off += (size_t)3 * p.w[k] * p.h[k];
    return off;
}

void build_plans(const ssimu2_ctx* c, const Pyramid& p, const uint8_t* d_ref, const uint8_t* d_dist,
                 bool ref_xyb_cached, MarchPlan* mp, FinalizeArgs* fa, int* total_blocks) {
    memset(mp, 0, sizeof *mp);
    memset(fa, 0, sizeof *fa);
    mp->nscales = fa->nscales = p.nscales;
    size_t poff = 0;
    int blocks = 0;
    for (int s = 0; s < p.nscales; ++s) {
        const int seg = march_seg_rows(c, p, s);
        const int nstrips = (p.w[s] + MW - 1) / MW;
        const int nb = nstrips * ((p.h[s] + seg - 1) / seg);
        blocks += nb;
        mp->blk_end[s] = blocks;
        mp->w[s] = p.w[s];
        mp->h[s] = p.h[s];
        mp->seg[s] = seg;
        mp->nstrips[s] = nstrips;
        mp->nblocks[s] = nb;
        mp->ref[s] = s == 0 ? (const void*)d_ref : (const void*)(c->d_lin_ref + p.lin_off[s]);
        mp->dist[s] = s == 0 ? (const void*)d_dist : (const void*)(c->d_lin_dist + p.lin_off[s]);
        mp->ref_xyb[s] = ref_xyb_cached ? c->d_xyb_ref + xyb_off(p, s) : nullptr;
        const bool blur_cached = ref_xyb_cached && c->d_ref_blur && c->cap_blur;
        mp->ref_s11[s] = blur_cached ? c->d_ref_blur + xyb_off(p, s) : nullptr;
        mp->part[s] = c->d_partials + poff;
        fa->part[s] = mp->part[s];
        fa->nblocks[s] = nb;
        fa->inv_pixels[s] = 1.0 / ((double)p.w[s] * (double)p.h[s]);
        poff += (size_t)nb * kStats;
    }
    *total_blocks = blocks;
}

// ---- the published-recursion modes (ssimu2_recursive.h) ------------------------------------------
// plane offset (floats per plane) of scale s in the packed recursive-mode buffers
size_t rg_plane_off(const Pyramid& p, int s) {
    size_t off = 0;
    for (int k = 0; k < s; ++k) off += (size_t)rg_pitch(p.w[k]) * p.h[k];  // rows padded to 128 floats ("Row pitch")
    return off;
}

constexpr uint64_t kRgMaxPixels = 1ull << 28;  // 21 planes of 1.33 n floats: 30 GB at this size

int rg_check_size(ssimu2_ctx* c, uint32_t w, uint32_t h) {
    if ((uint64_t)w * h > kRgMaxPixels)
        return c->fail(SSIMU2_ERR_INVALID_ARG, "recursive blur mode: image larger than 2^28 pixels");
    return SSIMU2_OK;
}

// Scratch + cache of the recursive modes for this frame size.  Growing drops a cached reference
// (its planes live here).
int rg_ensure(ssimu2_ctx* c, const Pyramid& p) {
    const size_t ntot = rg_plane_off(p, p.nscales);
    const size_t need = 21 * ntot + (size_t)rg_pitch(p.w[0]) + 16;  // + the dump row of k_rg_v_emit
    size_t need_part = 8;
    for (int s = 0; s < p.nscales; ++s) need_part += (size_t)kStats * ((p.w[s] + RG_VW - 1) / RG_VW);
    if (need > c->cap_rg || need_part > c->cap_rg_part) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_rg);
        (void)hipFree(c->d_rg_part);
        c->d_rg = nullptr;
        c->d_rg_part = nullptr;
        c->cap_rg = c->cap_rg_part = 0;
        c->have_ref = false;
        hipError_t e = hipMalloc(&c->d_rg, need * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(&c->d_rg_part, need_part * sizeof(double));
        if (e != hipSuccess) {
            (void)hipFree(c->d_rg);
            c->d_rg = nullptr;
            c->d_rg_part = nullptr;
            return c->fail(SSIMU2_ERR_OOM, "hipMalloc(recursive-blur planes: 84 bytes per pixel and scale)", e);
        }
        c->cap_rg = need;
        c->cap_rg_part = need_part;
    }
    if (!c->d_rg_q) {
        HIP_TRY(c, hipMalloc(&c->d_rg_q, 4 * sizeof(unsigned)));
        c->num_cus = c->dev.compute_units > 0 ? (int)c->dev.compute_units : 256;
    }
    if (c->rg_dbg_scale >= 0 && c->rg_dbg_scale < p.nscales) {
        const size_t nd = (size_t)24 * rg_pitch(p.w[c->rg_dbg_scale]) * p.h[c->rg_dbg_scale] + 16;  // 15 h planes + 9 v planes
        if (nd > c->cap_rg_dbg) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            (void)hipFree(c->d_rg_dbg);
            c->d_rg_dbg = nullptr;
            c->cap_rg_dbg = 0;
            const hipError_t e = hipMalloc(&c->d_rg_dbg, nd * sizeof(float));
            if (e != hipSuccess) {
                c->d_rg_dbg = nullptr;
                return c->fail(SSIMU2_ERR_OOM, "hipMalloc(recursive-blur debug planes)", e);
            }
            c->cap_rg_dbg = nd;
        }
    }
    return SSIMU2_OK;
}

// Plan of one frame's share of a recursive-mode score: `ref_frame` selects where the frame's XYB
// planes go (xa, the reference's, or xb).
void rg_build_plan(const ssimu2_ctx* c, const Pyramid& p, bool ref_frame, RgPlan* rp, int* hblocks, int* vblocks) {
    memset(rp, 0, sizeof *rp);
    const size_t ntot = rg_plane_off(p, p.nscales);
    float* xa = c->d_rg;
    float* xb = xa + 3 * ntot;
    float* cache = xb + 3 * ntot;
    float* hbuf = cache + 6 * ntot;
    rp->nscales = p.nscales;
    int hb_ = 0, vb_ = 0;
    size_t poff = 0;
    for (int s = 0; s < p.nscales; ++s) {
        const size_t off = rg_plane_off(p, s);
        rp->w[s] = p.w[s];
        rp->h[s] = p.h[s];
        rp->pitch[s] = rg_pitch(p.w[s]);
        hb_ += 3 * ((p.h[s] + RG_HL - 1) / RG_HL);
        rp->vgroups[s] = (p.w[s] + RG_VW - 1) / RG_VW;
        vb_ += 3 * rp->vgroups[s];
        rp->hblk_end[s] = hb_;
        rp->vblk_end[s] = vb_;
        rp->xa[s] = xa + 3 * off;
        rp->xb[s] = xb + 3 * off;
        rp->xout[s] = (ref_frame ? xa : xb) + 3 * off;
        rp->cache[s] = cache + 6 * off;
        rp->hbuf[s] = hbuf + 9 * off;
        rp->part[s] = c->d_rg_part + poff;
        poff += (size_t)kStats * rp->vgroups[s];
    }
    rp->dump = hbuf + 9 * ntot;
    // jobs of the persistent kernels: horizontal = one plane of one channel over 20 rows (NK planes
    // per channel), vertical = one channel of 64 columns; both listed largest scale first
    const int nk = ref_frame ? 2 : 3;
    for (int s = 0; s < p.nscales; ++s) rp->hjob_end[s] = rp->hblk_end[s] * nk;
    rp->hjobs = hb_ * nk;
    rp->hlong = rp->hjobs < (RG_HW / 2) * c->num_cus ? rp->hjobs : (RG_HW / 2) * c->num_cus;
    rp->vjobs = vb_;
    rp->q = c->d_rg_q;
    *hblocks = hb_;
    *vblocks = vb_;
}

// Instrumented builds (ssimu2_instr_rg_stop_after_scale): keep the raw planes of one scale for the
// parity tests.  After a horizontal pass its planes are copied out of hbuf; the per-pass planes of
// the vertical pass exist only in LDS, so k_rg_v_emit recomputes them into the debug buffer.
bool rg_debugging(const ssimu2_ctx* c, const Pyramid& p) {
    return c->d_rg_dbg && c->rg_dbg_scale >= 0 && c->rg_dbg_scale < p.nscales;
}

void rg_debug_keep_h(ssimu2_ctx* c, const Pyramid& p, const RgPlan& rp, bool ref) {
    const int s = c->rg_dbg_scale, nk = ref ? 2 : 3;
    const size_t n = (size_t)rg_pitch(p.w[s]) * p.h[s];
    for (int ch = 0; ch < 3; ++ch)
        for (int k = 0; k < nk; ++k)
            (void)hipMemcpyAsync(c->d_rg_dbg + (size_t)rg_plane15(ref, ch, k) * n, rp.hbuf[s] + (size_t)(ch * nk + k) * n,
                                 n * sizeof(float), hipMemcpyDeviceToDevice, c->stream);
}

// Positive-XYB planes of one frame at every scale, straight from its bytes: the band pyramid with
// XYB outputs (the recursive modes read nothing but XYB planes; no linear-light level is stored).
void rg_launch_convert(ssimu2_ctx* c, const Pyramid& p, const uint8_t* d_frame, const RgPlan& rp) {
    const uint8_t* frames[1] = {d_frame};
    float* none[1] = {nullptr};
    PyrBandArgs a = pyramid_args(p, 1, frames, none);
    a.xyb0[0] = rp.xout[0];
    a.zero4 = rp.q;
    for (int l = 0; l < a.nlevels; ++l) a.out[0][l] = rp.xout[l + 1];
    for (int l = 0; l < p.nscales && l < 6; ++l) a.opitch[l] = rp.pitch[l];
    launch(k_pyramid_bands_xyb, dim3(a.bands_x * a.bands_y), dim3(PYR_THREADS), 0, c->stream, a);
}

// The horizontal pass: one workgroup per 20 rows and channel (RG_H_PERSISTENT = 1, an A/B build: one
// workgroup per CU pulling jobs, ssimu2_recursive.h).
template <bool REF>
void rg_launch_h(ssimu2_ctx* c, bool fma, int hblocks, const RgPlan& rp) {
    if (hblocks <= 0) return;
#if RG_H_PERSISTENT
    if (fma) launch((k_rg_h_persistent<true, REF>), dim3(c->num_cus), dim3(64 * RG_HW), 0, c->stream, rp);
    else launch((k_rg_h_persistent<false, REF>), dim3(c->num_cus), dim3(64 * RG_HW), 0, c->stream, rp);
#else
    if (fma) launch((k_rg_h<true, REF>), dim3(hblocks), dim3(REF ? 128 : 192), 0, c->stream, rp);
    else launch((k_rg_h<false, REF>), dim3(hblocks), dim3(REF ? 128 : 192), 0, c->stream, rp);
#endif
}

// What depends on the reference alone: its XYB planes and mu1 = blur(x), s11 = blur(x * x) at
// every scale (the reference's linear pyramid is already enqueued).
void rg_enqueue_reference(ssimu2_ctx* c, const Pyramid& p, const uint8_t* d_ref) {
    RgPlan rp;
    int hblocks, vblocks;
    rg_build_plan(c, p, true, &rp, &hblocks, &vblocks);
    if (p.nscales == 0) return;  // a frame below 8 x 8 has no scale to score
    for (int s = 0; s < p.nscales; ++s) rp.emit[s] = rp.cache[s];
    const bool fma = c->blur_mode == SSIMU2_BLUR_RECURSIVE_FMA, dbg = rg_debugging(c, p);
    rg_launch_convert(c, p, d_ref, rp);
    rg_launch_h<true>(c, fma, hblocks, rp);
    if (dbg) rg_debug_keep_h(c, p, rp, true);
    if (fma) launch((k_rg_v_emit<true, 2>), dim3(vblocks), dim3(128), 0, c->stream, rp);
    else launch((k_rg_v_emit<false, 2>), dim3(vblocks), dim3(128), 0, c->stream, rp);
}

// One pass against the reference planes in place: XYB of the distorted frame, the recursion over
// {y, y*y, x*y}, maps, final reduction (its linear pyramid is already enqueued).
int rg_enqueue_pass(ssimu2_ctx* c, const Pyramid& p, const uint8_t* d_dist) {
    RgPlan rp;
    int hblocks, vblocks;
    rg_build_plan(c, p, false, &rp, &hblocks, &vblocks);
    const bool fma = c->blur_mode == SSIMU2_BLUR_RECURSIVE_FMA, dbg = rg_debugging(c, p);
    FinalizeArgs fa;
    memset(&fa, 0, sizeof fa);
    fa.nscales = p.nscales;
    for (int s = 0; s < p.nscales; ++s) {
        fa.part[s] = rp.part[s];
        fa.nblocks[s] = rp.vgroups[s];
        fa.inv_pixels[s] = 1.0 / ((double)p.w[s] * (double)p.h[s]);
    }
    if (p.nscales > 0) {  // a frame below 8 x 8 has no scale to score
        rg_launch_convert(c, p, d_dist, rp);
        const int vgrid = vblocks < c->num_cus ? vblocks : c->num_cus;
        rg_launch_h<false>(c, fma, hblocks, rp);
        if (fma) launch((k_rg_v<true>), dim3(vgrid), dim3(512), c->rg_v_pad, c->stream, rp);
        else launch((k_rg_v<false>), dim3(vgrid), dim3(512), c->rg_v_pad, c->stream, rp);
        if (dbg) {
            rg_debug_keep_h(c, p, rp, false);
            const int s = c->rg_dbg_scale;
            rp.emit[s] = c->d_rg_dbg + (size_t)15 * rg_pitch(p.w[s]) * p.h[s];  // [channel][{y, yy, xy}][n]
            if (fma) launch((k_rg_v_emit<true, 3>), dim3(vblocks), dim3(192), 0, c->stream, rp);
            else launch((k_rg_v_emit<false, 3>), dim3(vblocks), dim3(192), 0, c->stream, rp);
        }
    }
    launch(k_finalize, dim3(1), dim3(1024), 0, c->stream, fa, c->h_result);  // result: see enqueue_score
    HIP_TRY(c, hipGetLastError());
    c->pending = true;
    return SSIMU2_OK;
}

// Enqueue the whole score of (d_ref, d_dist) on the ctx stream.  `ref_pyramid_ready`:
// the reference's linear pyramid in d_lin_ref is already valid for this frame size.
int enqueue_score_launches(ssimu2_ctx* c, const uint8_t* d_ref, const uint8_t* d_dist, uint32_t w,
                           uint32_t h, bool ref_pyramid_ready) {
    const Pyramid p = make_pyramid(w, h);
    const bool recursive = c->blur_mode != SSIMU2_BLUR_FIR;
    if (recursive) {  // before anything is enqueued
        int rc = rg_check_size(c, w, h);
        if (rc) return rc;
        const bool had_ref = c->have_ref;
        if ((rc = rg_ensure(c, p))) return rc;
        if (ref_pyramid_ready && had_ref && !c->have_ref)
            return c->fail(SSIMU2_ERR_NO_REFERENCE, "recursive blur mode: the cached reference was dropped");
    }
    if (p.nscales > 1 && !recursive) {  // the recursive modes convert straight to XYB planes
        if (ref_pyramid_ready) {
            const uint8_t* frames[1] = {d_dist};
            float* lin[1] = {c->d_lin_dist};
            launch_pyramid(c, p, 1, frames, lin);
        } else {
            const uint8_t* frames[2] = {d_ref, d_dist};
            float* lin[2] = {c->d_lin_ref, c->d_lin_dist};
            launch_pyramid(c, p, 2, frames, lin);
        }
    }
    if (recursive) {
        if (!ref_pyramid_ready) rg_enqueue_reference(c, p, d_ref);
        return rg_enqueue_pass(c, p, d_dist);
    }
    MarchPlan mp;
    FinalizeArgs fa;
    int blocks = 0;
    build_plans(c, p, d_ref, d_dist, ref_pyramid_ready && c->d_xyb_ref != nullptr, &mp, &fa, &blocks);
    if (blocks > 0) {
        if (mp.ref_s11[0])  // reference XYB and blur(ref*ref) cached: the search's per-pass kernel
            launch(k_march_refblur, dim3(blocks), dim3(MARCH_THREADS), 0, c->stream, mp);
        else
            launch(k_march, dim3(blocks), dim3(MARCH_THREADS), 0, c->stream, mp);
    }
    // the 880-byte result goes straight into the context's page-locked mirror: no D2H copy command per score
    launch(k_finalize, dim3(1), dim3(1024), 0, c->stream, fa, c->h_result);
    HIP_TRY(c, hipGetLastError());
    c->pending = true;
    return SSIMU2_OK;
}

#ifdef SSIMU2_INSTRUMENTED_BUILD
void graph_drop(ssimu2_ctx* c) {
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    if (c->graph) (void)hipGraphDestroy(c->graph);
    c->graph_exec = nullptr;
    c->graph = nullptr;
    c->graph_n = 0;
}

// The recorded chain of one score as ONE graph launch (see LaunchRecorder).
int graph_flush(ssimu2_ctx* c, LaunchRecorder& r) {
    if (r.overflow) return c->fail(SSIMU2_ERR_INVALID_ARG, "hipGraph experiment: a score's launches do not fit the recorder");
    if (r.n == 0) return SSIMU2_OK;
    bool same = c->graph_exec != nullptr && c->graph_n == r.n;
    for (int k = 0; same && k < r.n; ++k)
        same = c->graph_func[k] == r.func[k] && c->graph_lds[k] == r.lds[k] && c->graph_grid[k].x == r.grid[k].x &&
               c->graph_grid[k].y == r.grid[k].y && c->graph_grid[k].z == r.grid[k].z && c->graph_block[k].x == r.block[k].x &&
               c->graph_block[k].y == r.block[k].y && c->graph_block[k].z == r.block[k].z;
    hipKernelNodeParams kp[LaunchRecorder::kMax];
    for (int k = 0; k < r.n; ++k) {
        memset(&kp[k], 0, sizeof kp[k]);
        kp[k].func = r.func[k];
        kp[k].gridDim = r.grid[k];
        kp[k].blockDim = r.block[k];
        kp[k].sharedMemBytes = r.lds[k];
        kp[k].kernelParams = r.argv[k];
        kp[k].extra = nullptr;
    }
    if (!same) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));  // the old executable graph may still be running
        graph_drop(c);
        HIP_TRY(c, hipGraphCreate(&c->graph, 0));
        for (int k = 0; k < r.n; ++k) {
            HIP_TRY(c, hipGraphAddKernelNode(&c->graph_node[k], c->graph, k ? &c->graph_node[k - 1] : nullptr, k ? 1 : 0, &kp[k]));
            c->graph_func[k] = r.func[k];
            c->graph_grid[k] = r.grid[k];
            c->graph_block[k] = r.block[k];
            c->graph_lds[k] = r.lds[k];
        }
        HIP_TRY(c, hipGraphInstantiate(&c->graph_exec, c->graph, nullptr, nullptr, 0));
        c->graph_n = r.n;
        ++c->graph_builds;
    } else {
        for (int k = 0; k < r.n; ++k) HIP_TRY(c, hipGraphExecKernelNodeSetParams(c->graph_exec, c->graph_node[k], &kp[k]));
    }
    HIP_TRY(c, hipGraphLaunch(c->graph_exec, c->stream));
    ++c->graph_launches;
    return SSIMU2_OK;
}
#endif

// Enqueue the whole score of (d_ref, d_dist) on the ctx stream (enqueue_score_launches above makes the launches; the
// instrumented build's hipGraph experiment submits them as one graph instead).
int enqueue_score(ssimu2_ctx* c, const uint8_t* d_ref, const uint8_t* d_dist, uint32_t w, uint32_t h, bool ref_pyramid_ready) {
#ifdef SSIMU2_INSTRUMENTED_BUILD
    if (c->use_graph && c->rg_dbg_scale < 0 && !g_launch_timer) {
        LaunchRecorder* rec = new (std::nothrow) LaunchRecorder();
        if (!rec) return c->fail(SSIMU2_ERR_OOM, "launch recorder");
        g_launch_recorder = rec;
        int rc = enqueue_score_launches(c, d_ref, d_dist, w, h, ref_pyramid_ready);
        g_launch_recorder = nullptr;
        if (rc == SSIMU2_OK) rc = graph_flush(c, *rec);
        delete rec;
        return rc;
    }
#endif
    return enqueue_score_launches(c, d_ref, d_dist, w, h, ref_pyramid_ready);
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

const char* ssimu2_version(void) { return "oavif_amd ssimu2 gfx950 v8 (pair ring, b64 taps, dword pixel loads; recursive blur: cached reference, 3 lanes per line, persistent vertical pass; placed streams; tagged option structs; device check at context creation; pinned host buffers)"; }

int ssimu2_ctx_set_blur(ssimu2_ctx* c, int mode) {
    if (!c) return SSIMU2_ERR_INVALID_ARG;
    if (mode != SSIMU2_BLUR_FIR && mode != SSIMU2_BLUR_RECURSIVE && mode != SSIMU2_BLUR_RECURSIVE_FMA)
        return c->fail(SSIMU2_ERR_INVALID_ARG, "unknown blur mode");
    if (c->pending) return c->fail(SSIMU2_ERR_INVALID_ARG, "ssimu2_ctx_set_blur: a score is still enqueued");
    if (mode == SSIMU2_BLUR_FIR && c->d_rg) {  // the recursive modes' planes are of no use to the default mode
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        free_recursive(c);
    }
    c->blur_mode = mode;
    c->have_ref = false;
    return SSIMU2_OK;
}

const char* ssimu2_last_error(const ssimu2_ctx* ctx) {
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

static int ctx_create_impl(int device, void* hip_stream, ssimu2_ctx** out_ctx);

}  // extern "C"

// ---- stream placement ---------------------------------------------------------------------------
// HIP maps streams onto a few hardware queues, and two streams that share one do not overlap at
// all: of six contexts created back to back on an MI355X the pairs (0,5), (1,4), (2,3) score at the
// one-stream rate, every other pair 13 % faster (scripts/gpu_stream_pairs.py).  A caller that
// scores independent frames on two contexts (a batch, speculative probes) cannot see this, so the
// library places the streams: once per process and device it creates a few candidate streams,
// runs short dependent spin kernels on pairs of them and keeps a set whose members all ran
// CONCURRENTLY with each other.  Contexts created without a caller stream borrow the first free
// member of that set (exclusively; ssimu2_ctx_destroy returns it), and only contexts beyond the
// set get a stream of their own on whatever queue HIP picks.
namespace {

__global__ void k_spin(long long ticks) {  // one wave; wall_clock64 runs at 100 MHz
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

constexpr int kPoolCandidates = 6, kPoolMax = 4;
struct StreamPool {
    bool built = false;
    int n = 0;
    hipStream_t stream[kPoolMax] = {nullptr, nullptr, nullptr, nullptr};
    bool in_use[kPoolMax] = {false, false, false, false};
};
std::mutex g_pool_mu;
StreamPool g_pool[64];

// Do streams a and b overlap their work?  Two dependent 100 us spin kernels on each, enqueued
// alternately.  Kernels of ONE stream are ordered by barrier packets, and a barrier packet waits
// for everything enqueued before it on its hardware queue -- the other stream's kernels too, if
// the two streams share the queue: >= 300 us then, 200 us on distinct queues.  (One kernel per
// stream proves nothing: without a barrier between them two packets of one queue run side by side.)
bool streams_overlap(hipStream_t a, hipStream_t b) {
    // Timed on the DEVICE (events on the two streams), best of four: a host clock around launch and
    // synchronize adds the launch path and the wake-up of the waiting thread to a 200-vs-300 us
    // decision, and one misread on a loaded host shrank the set to a single stream.
    const long long ticks = 10000;  // 100 us
    hipEvent_t e0 = nullptr, ea = nullptr, eb = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&ea) != hipSuccess || hipEventCreate(&eb) != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        if (ea) (void)hipEventDestroy(ea);
        (void)hipGetLastError();
        return false;
    }
    double best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipStreamSynchronize(a);
        (void)hipStreamSynchronize(b);
        (void)hipEventRecord(e0, a);
        for (int k = 0; k < 2; ++k) {
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, ticks);
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, ticks);
        }
        (void)hipEventRecord(ea, a);
        (void)hipEventRecord(eb, b);
        (void)hipStreamSynchronize(a);
        (void)hipStreamSynchronize(b);
        float ma = 0.f, mb = 0.f;
        if (hipEventElapsedTime(&ma, e0, ea) != hipSuccess || hipEventElapsedTime(&mb, e0, eb) != hipSuccess) continue;
        const double us = 1e3 * (ma > mb ? ma : mb);
        if (us < best) best = us;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    (void)hipGetLastError();
    return best < 260.0;
}

// caller holds g_pool_mu and has set the device
void build_pool(StreamPool& pool) {
    pool.built = true;
    hipStream_t cand[kPoolCandidates];
    int nc = 0;
    for (; nc < kPoolCandidates; ++nc)
        if (hipStreamCreateWithFlags(&cand[nc], hipStreamNonBlocking) != hipSuccess) break;
    if (nc > 0) {  // code object load and clocks, untimed
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, cand[0], 2000LL);
        (void)hipStreamSynchronize(cand[0]);
    }
    bool keep[kPoolCandidates] = {false};
    for (int i = 0; i < nc && pool.n < kPoolMax; ++i) {
        bool ok = true;
        for (int j = 0; j < pool.n && ok; ++j) ok = streams_overlap(pool.stream[j], cand[i]);
        if (ok) {
            pool.stream[pool.n++] = cand[i];
            keep[i] = true;
        }
    }
    for (int i = 0; i < nc; ++i)
        if (!keep[i]) (void)hipStreamDestroy(cand[i]);
    (void)hipGetLastError();
}

hipStream_t pool_acquire(int device) {
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(g_pool_mu);
    StreamPool& pool = g_pool[device];
    if (!pool.built) build_pool(pool);
    for (int i = 0; i < pool.n; ++i)
        if (!pool.in_use[i]) {
            pool.in_use[i] = true;
            return pool.stream[i];
        }
    return nullptr;
}

// streams of the placed set of `device` (0 before the first context without a caller stream exists)
int pool_size(int device) {
    if (device < 0 || device >= 64) return 0;
    std::lock_guard<std::mutex> lock(g_pool_mu);
    return g_pool[device].n;
}

void pool_release(int device, hipStream_t s) {
    if (device < 0 || device >= 64) return;
    std::lock_guard<std::mutex> lock(g_pool_mu);
    StreamPool& pool = g_pool[device];
    for (int i = 0; i < pool.n; ++i)
        if (pool.stream[i] == s) pool.in_use[i] = false;
}

}  // namespace

extern "C" {

// ssimu2_prefetch: the once-per-process cost of the scorer (HIP runtime initialisation, loading
// the code object, the constant table: 140-340 ms on the MI355X box, scripts/gpu_coldstart.py) on
// a background thread, so that a one-image run hides it behind its image load and first encode.
static std::mutex g_prefetch_mu;
static std::shared_future<void> g_prefetch[64];

int ssimu2_prefetch(int device) {
    if (device < 0 || device >= 64) return SSIMU2_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(g_prefetch_mu);
    if (g_prefetch[device].valid()) return SSIMU2_OK;
    try {
        g_prefetch[device] = std::async(std::launch::async, [device] {
            ssimu2_ctx* tmp = nullptr;  // a throw-away context does all of the above
            if (ctx_create_impl(device, nullptr, &tmp) == SSIMU2_OK) ssimu2_ctx_destroy(tmp);
        }).share();
    } catch (...) {
        return SSIMU2_ERR_OOM;
    }
    return SSIMU2_OK;
}

int ssimu2_prefetch_join(int device) {
    if (device < 0 || device >= 64) return SSIMU2_ERR_INVALID_ARG;
    std::shared_future<void> f;
    {
        std::lock_guard<std::mutex> lock(g_prefetch_mu);
        f = g_prefetch[device];
    }
    if (f.valid()) f.wait();
    return SSIMU2_OK;
}

int ssimu2_ctx_create(int device, void* hip_stream, ssimu2_ctx** out_ctx) {
    if (device >= 0 && device < 64) {
        std::shared_future<void> f;
        {
            std::lock_guard<std::mutex> lock(g_prefetch_mu);
            f = g_prefetch[device];
        }
        if (f.valid()) f.wait();  // a prefetch in flight finishes first
    }
    return ctx_create_impl(device, hip_stream, out_ctx);
}

// What the runtime and sysfs say of HIP device `device`; `why` (optional) receives the reason when the device is not
// one this library can run on.  No context, no allocation; hipGetDeviceProperties initialises the runtime.
static int query_device_impl(int device, ssimu2_device_info* d, std::string* why) {
    static std::mutex mu;                      // contexts are created from worker threads: one record per device,
    static ssimu2_device_info accepted[64];    // read from the runtime and sysfs once
    static bool have[64] = {false};
    std::lock_guard<std::mutex> lock(mu);
    if (device >= 0 && device < 64 && have[device]) {
        *d = accepted[device];
        return SSIMU2_OK;
    }
    memset(d, 0, sizeof *d);
    d->struct_size = (uint32_t)sizeof *d;
    d->device = device;
    d->numa_node = -1;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        if (why) *why = std::string("no usable HIP device (hipGetDeviceCount: ") + hipGetErrorString(e) + ", " +
                        std::to_string(ndev) + " device(s), index " + std::to_string(device) + ")";
        return SSIMU2_ERR_NO_DEVICE;
    }
    hipDeviceProp_t pr;
    if ((e = hipGetDeviceProperties(&pr, device)) != hipSuccess) {
        if (why) *why = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
        return SSIMU2_ERR_HIP;
    }
    snprintf(d->arch, sizeof d->arch, "%s", pr.gcnArchName);
    snprintf(d->name, sizeof d->name, "%s", pr.name);
    char bus[64] = "";
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess)
        snprintf(bus, sizeof bus, "%04x:%02x:%02x.0", pr.pciDomainID, pr.pciBusID, pr.pciDeviceID);
    for (char* q = bus; *q; ++q) *q = (char)tolower((unsigned char)*q);
    snprintf(d->pci_bus_id, sizeof d->pci_bus_id, "%s", bus);
    d->compute_units = pr.multiProcessorCount > 0 ? (uint32_t)pr.multiProcessorCount : 0;
    // LDS of one CU: the runtime reports it as maxSharedMemoryPerMultiProcessor; the per-workgroup limit
    // (sharedMemPerBlock) is the same number on gfx950 (a workgroup may take the whole 160 KB)
    d->lds_bytes_per_workgroup = (uint32_t)pr.sharedMemPerBlock;
    size_t per_cu = pr.maxSharedMemoryPerMultiProcessor;
    int attr = 0;
    if (hipDeviceGetAttribute(&attr, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, device) == hipSuccess && attr > 0 &&
        (size_t)attr > per_cu)
        per_cu = (size_t)attr;
    if (pr.sharedMemPerBlock > per_cu) per_cu = pr.sharedMemPerBlock;  // a workgroup's limit cannot exceed its CU's LDS
    d->lds_bytes_per_cu = (uint32_t)per_cu;
    d->wavefront_size = pr.warpSize > 0 ? (uint32_t)pr.warpSize : 0;
    d->hbm_bytes = (uint64_t)pr.totalGlobalMem;
    {
        char path[128];
        snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", d->pci_bus_id);
        if (FILE* f = fopen(path, "r")) {
            int node = -1;
            if (fscanf(f, "%d", &node) == 1) d->numa_node = node;
            fclose(f);
        }
    }
    (void)hipGetLastError();
    if (strncmp(d->arch, "gfx950", 6) != 0) {
        if (why) *why = std::string("device ") + std::to_string(device) + " is " + d->arch +
                        ", not gfx950: this library holds gfx950 (MI355X) code only";
        return SSIMU2_ERR_NO_DEVICE;
    }
    if (d->lds_bytes_per_cu < kMinLdsPerCu) {
        if (why) *why = std::string("device ") + std::to_string(device) + " (" + d->arch + ") reports " +
                        std::to_string(d->lds_bytes_per_cu) + " bytes of LDS per compute unit, LDS < 160 KB";
        return SSIMU2_ERR_NO_DEVICE;
    }
    if (d->wavefront_size != 64) {
        if (why) *why = std::string("device ") + std::to_string(device) + " runs wavefronts of " +
                        std::to_string(d->wavefront_size) + " lanes, the kernels are written for 64";
        return SSIMU2_ERR_NO_DEVICE;
    }
    d->usable = 1;
    if (device < 64) {
        accepted[device] = *d;
        have[device] = true;
    }
    return SSIMU2_OK;
}

int ssimu2_query_device(int device, ssimu2_device_info* out) {
    if (!out || out->struct_size != sizeof(ssimu2_device_info)) return SSIMU2_ERR_INVALID_ARG;
    ssimu2_device_info d;
    std::string why;
    const int rc = query_device_impl(device, &d, &why);
    if (rc == SSIMU2_ERR_NO_DEVICE && d.arch[0] == 0) {  // no such device at all
        g_create_error = why;
        return rc;
    }
    if (rc == SSIMU2_ERR_HIP) {
        g_create_error = why;
        return rc;
    }
    *out = d;  // a device of another kind is described too, with usable = 0
    return SSIMU2_OK;
}

int ssimu2_ctx_device_info(const ssimu2_ctx* c, ssimu2_device_info* out) {
    if (!c || !out || out->struct_size != sizeof(ssimu2_device_info)) return SSIMU2_ERR_INVALID_ARG;
    *out = c->dev;
    return SSIMU2_OK;
}

int ssimu2_host_alloc(ssimu2_ctx* c, size_t bytes, void** out_ptr) {
    if (!c) return SSIMU2_ERR_INVALID_ARG;
    if (!out_ptr || bytes == 0) return c->fail(SSIMU2_ERR_INVALID_ARG, "ssimu2_host_alloc: null out_ptr or zero bytes");
    *out_ptr = nullptr;
    HIP_TRY(c, hipSetDevice(c->device));
    const hipError_t e = hipHostMalloc(out_ptr, bytes, hipHostMallocPortable);
    if (e != hipSuccess) {
        *out_ptr = nullptr;
        (void)hipGetLastError();
        return c->fail(SSIMU2_ERR_OOM, "hipHostMalloc(pinned frame buffer)", e);
    }
    return SSIMU2_OK;
}

int ssimu2_host_free(ssimu2_ctx* c, void* ptr) {
    if (!c) return SSIMU2_ERR_INVALID_ARG;
    if (!ptr) return SSIMU2_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipHostFree(ptr));
    return SSIMU2_OK;
}

static int ctx_create_impl(int device, void* hip_stream, ssimu2_ctx** out_ctx) {
    if (!out_ctx) return SSIMU2_ERR_INVALID_ARG;
    *out_ctx = nullptr;
    // capability check first (include/ssimu2_hip.h, SSIMU2_ERR_NO_DEVICE): a device index that exists, gfx950,
    // 160 KB of LDS per CU -- instead of a failure at the first launch on anything else
    ssimu2_device_info info;
    {
        std::string why;
        const int rc = query_device_impl(device, &info, &why);
        if (rc != SSIMU2_OK) {
            g_create_error = why;
            return rc;
        }
    }
    ssimu2_ctx* c = new (std::nothrow) ssimu2_ctx();
    if (!c) return SSIMU2_ERR_OOM;
    c->device = device;
    c->dev = info;
    c->rg_v_pad = rg_v_pad_bytes(info.lds_bytes_per_cu);
#define CREATE_TRY(call)                                                        \
    do {                                                                        \
        hipError_t e2 = (call);                                                 \
        if (e2 != hipSuccess) {                                                 \
            g_create_error = std::string(#call) + ": " + hipGetErrorString(e2); \
            ssimu2_ctx_destroy(c);                                              \
            return SSIMU2_ERR_HIP;                                              \
        }                                                                       \
    } while (0)
    CREATE_TRY(hipSetDevice(device));
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
    } else if ((c->stream = pool_acquire(device)) != nullptr) {
        c->pool_stream = true;
    } else {
        CREATE_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    CREATE_TRY(hipEventCreate(&c->ev0));
    CREATE_TRY(hipEventCreate(&c->ev1));
    CREATE_TRY(hipMalloc(&c->d_result, 110 * sizeof(double)));
    CREATE_TRY(hipHostMalloc(&c->h_result, 110 * sizeof(double), hipHostMallocDefault));
    {
        // The constant table lives in device memory of this module, one copy per device, shared
        // by every context on that device: upload it once per device (contexts may be created
        // from worker threads while other contexts' kernels are running).
        static std::mutex mu;
        static bool uploaded[64] = {false};
        std::lock_guard<std::mutex> lock(mu);
        if (device >= 64 || !uploaded[device]) {
            DevConst* k = new (std::nothrow) DevConst();
            if (!k) {
                ssimu2_ctx_destroy(c);
                return SSIMU2_ERR_OOM;
            }
            for (int i = 0; i < 256; ++i) {
                const double v = (double)i / 255.0;
                k->lut[i] = (float)(v <= 0.04045 ? v / 12.92 : pow((v + 0.055) / 1.055, 2.4));
            }
            gaussian_taps(1.5, k->taps, k->rg_n2, k->rg_d1);
            k->cbrt_bias = cbrt_repro_host(kOpsinBias);
            memcpy(k->weights, kWeightsHost, sizeof kWeightsHost);
            hipError_t ec = hipMemcpyToSymbol(HIP_SYMBOL(c_k), k, sizeof(DevConst));
            delete k;
            CREATE_TRY(ec);
            if (device < 64) uploaded[device] = true;
        }
    }
#undef CREATE_TRY
    *out_ctx = c;
    return SSIMU2_OK;
}

void ssimu2_ctx_destroy(ssimu2_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
#ifdef SSIMU2_INSTRUMENTED_BUILD
    graph_drop(c);
#endif
    free_buffers(c);
    (void)hipFree(c->d_result);
    (void)hipHostFree(c->h_result);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    if (c->pool_stream) pool_release(c->device, c->stream);
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
    if (!out_score)
        return c ? c->fail(SSIMU2_ERR_INVALID_ARG, "null out_score") : SSIMU2_ERR_INVALID_ARG;
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

static int set_reference_impl(ssimu2_ctx* c, const void* ref, uint32_t w, uint32_t h,
                              hipMemcpyKind kind) {
    int rc = check_args(c, ref, ref, w, h);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = ensure_capacity(c, w, h))) return rc;
    const Pyramid p = make_pyramid(w, h);
    if (c->blur_mode != SSIMU2_BLUR_FIR) {  // size limit and planes of the recursive modes, before any launch
        if ((rc = rg_check_size(c, w, h))) return rc;
        if ((rc = rg_ensure(c, p))) return rc;
    }
    c->have_ref = false;
    const size_t bytes = (size_t)w * h * 3;
    HIP_TRY(c, hipMemcpyAsync(c->d_ref_u8, ref, bytes, kind, c->stream));
    if (p.nscales > 1 && c->blur_mode == SSIMU2_BLUR_FIR) {  // the reference's linear pyramid, once per search
        const uint8_t* frames[1] = {c->d_ref_u8};
        float* lin[1] = {c->d_lin_ref};
        launch_pyramid(c, p, 1, frames, lin);
    }
    if (c->blur_mode != SSIMU2_BLUR_FIR) {  // XYB planes, blur(x) and blur(x*x) by the published recursion
        rg_enqueue_reference(c, p, c->d_ref_u8);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream));  // caller may free `ref` after return
        c->have_ref = true;
        c->ref_w = w;
        c->ref_h = h;
        return SSIMU2_OK;
    }
    // ... and its positive-XYB planes at every scale, so that the per-pass kernel skips the
    // LUT / opsin / cube-root work for the reference frame (same values, bit-identical scores)
    const size_t need_xyb = xyb_off(p, p.nscales) + 4;
    if (need_xyb > c->cap_xyb) {
        (void)hipFree(c->d_xyb_ref);
        c->d_xyb_ref = nullptr;
        c->cap_xyb = 0;
        hipError_t e = hipMalloc(&c->d_xyb_ref, need_xyb * sizeof(float));
        if (e != hipSuccess) {
            // not fatal: the search still works, the reference is just converted on every pass
            c->d_xyb_ref = nullptr;
            (void)hipGetLastError();
        } else {
            c->cap_xyb = need_xyb;
        }
    }
    if (c->d_xyb_ref) {
        for (int sc = 0; sc < p.nscales; ++sc) {
            const size_t n = (size_t)p.w[sc] * p.h[sc];
            const void* in = sc == 0 ? (const void*)c->d_ref_u8 : (const void*)(c->d_lin_ref + p.lin_off[sc]);
            launch(k_ref_xyb, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, in,
                               sc == 0, p.w[sc], p.h[sc], c->d_xyb_ref + xyb_off(p, sc));
        }
    }
    // ... and blur(ref*ref) at every scale, which depends on the reference alone: the per-pass
    // kernel then blurs four planes instead of five (caching blur(ref) too was measured slower)
    if (c->d_xyb_ref && c->cache_ref_blur) {
        if (need_xyb > c->cap_blur) {
            (void)hipFree(c->d_ref_blur);
            c->d_ref_blur = nullptr;
            c->cap_blur = 0;
            hipError_t e = hipMalloc(&c->d_ref_blur, need_xyb * sizeof(float));
            if (e != hipSuccess) {  // not fatal either: the pass blurs all five planes
                c->d_ref_blur = nullptr;
                (void)hipGetLastError();
            } else {
                c->cap_blur = need_xyb;
            }
        }
        if (c->d_ref_blur) {
            MarchPlan mp;
            FinalizeArgs fa;
            int blocks = 0;
            build_plans(c, p, c->d_ref_u8, c->d_ref_u8, true, &mp, &fa, &blocks);
            for (int sc = 0; sc < p.nscales; ++sc) mp.dist[sc] = mp.ref[sc];  // second frame unused
            if (blocks > 0)
                launch(k_ref_blur, dim3(blocks), dim3(MARCH_THREADS), 0, c->stream, mp);
        }
    } else if (c->d_ref_blur) {
        (void)hipFree(c->d_ref_blur);
        c->d_ref_blur = nullptr;
        c->cap_blur = 0;
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // caller may free `ref` after return
    c->have_ref = true;
    c->ref_w = w;
    c->ref_h = h;
    return SSIMU2_OK;
}

int ssimu2_set_reference(ssimu2_ctx* c, const uint8_t* ref, uint32_t w, uint32_t h) {
    return set_reference_impl(c, ref, w, h, hipMemcpyHostToDevice);
}

int ssimu2_set_reference_device(ssimu2_ctx* c, const void* d_ref, uint32_t w, uint32_t h) {
    return set_reference_impl(c, d_ref, w, h, hipMemcpyDeviceToDevice);
}

int ssimu2_enqueue_against_reference_device(ssimu2_ctx* c, const void* d_dist) {
    if (!c) return SSIMU2_ERR_INVALID_ARG;
    if (!c->have_ref) return c->fail(SSIMU2_ERR_NO_REFERENCE, "no reference set");
    if (!d_dist) return c->fail(SSIMU2_ERR_INVALID_ARG, "null pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    return enqueue_score(c, c->d_ref_u8, (const uint8_t*)d_dist, c->ref_w, c->ref_h, true);
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

int ssimu2_score_against_reference_strided(ssimu2_ctx* c, const uint8_t* pixels, uint32_t row_bytes,
                                           uint32_t channels, double* out_score) {
    if (!c) return SSIMU2_ERR_INVALID_ARG;
    if (!c->have_ref) return c->fail(SSIMU2_ERR_NO_REFERENCE, "no reference set");
    if (!pixels || !out_score) return c->fail(SSIMU2_ERR_INVALID_ARG, "null pointer");
    if (channels != 3 && channels != 4)
        return c->fail(SSIMU2_ERR_UNSUPPORTED, "channels must be 3 (RGB) or 4 (RGBA)");
    const uint32_t w = c->ref_w, h = c->ref_h;
    if ((uint64_t)row_bytes < (uint64_t)w * channels)
        return c->fail(SSIMU2_ERR_INVALID_ARG, "row_bytes smaller than one row of pixels");
    if (channels == 3 && row_bytes == w * 3)  // already the scorer's layout
        return ssimu2_score_against_reference(c, pixels, out_score);
    HIP_TRY(c, hipSetDevice(c->device));
    // the last row needs only its pixels, not its padding (libavif allocates rowBytes * height,
    // a cropped view of a larger buffer may not)
    const size_t bytes = (size_t)row_bytes * (h - 1) + (size_t)w * channels;
    if (bytes > c->cap_stage) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_stage);
        c->d_stage = nullptr;
        c->cap_stage = 0;
        hipError_t e = hipMalloc(&c->d_stage, bytes + 16);
        if (e != hipSuccess) return c->fail(SSIMU2_ERR_OOM, "hipMalloc(staging frame)", e);
        c->cap_stage = bytes;
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_stage, pixels, bytes, hipMemcpyHostToDevice, c->stream));
    if (channels == 4 && w % 4 == 0 && row_bytes % 4 == 0)
        launch(k_unpack_rgb<true>, dim3((w / 4 + 255) / 256, h), dim3(256), 0, c->stream,
                           c->d_stage, row_bytes, channels, w, h, c->d_dist_u8);
    else
        launch(k_unpack_rgb<false>, dim3((w + 255) / 256, h), dim3(256), 0, c->stream,
                           c->d_stage, row_bytes, channels, w, h, c->d_dist_u8);
    int rc = enqueue_score(c, c->d_ref_u8, c->d_dist_u8, w, h, true);
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

}  // extern "C"
