// Measurement and parity hooks of the MI355X SSIMULACRA2 scorer -- NOT part of the product.
//
// This translation unit is the product translation unit (ssimu2_hip.hip, included below: same
// kernels, same host code, same flags) plus the entry points of include/ssimu2_hip_internal.h.
// It is built into liboavif_hip_instr.so, which only bench.py, scripts/ and a few tests load;
// liboavif_hip.so -- what a caller links -- contains none of this.
#define SSIMU2_INSTRUMENTED_BUILD 1
#include "ssimu2_hip.hip"

#include "../../include/ssimu2_hip_internal.h"

namespace ssimu2 {

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

extern "C" {

int ssimu2_instr_set_segment_rows(ssimu2_ctx* c, int rows_scale0, int rows_other_scales) {
    if (!c) return SSIMU2_ERR_INVALID_ARG;
    if ((rows_scale0 != 0 && (rows_scale0 < 8 || rows_scale0 > 160)) ||
        (rows_other_scales != 0 && (rows_other_scales < 8 || rows_other_scales > 160)))
        return c->fail(SSIMU2_ERR_INVALID_ARG, "segment rows must be 0 (default rule) or 8..160");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->seg_rows_override = rows_scale0;
    c->seg_rows_tail_override = rows_other_scales;
    free_buffers(c);  // the partial-sum buffer is sized by the segment rule
    c->have_ref = false;
    return SSIMU2_OK;
}

int ssimu2_instr_use_graph(ssimu2_ctx* c, int enabled, unsigned long long* out_builds, unsigned long long* out_launches) {
    if (!c) return SSIMU2_ERR_INVALID_ARG;
    if (c->pending) return c->fail(SSIMU2_ERR_INVALID_ARG, "ssimu2_instr_use_graph: a score is still enqueued");
    if (enabled >= 0) c->use_graph = enabled != 0;
    if (out_builds) *out_builds = c->graph_builds;
    if (out_launches) *out_launches = c->graph_launches;
    return SSIMU2_OK;
}

int ssimu2_instr_placed_streams(ssimu2_ctx* c, int* out_n) {
    if (!c || !out_n) return SSIMU2_ERR_INVALID_ARG;
    *out_n = pool_size(c->device);
    return SSIMU2_OK;
}

int ssimu2_instr_rg_stop_after_scale(ssimu2_ctx* c, int scale) {
    if (!c) return SSIMU2_ERR_INVALID_ARG;
    c->rg_dbg_scale = scale < 0 || scale >= kNumScales ? -1 : scale;
    return SSIMU2_OK;
}

int ssimu2_instr_cache_reference_blur(ssimu2_ctx* c, int enabled) {
    if (!c) return SSIMU2_ERR_INVALID_ARG;
    c->cache_ref_blur = enabled != 0;
    c->have_ref = false;
    return SSIMU2_OK;
}

int ssimu2_debug_download(ssimu2_ctx* c, int what, int scale, uint32_t w, uint32_t h, float* out,
                          uint32_t* out_w, uint32_t* out_h) {
    if (!c || !out) return SSIMU2_ERR_INVALID_ARG;
    if (w == 0 || h == 0) return c->fail(SSIMU2_ERR_INVALID_ARG, "zero image dimension");
    const Pyramid p = make_pyramid(w, h);
    const float* src = nullptr;
    if (what == SSIMU2_DEBUG_LIN_REF || what == SSIMU2_DEBUG_LIN_DIST) {
        if (scale < 1 || scale >= p.nscales || !c->d_lin_ref) return c->fail(SSIMU2_ERR_INVALID_ARG, "no such level");
        src = (what == SSIMU2_DEBUG_LIN_REF ? c->d_lin_ref : c->d_lin_dist) + p.lin_off[scale];
    } else if (what == SSIMU2_DEBUG_XYB_REF) {
        if (scale < 0 || scale >= p.nscales || !c->d_xyb_ref || !c->have_ref || c->ref_w != w || c->ref_h != h)
            return c->fail(SSIMU2_ERR_INVALID_ARG, "no cached reference XYB for that level");
        src = c->d_xyb_ref + xyb_off(p, scale);
    } else if (what == SSIMU2_DEBUG_RG_H || what == SSIMU2_DEBUG_RG_V) {
        // 15 raw planes of the scale selected with ssimu2_instr_rg_stop_after_scale before the score
        if (scale < 0 || scale >= p.nscales || scale != c->rg_dbg_scale || !c->d_rg_dbg || !c->d_rg)
            return c->fail(SSIMU2_ERR_INVALID_ARG, "no recursive-blur planes kept for that scale");
        // device planes keep their rows rg_pitch(w) floats apart (ssimu2_recursive.h "Row pitch"); `out` is tight
        const size_t pitch = (size_t)rg_pitch(p.w[scale]), wd = (size_t)p.w[scale], ht = (size_t)p.h[scale];
        const size_t nd = pitch * ht, n1 = wd * ht;
        if (24 * nd > c->cap_rg_dbg) return c->fail(SSIMU2_ERR_INVALID_ARG, "recursive-blur planes are of another frame size");
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        auto plane = [&](float* dst, const float* src) {
            return hipMemcpy2D(dst, wd * sizeof(float), src, pitch * sizeof(float), wd * sizeof(float), ht, hipMemcpyDeviceToHost);
        };
        if (what == SSIMU2_DEBUG_RG_H) {
            for (int k = 0; k < 15; ++k) HIP_TRY(c, plane(out + (size_t)k * n1, c->d_rg_dbg + (size_t)k * nd));
        } else {
            // x, xx: the reference cache [channel][{mu1, s11}]; y, yy, xy: k_rg_v_emit's planes
            const float* cache = c->d_rg + 6 * rg_plane_off(p, p.nscales) + 6 * rg_plane_off(p, scale);
            const float* pass = c->d_rg_dbg + 15 * nd;
            for (int ch = 0; ch < 3; ++ch) {
                for (int k = 0; k < 2; ++k)
                    HIP_TRY(c, plane(out + (size_t)rg_plane15(true, ch, k) * n1, cache + (size_t)(ch * 2 + k) * nd));
                for (int k = 0; k < 3; ++k)
                    HIP_TRY(c, plane(out + (size_t)rg_plane15(false, ch, k) * n1, pass + (size_t)(ch * 3 + k) * nd));
            }
        }
        if (out_w) *out_w = (uint32_t)p.w[scale];
        if (out_h) *out_h = (uint32_t)p.h[scale];
        return SSIMU2_OK;
    } else if (what == SSIMU2_DEBUG_REF_BLUR) {
        if (scale < 0 || scale >= p.nscales || !c->d_ref_blur || !c->have_ref || c->ref_w != w || c->ref_h != h)
            return c->fail(SSIMU2_ERR_INVALID_ARG, "no cached reference blur for that level");
        src = c->d_ref_blur + xyb_off(p, scale);
    } else {
        return c->fail(SSIMU2_ERR_INVALID_ARG, "bad `what`");
    }
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const size_t n = (size_t)3 * p.w[scale] * p.h[scale];
    HIP_TRY(c, hipMemcpy(out, src, n * sizeof(float), hipMemcpyDeviceToHost));
    if (out_w) *out_w = (uint32_t)p.w[scale];
    if (out_h) *out_h = (uint32_t)p.h[scale];
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

int ssimu2_time_stage(ssimu2_ctx* c, const void* d_ref, const void* d_dist, uint32_t w, uint32_t h,
                      int stage, int iters, float* out_ms_avg) {
    int rc = check_args(c, d_ref, d_dist, w, h);
    if (rc) return rc;
    if (iters <= 0 || !out_ms_avg) return c->fail(SSIMU2_ERR_INVALID_ARG, "bad iters/out");
    double score;
    if ((rc = ssimu2_score_rgb8_device(c, d_ref, d_dist, w, h, &score))) return rc;  // valid inputs
    const Pyramid p = make_pyramid(w, h);
    MarchPlan mp;
    FinalizeArgs fa;
    int blocks = 0;
    build_plans(c, p, (const uint8_t*)d_ref, (const uint8_t*)d_dist, false, &mp, &fa, &blocks);
    if (stage < 0 || stage > 2) return c->fail(SSIMU2_ERR_INVALID_ARG, "bad stage");
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < iters; ++i) {
        if (stage == SSIMU2_STAGE_PYRAMID && p.nscales > 1) {
            const uint8_t* frames[2] = {(const uint8_t*)d_ref, (const uint8_t*)d_dist};
            float* lin[2] = {c->d_lin_ref, c->d_lin_dist};
            launch_pyramid(c, p, 2, frames, lin);
        } else if (stage == SSIMU2_STAGE_MARCH && blocks > 0) {
            hipLaunchKernelGGL(k_march, dim3(blocks), dim3(MARCH_THREADS), 0, c->stream, mp);
        } else if (stage == SSIMU2_STAGE_FINALIZE) {
            hipLaunchKernelGGL(k_finalize, dim3(1), dim3(1024), 0, c->stream, fa, c->d_result);
        }
    }
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *out_ms_avg = ms / (float)iters;
    return SSIMU2_OK;
}

int ssimu2_time_march_rotating(ssimu2_ctx* c, const void* const* d_refs, const void* const* d_dists,
                               int npairs, uint32_t w, uint32_t h, int iters, float* out_ms_avg) {
    if (!c) return SSIMU2_ERR_INVALID_ARG;
    if (!d_refs || !d_dists || npairs <= 0 || npairs > 64 || iters <= 0 || !out_ms_avg)
        return c->fail(SSIMU2_ERR_INVALID_ARG, "bad pairs/iters/out");
    int rc = check_args(c, d_refs[0], d_dists[0], w, h);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = ensure_capacity(c, w, h))) return rc;
    c->have_ref = false;
    const Pyramid p = make_pyramid(w, h);
    // per-pair linear-light pyramids (what the marching kernel reads at scales >= 1)
    const size_t lin_floats = p.lin_total + 4;
    float* lin = nullptr;
    hipError_t e = hipMalloc(&lin, (size_t)npairs * 2 * lin_floats * sizeof(float));
    if (e != hipSuccess) return c->fail(SSIMU2_ERR_OOM, "hipMalloc(rotating pyramids)", e);
    MarchPlan* plans = new (std::nothrow) MarchPlan[npairs];
    if (!plans) {
        (void)hipFree(lin);
        return c->fail(SSIMU2_ERR_OOM, "plans");
    }
    int blocks = 0;
    for (int i = 0; i < npairs; ++i) {
        if (!d_refs[i] || !d_dists[i]) {
            delete[] plans;
            (void)hipFree(lin);
            return c->fail(SSIMU2_ERR_INVALID_ARG, "null pair pointer");
        }
        float* lr = lin + (size_t)(2 * i) * lin_floats;
        float* ld = lin + (size_t)(2 * i + 1) * lin_floats;
        if (p.nscales > 1) {
            const uint8_t* frames[2] = {(const uint8_t*)d_refs[i], (const uint8_t*)d_dists[i]};
            float* lins[2] = {lr, ld};
            launch_pyramid(c, p, 2, frames, lins);
        }
        FinalizeArgs fa;
        build_plans(c, p, (const uint8_t*)d_refs[i], (const uint8_t*)d_dists[i], false, &plans[i], &fa, &blocks);
        for (int s = 1; s < p.nscales; ++s) {
            plans[i].ref[s] = lr + p.lin_off[s];
            plans[i].dist[s] = ld + p.lin_off[s];
        }
    }
    float ms = 0.f;
    if (blocks > 0) {
        // untimed: ~30 ms of the same launches first.  Allocating the scratch above leaves the GPU
        // idle for a moment, and an MI355X that has been idle runs its next ~100 launches 5-15 %
        // slower while its clocks come back up (kernel-trace of bench.py: 166 -> 154 -> 144 -> 141 us)
        const int warm = npairs * 2 > 192 ? npairs * 2 : 192;
        for (int j = 0; j < warm; ++j)
            hipLaunchKernelGGL(k_march, dim3(blocks), dim3(MARCH_THREADS), 0, c->stream, plans[j % npairs]);
        e = hipEventRecord(c->ev0, c->stream);
        for (int j = 0; j < iters && e == hipSuccess; ++j)
            hipLaunchKernelGGL(k_march, dim3(blocks), dim3(MARCH_THREADS), 0, c->stream, plans[j % npairs]);
        if (e == hipSuccess) e = hipEventRecord(c->ev1, c->stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipEventSynchronize(c->ev1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev0, c->ev1);
    } else {
        e = hipStreamSynchronize(c->stream);
    }
    (void)hipStreamSynchronize(c->stream);
    delete[] plans;
    (void)hipFree(lin);
    if (e != hipSuccess) return c->fail(SSIMU2_ERR_HIP, "rotating march timing", e);
    *out_ms_avg = ms / (float)iters;
    return SSIMU2_OK;
}

// The marching body as a PLAIN blur stage (k_ref_blur: positive-XYB planes of one frame in, one
// blurred plane per channel out, all scales in one launch), rotating over `nframes` frames' plane
// sets so that inputs and outputs come from / go to HBM, not the Infinity Cache.
int ssimu2_time_blur_stage_rotating(ssimu2_ctx* c, const void* const* d_frames, int nframes, uint32_t w,
                                    uint32_t h, int iters, float* out_ms_avg, double* out_bytes_per_launch) {
    if (!c) return SSIMU2_ERR_INVALID_ARG;
    if (!d_frames || nframes <= 0 || nframes > 16 || iters <= 0 || !out_ms_avg)
        return c->fail(SSIMU2_ERR_INVALID_ARG, "bad frames/iters/out");
    int rc = check_args(c, d_frames[0], d_frames[0], w, h);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = ensure_capacity(c, w, h))) return rc;
    c->have_ref = false;
    const Pyramid p = make_pyramid(w, h);
    const size_t planes = xyb_off(p, p.nscales) + 4;  // floats of one plane set (all scales)
    float* buf = nullptr;
    hipError_t e = hipMalloc(&buf, (size_t)nframes * 2 * planes * sizeof(float));
    if (e != hipSuccess) return c->fail(SSIMU2_ERR_OOM, "hipMalloc(rotating blur-stage planes)", e);
    MarchPlan* plans = new (std::nothrow) MarchPlan[nframes];
    if (!plans) {
        (void)hipFree(buf);
        return c->fail(SSIMU2_ERR_OOM, "plans");
    }
    int blocks = 0;
    for (int i = 0; i < nframes; ++i) {
        const uint8_t* f = (const uint8_t*)d_frames[i];
        float* xyb = buf + (size_t)(2 * i) * planes;
        float* blur = buf + (size_t)(2 * i + 1) * planes;
        if (p.nscales > 1) {
            const uint8_t* frames[1] = {f};
            float* lins[1] = {c->d_lin_ref};
            launch_pyramid(c, p, 1, frames, lins);
        }
        for (int sc = 0; sc < p.nscales; ++sc) {
            const size_t n = (size_t)p.w[sc] * p.h[sc];
            const void* in = sc == 0 ? (const void*)f : (const void*)(c->d_lin_ref + p.lin_off[sc]);
            hipLaunchKernelGGL(k_ref_xyb, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, in, sc == 0,
                               p.w[sc], p.h[sc], xyb + xyb_off(p, sc));
        }
        FinalizeArgs fa;
        build_plans(c, p, f, f, false, &plans[i], &fa, &blocks);
        for (int sc = 0; sc < p.nscales; ++sc) {
            plans[i].dist[sc] = plans[i].ref[sc];
            plans[i].ref_xyb[sc] = xyb + xyb_off(p, sc);
            plans[i].ref_s11[sc] = blur + xyb_off(p, sc);
        }
    }
    float ms = 0.f;
    if (blocks > 0) {
        for (int j = 0; j < 64; ++j)  // clocks (see ssimu2_time_march_rotating)
            hipLaunchKernelGGL(k_ref_blur, dim3(blocks), dim3(MARCH_THREADS), 0, c->stream, plans[j % nframes]);
        e = hipEventRecord(c->ev0, c->stream);
        for (int j = 0; j < iters && e == hipSuccess; ++j)
            hipLaunchKernelGGL(k_ref_blur, dim3(blocks), dim3(MARCH_THREADS), 0, c->stream, plans[j % nframes]);
        if (e == hipSuccess) e = hipEventRecord(c->ev1, c->stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipEventSynchronize(c->ev1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev0, c->ev1);
    }
    (void)hipStreamSynchronize(c->stream);
    delete[] plans;
    (void)hipFree(buf);
    if (e != hipSuccess) return c->fail(SSIMU2_ERR_HIP, "rotating blur-stage timing", e);
    *out_ms_avg = ms / (float)iters;
    // algorithmic bytes of one launch: every plane element read once and written once
    if (out_bytes_per_launch) *out_bytes_per_launch = 2.0 * (double)xyb_off(p, p.nscales) * sizeof(float);
    return SSIMU2_OK;
}

// Every kernel of a score timed where it runs: `iters` scores enqueued through the product's own enqueue_score() while a
// timing scope is open, so each launch is made with hipExtLaunchKernelGGL and a start / stop event pair -- the kernel's
// duration as its dispatch packet recorded it (what rocprofv3's kernel trace reads), no packet added between the launches.
//   d_refs == NULL: reference-cached passes against `d_ref` (set here with ssimu2_set_reference_device), rotating over
//                   the distorted frames d_dists[0..n-1]   (FIR: pyramid, k_march_refblur, k_finalize;
//                                                            recursive: k_pyramid_bands_xyb, k_rg_h, k_rg_v, k_finalize)
//   d_refs != NULL: pair scores of (d_refs[i], d_dists[i])  (FIR: pyramid, k_march, k_finalize;
//                                                            recursive: the reference's three launches, then the pass's four)
// out_ms_avg[k] = average milliseconds of the k-th launch of a score, *out_launches = launches per score (<= 8);
// *out_ms_wall_timed / *out_ms_wall_plain = stream time per score (events around all `iters` scores) of the timed run and
// of the same run with plain launches: the difference is what the per-launch timestamps cost.
int ssimu2_time_kernels(ssimu2_ctx* c, const void* d_ref, const void* const* d_refs, const void* const* d_dists, int n,
                        uint32_t w, uint32_t h, int iters, float* out_ms_avg, int* out_launches, float* out_ms_wall_timed,
                        float* out_ms_wall_plain) {
    if (!c) return SSIMU2_ERR_INVALID_ARG;
    if (!d_dists || n <= 0 || n > 256 || iters <= 0 || iters > 512 || !out_ms_avg || !out_launches)
        return c->fail(SSIMU2_ERR_INVALID_ARG, "bad frames/iters/out");
    if (!d_refs && !d_ref) return c->fail(SSIMU2_ERR_INVALID_ARG, "neither a cached reference nor pairs");
    int rc = check_args(c, d_refs ? d_refs[0] : d_ref, d_dists[0], w, h);
    if (rc) return rc;
    for (int i = 0; i < n; ++i)
        if (!d_dists[i] || (d_refs && !d_refs[i])) return c->fail(SSIMU2_ERR_INVALID_ARG, "null frame pointer");
    const bool cached = d_refs == nullptr;
    if (cached) {
        if ((rc = ssimu2_set_reference_device(c, d_ref, w, h))) return rc;
    } else {
        HIP_TRY(c, hipSetDevice(c->device));
        if ((rc = ensure_capacity(c, w, h))) return rc;
        c->have_ref = false;
    }
    auto one = [&](int j) {
        return cached ? enqueue_score(c, c->d_ref_u8, (const uint8_t*)d_dists[j % n], w, h, true)
                      : enqueue_score(c, (const uint8_t*)d_refs[j % n], (const uint8_t*)d_dists[j % n], w, h, false);
    };
    constexpr int kMaxLaunches = 8;
    const int nev = 2 * kMaxLaunches * iters;
    hipEvent_t* ev = new (std::nothrow) hipEvent_t[nev];
    if (!ev) return c->fail(SSIMU2_ERR_OOM, "events");
    int made = 0;
    hipError_t e = hipSuccess;
    for (; made < nev; ++made)
        if ((e = hipEventCreate(&ev[made])) != hipSuccess) break;
    double sum[kMaxLaunches] = {0};
    float wall_timed = 0.f, wall_plain = 0.f;
    int per_score = 0;
    if (e == hipSuccess) {
        const int warm = 2 * n > 24 ? 2 * n : 24;  // clocks and caches as in a run of scores
        for (int j = 0; j < warm && rc == 0; ++j) rc = one(j);
        // plain launches first: the stream time per score without the timestamps
        if (rc == 0) e = hipEventRecord(c->ev0, c->stream);
        for (int j = 0; j < iters && rc == 0; ++j) rc = one(j);
        if (rc == 0 && e == hipSuccess) e = hipEventRecord(c->ev1, c->stream);
        if (rc == 0 && e == hipSuccess) e = hipEventSynchronize(c->ev1);
        if (rc == 0 && e == hipSuccess) e = hipEventElapsedTime(&wall_plain, c->ev0, c->ev1);
        // the same scores with a start / stop event pair on every launch
        LaunchTimer timer{ev, 0, nev};
        if (rc == 0 && e == hipSuccess) e = hipEventRecord(c->ev0, c->stream);
        g_launch_timer = &timer;
        for (int j = 0; j < iters && rc == 0; ++j) rc = one(j);
        g_launch_timer = nullptr;
        if (rc == 0 && e == hipSuccess) e = hipEventRecord(c->ev1, c->stream);
        if (rc == 0 && e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (rc == 0 && e == hipSuccess) e = hipEventElapsedTime(&wall_timed, c->ev0, c->ev1);
        per_score = timer.n / 2 / iters;
        if (rc == 0 && e == hipSuccess && (timer.n % (2 * iters) != 0 || per_score < 1 || per_score > kMaxLaunches)) {
            rc = c->fail(SSIMU2_ERR_INVALID_ARG, "the scores did not all make the same number of launches");
        }
        for (int j = 0; j < iters && rc == 0 && e == hipSuccess; ++j)
            for (int k = 0; k < per_score && e == hipSuccess; ++k) {
                float ms = 0.f;
                e = hipEventElapsedTime(&ms, ev[2 * (j * per_score + k)], ev[2 * (j * per_score + k) + 1]);
                sum[k] += ms;
            }
    }
    (void)hipStreamSynchronize(c->stream);
    c->pending = false;
    for (int i = 0; i < made; ++i) (void)hipEventDestroy(ev[i]);
    delete[] ev;
    if (rc) return rc;
    if (e != hipSuccess) return c->fail(SSIMU2_ERR_HIP, "per-kernel timing", e);
    for (int k = 0; k < per_score; ++k) out_ms_avg[k] = (float)(sum[k] / iters);
    *out_launches = per_score;
    if (out_ms_wall_timed) *out_ms_wall_timed = wall_timed / (float)iters;
    if (out_ms_wall_plain) *out_ms_wall_plain = wall_plain / (float)iters;
    return SSIMU2_OK;
}

int ssimu2_measure_read_stream(ssimu2_ctx* c, size_t bytes, int iters, double* out_gbps) {
    if (!c) return SSIMU2_ERR_INVALID_ARG;
    if (bytes < (1u << 20) || iters <= 0 || !out_gbps)
        return c->fail(SSIMU2_ERR_INVALID_ARG, "bad bytes/iters/out");
    HIP_TRY(c, hipSetDevice(c->device));
    void* buf = nullptr;
    hipError_t e = hipMalloc(&buf, bytes + 64);
    if (e != hipSuccess) return c->fail(SSIMU2_ERR_OOM, "hipMalloc(read-stream scratch)", e);
    uint32_t* sink = (uint32_t*)((uint8_t*)buf + (bytes & ~(size_t)15));
    int rc = SSIMU2_OK;
    float ms = 0.f;
    const size_t n16 = bytes / 16;
    const int grid = 256 * 16;  // 16 workgroups of 4 waves per CU: the CUs' full wave capacity
    if ((e = hipMemsetAsync(buf, 0, bytes + 64, c->stream)) != hipSuccess) goto hip_fail;
    hipLaunchKernelGGL(k_read_stream, dim3(grid), dim3(256), 0, c->stream, (const uint4*)buf, n16, sink);
    if ((e = hipEventRecord(c->ev0, c->stream)) != hipSuccess) goto hip_fail;
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL(k_read_stream, dim3(grid), dim3(256), 0, c->stream, (const uint4*)buf, n16, sink);
    if ((e = hipEventRecord(c->ev1, c->stream)) != hipSuccess) goto hip_fail;
    if ((e = hipGetLastError()) != hipSuccess) goto hip_fail;
    if ((e = hipEventSynchronize(c->ev1)) != hipSuccess) goto hip_fail;
    if ((e = hipEventElapsedTime(&ms, c->ev0, c->ev1)) != hipSuccess) goto hip_fail;
    *out_gbps = (double)(n16 * 16) / ((double)ms / iters * 1e-3) * 1e-9;
    (void)hipFree(buf);
    return rc;
hip_fail:
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(buf);
    return c->fail(SSIMU2_ERR_HIP, "read-stream probe", e);
}

}  // extern "C"
