/*
 * ssimu2_hip_internal.h -- measurement and parity hooks of the MI355X SSIMULACRA2 scorer.
 *
 * NOT part of the drop-in boundary (include/ssimu2_hip.h) and NOT exported by liboavif_hip.so.
 * These entry points live in liboavif_hip_instr.so, a second build of the same scorer sources
 * (oavif_amd/csrc/ssimu2_instrument.hip) loaded only by bench.py, scripts/ and the tests that
 * compare intermediate planes.  Contexts of the two libraries are not interchangeable.
 */
#ifndef SSIMU2_HIP_INTERNAL_H_
#define SSIMU2_HIP_INTERNAL_H_

#include "ssimu2_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden */
#endif

/* Parity hook: download one intermediate plane set of the last score / reference.
   `what`: SSIMU2_DEBUG_LIN_REF / _LIN_DIST = linear-light pyramid level `scale` (1..5) of the
   reference / distorted frame, SSIMU2_DEBUG_XYB_REF = cached positive-XYB planes of the
   reference at `scale` (0..5; needs ssimu2_set_reference), SSIMU2_DEBUG_REF_BLUR = the cached
   blur(ref*ref) planes of that reference (written by the marching body in emit mode: the blur
   waves' arithmetic, downloadable).  `out` receives 3 planes of w_s*h_s floats; returns
   SSIMU2_ERR_INVALID_ARG if that level does not exist. */
enum { SSIMU2_DEBUG_LIN_REF = 0, SSIMU2_DEBUG_LIN_DIST = 1, SSIMU2_DEBUG_XYB_REF = 2, SSIMU2_DEBUG_REF_BLUR = 3,
       /* SSIMU2_BLUR_RECURSIVE: the FIFTEEN planes (5 * channel + {x, y, xx, yy, xy}) after the
          horizontal / after both passes, of the scale chosen with ssimu2_instr_rg_stop_after_scale
          (`out`: 15 planes) */
       SSIMU2_DEBUG_RG_H = 4, SSIMU2_DEBUG_RG_V = 5 };
int ssimu2_debug_download(ssimu2_ctx* ctx, int what, int scale, uint32_t w, uint32_t h, float* out,
                          uint32_t* out_w, uint32_t* out_h);

/* Timing hook: enqueue `iters` back-to-back scores of the same device pair bracketed by HIP
   events on the ctx stream; returns total device milliseconds. */
int ssimu2_time_device(ssimu2_ctx* ctx, const void* d_ref, const void* d_dist, uint32_t w,
                       uint32_t h, int iters, float* out_ms_total, double* out_score);

/* Roofline hook: average device milliseconds of one execution of a stage of the score,
   measured with HIP events on the ctx stream around `iters` back-to-back repetitions of that
   stage alone (a full score runs first so every input is valid).  SSIMU2_STAGE_MARCH is the
   single fused launch that covers all six scales (the dominant kernel); SSIMU2_STAGE_PYRAMID
   the 1-2 launches that build the linear-light pyramid; SSIMU2_STAGE_FINALIZE the final
   reduction. */
enum { SSIMU2_STAGE_PYRAMID = 0, SSIMU2_STAGE_MARCH = 1, SSIMU2_STAGE_FINALIZE = 2 };
int ssimu2_time_stage(ssimu2_ctx* ctx, const void* d_ref, const void* d_dist, uint32_t w, uint32_t h,
                      int stage, int iters, float* out_ms_avg);

/* The same for the dominant kernel only, rotating over `npairs` distinct device-resident pairs
   (d_refs[i], d_dists[i]; all w x h) so that the inputs of consecutive launches come from HBM,
   not from the 256 MiB Infinity Cache: launch j reads pair j % npairs.  The linear-light pyramids
   of all pairs are built first (not timed) into scratch owned by this call, and ~30 ms of the same
   launches run untimed before the timed ones (clock settling after the idle gap of the set-up). */
int ssimu2_time_march_rotating(ssimu2_ctx* ctx, const void* const* d_refs, const void* const* d_dists,
                               int npairs, uint32_t w, uint32_t h, int iters, float* out_ms_avg);

/* Measurement aid: the HBM read-stream ceiling of the ctx's device, measured with a plain
   16-byte-per-lane read kernel over a scratch buffer of `bytes` (use well over the 256 MiB
   Infinity Cache, e.g. 2 GiB), `iters` launches on the ctx stream timed with HIP events.
   *out_gbps = bytes / average launch time. */
int ssimu2_measure_read_stream(ssimu2_ctx* ctx, size_t bytes, int iters, double* out_gbps);

/* Experiment knobs (the product always uses the defaults).  Segment rows: rows per workgroup of
   the marching kernel at scale 0 / at the other scales, 8..160, 0 = the default rule; they only
   regroup the fp64 partial sums (last bits of the score).  Reference blur cache: whether
   ssimu2_set_reference also caches blur(ref*ref). */
int ssimu2_instr_set_segment_rows(ssimu2_ctx* ctx, int rows_scale0, int rows_other_scales);
int ssimu2_instr_cache_reference_blur(ssimu2_ctx* ctx, int enabled);
/* The marching body as a plain blur stage (k_ref_blur: XYB planes of one frame in, one blurred
   plane per channel out, all scales in one launch) timed over `iters` launches rotating over the
   plane sets of `nframes` device-resident RGB8 frames (HBM-fed).  *out_bytes_per_launch: the
   algorithmic bytes of one launch (every plane element read once, written once). */
int ssimu2_time_blur_stage_rotating(ssimu2_ctx* ctx, const void* const* d_frames, int nframes, uint32_t w,
                                    uint32_t h, int iters, float* out_ms_avg, double* out_bytes_per_launch);
/* Every kernel of a score, timed where it runs.  `iters` (<= 512) scores are enqueued through the library's own enqueue path
   with each launch made through hipExtLaunchKernelGGL and a start / stop event pair: the kernel's duration as its dispatch
   packet recorded it (what rocprofv3's kernel trace reads), no packet added between the launches of a score.
     d_refs == NULL: reference-cached passes against `d_ref` (ssimu2_set_reference_device is called here), rotating over the
                     n (<= 256) device-resident distorted frames d_dists[] so that every pass is HBM-fed.  Launches per pass --
                     FIR: k_pyramid_bands, k_march_refblur, k_finalize; recursive: k_pyramid_bands_xyb, k_rg_h, k_rg_v, k_finalize.
     d_refs != NULL: pair scores of (d_refs[i], d_dists[i]).  FIR: k_pyramid_bands, k_march, k_finalize; recursive: the
                     reference's three launches, then the pass's four.
   out_ms_avg[k] (room for 8) = average milliseconds of the k-th launch of a score; *out_launches = launches per score;
   *out_ms_wall_timed / *out_ms_wall_plain = stream time per score (one event pair around all the scores) with and without
   the per-launch timestamps -- wall_plain minus the sum of the kernels is what the launches of a score wait between them. */
int ssimu2_time_kernels(ssimu2_ctx* ctx, const void* d_ref, const void* const* d_refs, const void* const* d_dists, int n,
                        uint32_t w, uint32_t h, int iters, float* out_ms_avg, int* out_launches, float* out_ms_wall_timed,
                        float* out_ms_wall_plain);
/* The hipGraph experiment: with `enabled` = 1 every score of this context is submitted as ONE launch of
   an instantiated graph -- a chain of kernel nodes, one per launch of the score, kept per context and rewritten per score with
   hipGraphExecKernelNodeSetParams as long as the chain keeps its shape (same kernels, grids, blocks), rebuilt otherwise --
   instead of one hipLaunchKernelGGL per kernel.  Same kernels, same arguments, same order: the bits of a score do not change.
   `enabled` < 0 only reads the counters: graphs built / graph launches so far.  The measured outcome is in
   profiles/r06_graph_ab.log; the product library does not have this path. */
int ssimu2_instr_use_graph(ssimu2_ctx* ctx, int enabled, unsigned long long* out_builds, unsigned long long* out_launches);
/* Stream placement (ssimu2_hip.hip "stream placement"): how many streams on distinct hardware queues this
   library instance holds for ctx's device -- contexts created without a caller stream borrow them in turn.
   3 with HIP's default of four hardware queues; 1 would mean every probe misread (tests/test_gpu_streams.py). */
int ssimu2_instr_placed_streams(ssimu2_ctx* ctx, int* out_n);
/* recursive modes: keep the 15 raw planes of `scale` downloadable (SSIMU2_DEBUG_RG_H / _RG_V) --
   the horizontal-pass planes are copied out of the pass buffer, the vertical-pass planes of the
   distorted frame, which otherwise exist only in LDS, are recomputed into a debug buffer; the score
   of such a run is the normal one.  Negative = keep nothing.  (The name is round 2's, when the run
   stopped after that scale.) */
int ssimu2_instr_rg_stop_after_scale(ssimu2_ctx* ctx, int scale);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* SSIMU2_HIP_INTERNAL_H_ */
