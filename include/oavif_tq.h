/*
 * oavif_tq.h -- C ABI of the target-quality search (host logic, fp64 scalar).
 *
 * Mirrors /root/reference/src/tq.zig:
 *   TQCtx                 tq.zig:10-13   -> oavif_tq_result.{num_pass, score}
 *   PassResult            tq.zig:16-19   -> oavif_tq_pass
 *   computeScoreAtQuality tq.zig:21-38   -> the probe callback (encode at q -> decode ->
 *                                           score); oavif_tq_search_hip supplies the score
 *                                           half from the HIP scorer (ssimu2_hip.h)
 *   predictQFromScore     tq.zig:40-43   -> oavif_tq_predict_q_from_score
 *   interpolateQuantizer  tq.zig:73-122  -> oavif_tq_interpolate_quantizer
 *   findTargetQuality     tq.zig:124-210 -> oavif_tq_find_target_quality
 * The options are the three fields of AvifEncOptions the search reads
 * (/root/reference/src/parse_args.zig:55,58,59; ranges parse_args.zig:85-86,101-104).
 *
 * libavif/libaom encode and dav1d decode stay on the CPU and stay the caller's: they enter
 * through callbacks, exactly where tq.zig:24 and tq.zig:26 call io.encodeAvifToBuffer /
 * io.decodeAvifToRgb.
 */
#ifndef OAVIF_TQ_H_
#define OAVIF_TQ_H_

#include <stddef.h>
#include <stdint.h>

#include "ssimu2_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden */
#endif

#define OAVIF_TQ_MAX_PASS 12 /* parse_args.zig:104 */

typedef struct {
    double score_tgt;  /* -t/--score-tgt, 30..100, default 80   (parse_args.zig:55,85-86)  */
    double tolerance;  /* --tolerance,    1..100,  default 2    (parse_args.zig:58,101-102) */
    uint32_t max_pass; /* --max-pass,     1..12,   default 6    (parse_args.zig:59,103-104) */
} oavif_tq_options;

typedef struct {
    uint32_t q;   /* quantizer probed */
    double score; /* SSIMULACRA2 score of that probe */
} oavif_tq_pass;

typedef struct {
    uint32_t q;          /* EncCtx.q after the search: the chosen quantizer (main.zig:29)   */
    double score;        /* TQCtx.score: score reported for the chosen q (tq.zig:12)       */
    uint32_t num_pass;   /* TQCtx.num_pass: probes actually encoded+scored (tq.zig:11,29)  */
    int32_t buf_q;       /* EncBuffer.q: q of the LAST probe, whose AVIF bytes the caller
                            still holds (tq.zig:31-35); the caller re-encodes iff
                            buf_q != q (main.zig:109-113).  -1 if no probe ran.          */
    uint32_t history_len;
    oavif_tq_pass history[OAVIF_TQ_MAX_PASS]; /* probes in the order they were made */
} oavif_tq_result;

/* One search pass: produce the score of quantizer `q` (tq.zig:21-38).  Return 0 on
   success; any other value aborts the search and is returned by the search function
   (the reference's `try` at tq.zig:150). */
typedef int (*oavif_tq_probe_fn)(void* user, uint32_t q, double* out_score);

/* Encode `q` -> decode -> tight RGB8 into out_rgb (w*h*3 bytes, caller = search owns it).
   CPU codec of the caller (io.zig:544-636 + io.zig:638-666).  *out_avif_size receives
   the size of the encoded AVIF (EncBuffer.size).  Return 0 on success. */
typedef int (*oavif_tq_codec_fn)(void* user, uint32_t q, uint8_t* out_rgb, size_t* out_avif_size);

void oavif_tq_default_options(oavif_tq_options* o);

/* tq.zig:40-43 */
uint32_t oavif_tq_predict_q_from_score(double tgt);

/* tq.zig:73-122; history in probe order (it is copied and sorted inside). */
uint32_t oavif_tq_interpolate_quantizer(uint32_t lo_bound, uint32_t hi_bound,
                                        const oavif_tq_pass* history, uint32_t history_len,
                                        double target);

/* tq.zig:124-210 with the pass (tq.zig:21-38) supplied by `probe`. */
int oavif_tq_find_target_quality(const oavif_tq_options* o, oavif_tq_probe_fn probe, void* user,
                                 oavif_tq_result* out);

/* The same search with the scorer half of every pass done by the HIP scorer: `ref_rgb`
   (w*h*3, tight RGB8 = EncCtx.rgb) is uploaded once, each pass calls `codec` then scores
   the decoded frame on the GPU.  Equivalent to findTargetQuality with tq.zig:37 bound to
   ssimu2_score_rgb8. */
int oavif_tq_search_hip(const oavif_tq_options* o, ssimu2_ctx* scorer, const uint8_t* ref_rgb,
                        uint32_t w, uint32_t h, oavif_tq_codec_fn codec, void* user,
                        oavif_tq_result* out, size_t* out_last_avif_size);

/* ---- speculative probe fan-out (one search, several probes in flight) -------------------
 *
 * The reference runs the passes of a search strictly one after another: the next quantizer
 * depends on every score so far (tq.zig:135-139).  On a host with idle cores and a scorer
 * that takes 0.2 ms, the probes that the search is LIKELY to ask for next can be encoded,
 * decoded and scored at the same time as the one it asked for (one scorer context / HIP
 * stream each), and the search replayed over the cached scores.
 *
 * oavif_tq_find_target_quality_speculative runs exactly the control flow of
 * oavif_tq_find_target_quality.  When it needs the score of a quantizer nobody has probed
 * yet, it issues one WAVE through `batch`: that quantizer first, plus up to max_fanout-1
 * candidates for the following pass -- found by running the same search code forward under
 * hypothetical scores for the missing quantizer (an estimate from the probes known so far,
 * or from the model of tq.zig:40-43, shifted by tolerance+0.5, +1.5, ... points either way).
 * Wrong guesses cost CPU time only: the result -- q, score, num_pass, buf_q, history -- is
 * that of the sequential search as long as score(q) is a function of q (the HIP scorer is
 * deterministic).  num_pass keeps the reference's meaning (passes the sequential search
 * would have run); the extra work is reported in oavif_tq_spec_stats.
 */
#define OAVIF_TQ_MAX_FANOUT 16

/* Probe `n` distinct quantizers (encode -> decode -> score each; they may run concurrently)
   and store their scores in out_scores[0..n).  qs[0] is the one the search is waiting for.
   Return 0 on success; any other value aborts the search and is returned by it. */
typedef int (*oavif_tq_batch_probe_fn)(void* user, const uint32_t* qs, uint32_t n, double* out_scores);

/* ABI guard of oavif_tq_spec_options: a tag in the upper half + the struct's size as the CALLER compiled it in the
   lower half.  The struct grew once (round 2's layout was {max_fanout, first_wave_fanout}); a plain size would not
   tell that layout apart -- its max_fanout = 12 reads as "size 12" -- while no legal member value (fan-outs are
   1..OAVIF_TQ_MAX_FANOUT) reaches the tag.  The library accepts exactly the value of its own header and refuses
   everything else with SSIMU2_ERR_INVALID_ARG.  OAVIF_TQ_SPEC_OPTIONS_INIT fills it in. */
#define OAVIF_TQ_SPEC_OPTIONS_TAG 0x71530000u
typedef struct {
    uint32_t struct_size; /* OAVIF_TQ_SPEC_OPTIONS_TAG | sizeof(oavif_tq_spec_options) */
    uint32_t max_fanout;  /* probes per wave, 1..OAVIF_TQ_MAX_FANOUT; 1 = the sequential search */
    /* Probes of the FIRST wave, 1..max_fanout; 0 = max_fanout.  The first probe of a search is the
       model's guess (tq.zig:40-43), and many searches end on it: with 1 the first wave is that
       probe alone, so a one-pass search costs exactly what the sequential search costs (no extra
       encodes competing for the host's cores), and speculation starts with the second wave, when
       there is a measured score to extrapolate from. */
    uint32_t first_wave_fanout;
} oavif_tq_spec_options;
#define OAVIF_TQ_SPEC_OPTIONS_INIT(max_fanout_, first_wave_fanout_) \
    { OAVIF_TQ_SPEC_OPTIONS_TAG | (uint32_t)sizeof(oavif_tq_spec_options), (max_fanout_), (first_wave_fanout_) }

typedef struct {
    uint32_t waves;         /* calls of `batch` (the latency of the search, in passes)          */
    uint32_t probes_issued; /* quantizers encoded + scored in total (>= result.num_pass)        */
    uint32_t cache_hits;    /* passes of the search answered by an earlier wave                 */
} oavif_tq_spec_stats;

int oavif_tq_find_target_quality_speculative(const oavif_tq_options* o,
                                             const oavif_tq_spec_options* so,
                                             oavif_tq_batch_probe_fn batch, void* user,
                                             oavif_tq_result* out, oavif_tq_spec_stats* stats);

/* ---- source pre-scaling, hoisted out of the pass loop ------------------------------------
 *
 * io.encodeAvifToBuffer rescales the whole source to the output depth on EVERY pass
 * (io.zig:566-617: a fresh w*h*channels buffer and one loop over it), although the source
 * never changes during a search -- only `quality` does (io.zig:625).  These are the three
 * loops as functions, so that a caller computes the scaled buffer once per search and hands
 * the same pointer to avifImageRGBToYUV on every pass.  Integer
 * arithmetic, identical values.  `n` = w * h * channels elements.
 */
void oavif_prescale_8_to_10(const uint8_t* src, size_t n, uint16_t* dst);   /* (v*1023+127)/255  io.zig:572 */
void oavif_prescale_16_to_10(const uint16_t* src, size_t n, uint16_t* dst); /* v >> 6            io.zig:587 */
void oavif_prescale_16_to_8(const uint16_t* src, size_t n, uint8_t* dst);   /* v >> 8            io.zig:602 */

/* ---- PNG ingest -------------------------------------------------------
 *
 * io.loadPNG (io.zig:242-307) decodes through libspng with these output rules, which decide
 * what the encoder and the scorer are fed:
 *     bit depth 16       -> RGBA16, host-endian u16, channels = 4, hbd = true   (io.zig:270-272,292)
 *     8-bit truecolour   -> RGB8, channels = 3                                   (io.zig:275)
 *     everything else    -> RGBA8, channels = 4: gray, gray + alpha, palette, RGBA; sub-byte
 *                           gray is scaled to 8 bits                             (io.zig:276-280)
 * and the iCCP profile is handed on decompressed (io.zig:261-268).  The reference decodes with
 * flags 0 (io.zig:285), i.e. without SPNG_DECODE_TRNS: a tRNS chunk is NOT applied and files
 * without an alpha channel come out opaque (this reading of libspng is unpinned: libspng is not in
 * the image and the reference holds no PNG fixture).  These two functions are that loader without
 * libspng: the PNG specification (chunk CRCs, the five row filters -- Sub / Average / Paeth of 3- and 4-byte
 * pixels in SSE2 --, Adam7, PLTE / iCCP) with zlib for the CRCs and the iCCP profile and the library's own
 * resumable DEFLATE decoder for the IDAT stream (csrc/inflate_fast.h: 1.5-1.75 x zlib, checked against zlib
 * stream for stream), inflated strip by strip (the compressed stream + 256 KB or one row of memory, whatever the
 * header claims).  A header that
 * promises more scanline bytes than its IDAT data can inflate to (deflate's 1032 : 1 bound) fails
 * with OAVIF_PNG_ERR_DECODE in both calls.  Host code; no GPU involved.
 *
 * oavif_png_info_from_memory parses the file and reports the OUTPUT geometry; oavif_png_decode
 * writes `data_bytes` of pixels (2-byte aligned when hbd) and, if `out_icc` is non-NULL, the
 * `icc_bytes` of the profile.  Error codes mirror the reference's Zig errors.
 */
enum {
    OAVIF_PNG_OK = 0,
    OAVIF_PNG_ERR_ARG = -1,    /* null pointer / misaligned u16 output                            */
    OAVIF_PNG_ERR_HEADER = -2, /* error.GetHeaderFailed: not a PNG, bad IHDR                      */
    OAVIF_PNG_ERR_DECODE = -3, /* error.DecodeFailed: CRC, chunk order, zlib, filter, palette     */
    OAVIF_PNG_ERR_SIZE = -4,   /* error.ImageSizeFailed: size overflow, output buffer too small   */
    OAVIF_PNG_ERR_OOM = -5
};
typedef struct {
    uint32_t width, height;
    uint32_t channels;   /* of the OUTPUT: 3 or 4 (io.zig:287-290)                     */
    int hbd;             /* 1 = the output is u16 per sample (16-bit source)           */
    size_t data_bytes;   /* width * height * channels * (hbd ? 2 : 1)                  */
    size_t icc_bytes;    /* decompressed iCCP profile, 0 = none                        */
    uint32_t bit_depth, color_type, interlaced; /* of the file (IHDR)                 */
} oavif_png_info;
int oavif_png_info_from_memory(const uint8_t* png, size_t len, oavif_png_info* out);
int oavif_png_decode(const uint8_t* png, size_t len, uint8_t* out_pixels, size_t out_cap, uint8_t* out_icc,
                     size_t icc_cap);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* OAVIF_TQ_H_ */
