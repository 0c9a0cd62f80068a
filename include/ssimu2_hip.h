/*
 * ssimu2_hip.h -- C ABI of the MI355X (gfx950) SSIMULACRA2 scorer.
 *
 * Drop-in boundary for the single scorer call of oavif's target-quality search:
 *
 *     /root/reference/src/tq.zig:37
 *         return try fssimu2.computeSsimu2(allocator, e.rgb, decoded_rgb, e.w, e.h, 3, null);
 *
 * (`fssimu2` = third-party Zig module, /root/reference/build.zig.zon:7-10, wired in at
 * /root/reference/build.zig:30-33,65).  A Zig shim with that exact signature that forwards
 * to these entry points is in oavif_amd/zig/fssimu2.zig; INTEGRATION.md shows the build
 * wiring.  Plain pointers and sizes only; no C++/torch types cross this boundary; no
 * exceptions; no aborts.
 *
 * Input contract (the reference's): `ref` and `dist` are tightly packed 8-bit interleaved
 * RGB, row-major, w*h*3 bytes (main.zig:86, io.zig:57-133 for ref; io.zig:647-663 for dist).
 * The caller owns both buffers; the library never frees or retains host pointers after a
 * call returns (dist is freed by the caller right after the call: tq.zig:26-27).
 *
 * Threading: one ssimu2_ctx = one HIP stream + its device scratch; a ctx is not
 * re-entrant; distinct ctxs are independent (different streams and/or devices).
 */
#ifndef SSIMU2_HIP_H_
#define SSIMU2_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden */
#endif

typedef struct ssimu2_ctx ssimu2_ctx;

/* Return codes (0 = success).  The Zig shim maps them to Zig errors consumed by the
   `try` at tq.zig:37. */
enum {
    SSIMU2_OK = 0,
    SSIMU2_ERR_INVALID_ARG = -1,   /* null pointer, zero dimension                      */
    SSIMU2_ERR_UNSUPPORTED = -2,   /* channels != 3 (the reference always passes 3)     */
    SSIMU2_ERR_OOM = -3,           /* host or device allocation failed                  */
    SSIMU2_ERR_HIP = -4,           /* a HIP runtime call or kernel launch failed        */
    SSIMU2_ERR_NO_REFERENCE = -5,  /* score_against_reference without set_reference     */
    SSIMU2_ERR_NO_DEVICE = -6      /* no usable gfx950 device: no HIP device of that index, a device
                                      that is not gfx950 (the library holds gfx950 code only), or one
                                      with less than 160 KB of LDS per compute unit                    */
};

/* Number of per-scale statistics and scales of the published algorithm: 6 scales x
   (3 channels x {L1,L4} SSIM + 3 channels x {L1,L4} x {artifact, detail_lost}) = 108. */
#define SSIMU2_NUM_SCALES 6
#define SSIMU2_STATS_PER_SCALE 18

/* Create a scorer bound to HIP device `device`.  If `hip_stream` is non-NULL it is a
   hipStream_t owned by the caller that all work of this ctx is enqueued on.  If NULL the library
   provides a non-blocking stream, and places it: HIP maps streams onto a few hardware queues and
   two streams on one queue do not overlap at all, so once per process and device the library
   probes a few streams and keeps a set that run side by side; the first contexts created borrow
   those (up to four; three or four are found with HIP's default of four hardware queues, depending on
   where the null stream sits), so the scores of any two of them
   overlap (13 % more throughput at 4K than two contexts that share a queue); further contexts
   get a stream on whatever queue HIP picks. */
int ssimu2_ctx_create(int device, void* hip_stream, ssimu2_ctx** out_ctx);

/* What the library sees of a HIP device, and what ssimu2_ctx_create checks before anything is allocated: the
   architecture must be gfx950 (`arch` starts with "gfx950": the code object holds nothing else, and a launch on
   another device would fail at the first kernel) and a compute unit must have 160 KB of LDS (the vertical pass of
   the recursive blur modes holds one workgroup per CU by asking for more than half a CU's LDS; on a 64 KB part the
   launch would be refused).  A device that fails either check gives SSIMU2_ERR_NO_DEVICE from ssimu2_ctx_create,
   with the reason in ssimu2_last_error(NULL).
   ssimu2_query_device fills the record for ANY HIP device index (also one that ctx_create would refuse: `usable`
   says which) without creating a context; ssimu2_ctx_device_info returns the record the context was created with.
   `struct_size` must be sizeof(ssimu2_device_info) as the caller compiled it (SSIMU2_ERR_INVALID_ARG otherwise).
   Multi-GPU launchers use pci_bus_id / numa_node to prove that every rank sits on a device of its own
   (bench.py's and the batch driver's `collective` record). */
typedef struct {
    uint32_t struct_size;
    int32_t device;                   /* HIP device index the record describes                              */
    char arch[64];                    /* gcnArchName, e.g. "gfx950:sramecc+:xnack-"                           */
    char name[128];                   /* marketing name                                                      */
    char pci_bus_id[32];              /* "0000:05:00.0" (domain:bus:device.function), lower case               */
    uint32_t compute_units;
    uint32_t lds_bytes_per_cu;        /* what the 160 KB check reads                                          */
    uint32_t lds_bytes_per_workgroup; /* largest LDS allocation of one workgroup                              */
    uint32_t wavefront_size;
    uint64_t hbm_bytes;               /* total device memory                                                  */
    int32_t numa_node;                /* host NUMA node the device hangs off (sysfs), -1 = unknown            */
    int32_t usable;                   /* 1 = ssimu2_ctx_create accepts the device                             */
} ssimu2_device_info;
int ssimu2_query_device(int device, ssimu2_device_info* out);
int ssimu2_ctx_device_info(const ssimu2_ctx* ctx, ssimu2_device_info* out);

/* Page-locked ("pinned") host memory for frames handed to the host-pointer entry points below.  The reference
   decodes every probe into a buffer libavif allocates (io.zig:452-482: avifRGBImageAllocatePixels) and the scorer
   then reads it from pageable memory, which the HIP runtime first copies into a staging buffer of its own.  A host
   that points avifRGBImage.pixels at a buffer from ssimu2_host_alloc instead (libavif fills any caller-provided
   `pixels` / `rowBytes`) lets the upload run as one DMA from the decoder's own output.  Any pointer works with every
   entry point; pinned ones only skip the staging copy.  The memory is visible to every device of the process; free
   it with ssimu2_host_free (NULL is a no-op).  `ctx` names the device context of the allocation and receives the
   error text. */
int ssimu2_host_alloc(ssimu2_ctx* ctx, size_t bytes, void** out_ptr);
int ssimu2_host_free(ssimu2_ctx* ctx, void* ptr);

/* Optional: start the once-per-process initialisation of `device` (HIP runtime, code object,
   constant table -- 140-340 ms) on a background thread and return at once.  A later
   ssimu2_ctx_create for that device waits for it and is then quick.  Meant for one-image runs
   such as the oavif CLI: call it first thing in main(), and the cost disappears behind the image
   load and the first encode (main.zig:62-101, tq.zig:24).  Harmless to call twice or without a
   GPU (ssimu2_ctx_create then reports SSIMU2_ERR_NO_DEVICE as usual). */
int ssimu2_prefetch(int device);
/* Wait until a prefetch started for `device` has finished (returns at once if there is none).
   A process that called ssimu2_prefetch and then exits early -- before it ever creates a context
   -- calls this first, so that HIP start-up on the background thread does not race process
   teardown.  ssimu2_ctx_create does the same wait by itself. */
int ssimu2_prefetch_join(int device);
void ssimu2_ctx_destroy(ssimu2_ctx* ctx);

/* Which blur the scorer evaluates.  fssimu2's source is not available to this repository, so
   which of the two it follows is not known:
     SSIMU2_BLUR_FIR        (default; the THROUGHPUT mode) the 9-tap impulse response of the
                            published sigma-1.5 recursive Gaussian, zero padding, fused kernels --
                            what bench.py's `value` measures: 0.156 ms per 4K score with two contexts in
                            flight, 0.176 on one (profiles/r06_bench.json; by size, one context / two:
                            512x512 28 / 18 us, 1920x1080 68 / 45 us, 7680x4320 0.65 / 0.63 ms);
     SSIMU2_BLUR_RECURSIVE  (the CONSERVATIVE-PARITY mode) the published recursion itself (libjxl
                            FastGaussian: three second-order sections, products rounded to fp32
                            first, horizontal then vertical), operation for operation, planes
                            bit-identical to the CPU checker's.  0.36 ms per 4K pass against a
                            reference set with ssimu2_set_reference (whose XYB planes, blur(x) and
                            blur(x*x) are then cached, so a pass recurses 9 of the 15 planes),
                            0.70 ms for a pair score (profiles/r06_bench.json; passes at 512x512 /
                            1920x1080 / 7680x4320: 0.071 / 0.154 / 1.32 ms).  Both modes are checked
                            against this repository's CPU checker only: parity against fssimu2 itself
                            is UNPINNED (its source is not available here).  This is the mode
                            the search path runs by default (the Zig shim, the CLI mirror, the batch
                            driver and the C host set it); a bare context starts in SSIMU2_BLUR_FIR.
   The two differ by the recursion's own fp32 rounding noise, which grows with the line length:
   median 0.02 points on 384x256 frames, 0.13 at 1080p, 0.47 (max 2.4) at 4K; against the operator
   accumulated in fp64 the FIR form is within 0.0005 at 4K, the recursion about 1 point off.
   Applies to every later score of the ctx; a cached reference is dropped, and switching back to
   SSIMU2_BLUR_FIR frees the recursive modes' planes.  Frames of more than 2^28 pixels are refused
   in the recursive modes (by ssimu2_set_reference and by every scoring call, before anything is
   enqueued).  Device memory of a ctx in these modes: 112 bytes per pixel (84 per pixel and scale:
   0.40 GB of reference cache + 0.53 GB of per-pass planes at 4K), so the sixteen contexts of a
   probe fan-out hold 15 GB at 4K. */
enum { SSIMU2_BLUR_FIR = 0, SSIMU2_BLUR_RECURSIVE = 1,
       /* the same recursion with its last multiply-subtract fused, fma(-d1, prev, .), the way a
          compiler targeting an FMA unit contracts the published scalar code; the two orders are
          0.6 points apart on a 4K probe */
       SSIMU2_BLUR_RECURSIVE_FMA = 2 };
int ssimu2_ctx_set_blur(ssimu2_ctx* ctx, int mode);

/* Human-readable description of the last error on this ctx ("" if none).  The pointer
   stays valid until the next call on the ctx.  ctx == NULL returns the last creation
   error of the calling thread. */
const char* ssimu2_last_error(const ssimu2_ctx* ctx);

/* == fssimu2.computeSsimu2(allocator, ref, dist, w, h, channels, null)   (tq.zig:37) ==
   Host buffers in, one double out; blocking.  Uploads both frames, runs the pyramid,
   downloads the score. */
int ssimu2_score_rgb8(ssimu2_ctx* ctx, const uint8_t* ref, const uint8_t* dist, uint32_t w,
                      uint32_t h, uint32_t channels, double* out_score);

/* The search scores many `dist` frames against one fixed `ref` (tq.zig:37 passes the same
   e.rgb on every pass, main.zig:86).  set_reference uploads `ref` once and keeps only
   device copies (its linear-light pyramid); score_against_reference then uploads and
   scores one `dist`.  Besides the linear-light pyramid, the reference's positive-XYB planes and
   blur(ref*ref) -- one of the five blurs, which depends on the reference alone -- are cached at
   every scale, so each pass skips the colour conversion of the reference frame and that blur
   (about 0.3 GB of device memory per context at 4K).
   Results are bit-identical to ssimu2_score_rgb8 on the same pair. */
int ssimu2_set_reference(ssimu2_ctx* ctx, const uint8_t* ref, uint32_t w, uint32_t h);
int ssimu2_score_against_reference(ssimu2_ctx* ctx, const uint8_t* dist, double* out_score);

/* Decoded-frame hand-off (replaces the copy loop of io.decodeAvifToRgb, io.zig:638-666, together
   with tq.zig:37).  The reference decodes a probe into libavif's avifRGBImage -- 8 bits per
   channel (rgb.depth = 8, io.zig:470-471), RGB or RGBA (io.zig:473), rows `row_bytes` apart --
   and then copies it pixel by pixel on the CPU into a tight RGB buffer for the scorer
   (io.zig:654-663).  This entry point takes the avifRGBImage buffer as it is
   (`pixels` = rgb.pixels, `row_bytes` = rgb.rowBytes, `channels` = 3 or 4), uploads it and drops
   the alpha bytes / row padding on the device, then scores it against the cached reference.
   The score is bit-identical to ssimu2_score_against_reference on the CPU-copied frame.
   Errors: SSIMU2_ERR_UNSUPPORTED for channels other than 3 or 4, SSIMU2_ERR_INVALID_ARG for
   row_bytes < w * channels. */
int ssimu2_score_against_reference_strided(ssimu2_ctx* ctx, const uint8_t* pixels,
                                           uint32_t row_bytes, uint32_t channels, double* out_score);

/* Device-resident variants: `d_ref` / `d_dist` are device pointers (same RGB8 layout)
   valid on the ctx's device.  _enqueue only enqueues on the ctx stream and returns;
   ssimu2_wait blocks until the enqueued score is done and returns it.  Used by the
   batch driver and bench.py (inputs already in HBM), and for fanning speculative
   quantizer probes over several ctxs/streams. */
int ssimu2_score_rgb8_device(ssimu2_ctx* ctx, const void* d_ref, const void* d_dist,
                             uint32_t w, uint32_t h, double* out_score);
int ssimu2_enqueue_rgb8_device(ssimu2_ctx* ctx, const void* d_ref, const void* d_dist,
                               uint32_t w, uint32_t h);
int ssimu2_wait(ssimu2_ctx* ctx, double* out_score);

/* Reference-cached scoring with device-resident frames: set_reference_device copies `d_ref`
   (device pointer, RGB8 layout) into the ctx and builds the same caches as
   ssimu2_set_reference; enqueue_against_reference_device scores one device-resident `d_dist`
   against it without blocking (finish with ssimu2_wait). */
int ssimu2_set_reference_device(ssimu2_ctx* ctx, const void* d_ref, uint32_t w, uint32_t h);
int ssimu2_enqueue_against_reference_device(ssimu2_ctx* ctx, const void* d_dist);

/* The 108 plane averages behind the last finished score, [scale][18] with
   18 = 6 SSIM (channel*2 + {L1,L4}) then 12 edge (channel*4 + {art L1, art L4, det L1,
   det L4}); scales not evaluated (image too small) are zero.  For parity tests. */
int ssimu2_last_averages(ssimu2_ctx* ctx, double out[SSIMU2_NUM_SCALES * SSIMU2_STATS_PER_SCALE],
                         int* out_num_scales);

/* Library/build description, e.g. "oavif_amd ssimu2 gfx950 v8 (...)". */
const char* ssimu2_version(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* SSIMU2_HIP_H_ */
