/* TEST INFRASTRUCTURE -- a stand-in for the scorer half of include/ssimu2_hip.h on the CPU, so that the compiled
 * host (oavif_amd/csrc/oavif_host.c) can run its whole search path -- lazy scorer creation, the decoded-frame
 * hand-off, the pthread fan-out of oavif_tq_find_target_quality_speculative, the kept probe buffers -- under
 * ASan / UBSan / TSan in the CPU suite (tests/test_c_host.py).  It is NOT SSIMULACRA2 and is linked into nothing
 * but that test binary: the "score" is 100 minus a multiple of the mean absolute difference of the frames, a
 * deterministic, monotone function of the probe, which is all the search logic needs. */
#include <stdlib.h>
#include <string.h>

#include "ssimu2_hip.h"

struct ssimu2_ctx {
    uint8_t* ref;
    uint32_t w, h;
    int blur;
    char err[64];
};

int ssimu2_prefetch(int device) { (void)device; return SSIMU2_OK; }
int ssimu2_prefetch_join(int device) { (void)device; return SSIMU2_OK; }
const char* ssimu2_version(void) { return "stub scorer (tests/c/stub_scorer.c)"; }

int ssimu2_ctx_create(int device, void* hip_stream, ssimu2_ctx** out_ctx) {
    (void)device; (void)hip_stream;
    if (!out_ctx) return SSIMU2_ERR_INVALID_ARG;
    *out_ctx = (ssimu2_ctx*)calloc(1, sizeof(ssimu2_ctx));
    return *out_ctx ? SSIMU2_OK : SSIMU2_ERR_OOM;
}
void ssimu2_ctx_destroy(ssimu2_ctx* c) {
    if (c) { free(c->ref); free(c); }
}
int ssimu2_ctx_set_blur(ssimu2_ctx* c, int mode) {
    if (!c || mode < 0 || mode > 2) return SSIMU2_ERR_INVALID_ARG;
    c->blur = mode;
    free(c->ref); /* "a cached reference is dropped" */
    c->ref = NULL;
    return SSIMU2_OK;
}
const char* ssimu2_last_error(const ssimu2_ctx* c) { return c ? c->err : "stub: no context"; }
int ssimu2_set_reference(ssimu2_ctx* c, const uint8_t* ref, uint32_t w, uint32_t h) {
    if (!c || !ref || !w || !h) return SSIMU2_ERR_INVALID_ARG;
    free(c->ref);
    c->ref = (uint8_t*)malloc((size_t)w * h * 3);
    if (!c->ref) return SSIMU2_ERR_OOM;
    memcpy(c->ref, ref, (size_t)w * h * 3);
    c->w = w; c->h = h;
    return SSIMU2_OK;
}
int ssimu2_score_against_reference_strided(ssimu2_ctx* c, const uint8_t* pixels, uint32_t row_bytes, uint32_t channels,
                                           double* out_score) {
    if (!c || !pixels || !out_score || (channels != 3 && channels != 4) || row_bytes < (size_t)c->w * channels)
        return SSIMU2_ERR_INVALID_ARG;
    if (!c->ref) { strcpy(c->err, "no reference set"); return SSIMU2_ERR_NO_REFERENCE; }
    unsigned long long sad = 0;
    for (uint32_t y = 0; y < c->h; ++y)
        for (uint32_t x = 0; x < c->w; ++x)
            for (uint32_t k = 0; k < 3; ++k) {
                const int d = (int)pixels[(size_t)y * row_bytes + (size_t)x * channels + k] - (int)c->ref[((size_t)y * c->w + x) * 3 + k];
                sad += (unsigned)(d < 0 ? -d : d);
            }
    *out_score = 100.0 - 6.0 * (double)sad / ((double)c->w * c->h * 3);
    return SSIMU2_OK;
}
int ssimu2_score_against_reference(ssimu2_ctx* c, const uint8_t* dist, double* out_score) { /* referenced by tq.cpp */
    return c ? ssimu2_score_against_reference_strided(c, dist, c->w * 3, 3, out_score) : SSIMU2_ERR_INVALID_ARG;
}
/* page-locked frame buffers (round 5): plain heap memory here -- the host's caller-provided-pixels path runs as it
   does on the GPU box, and ASan checks the buffer's bounds (w * h * 4 bytes) against libavif's writes */
int ssimu2_host_alloc(ssimu2_ctx* c, size_t bytes, void** out_ptr) {
    if (!c || !out_ptr || !bytes) return SSIMU2_ERR_INVALID_ARG;
    *out_ptr = malloc(bytes);
    return *out_ptr ? SSIMU2_OK : SSIMU2_ERR_OOM;
}
int ssimu2_host_free(ssimu2_ctx* c, void* ptr) {
    if (!c) return SSIMU2_ERR_INVALID_ARG;
    free(ptr);
    return SSIMU2_OK;
}
