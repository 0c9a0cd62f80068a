/* Plain C consumer of include/ssimu2_hip.h + include/oavif_tq.h: what a C (or, through the
 * same symbols, Zig) host does.  Built and run by tests/test_c_consumer.py.
 *
 *   abi_smoke W H      -> prints "score <pair> <cached> <identical> passes <n> q <q>"
 *
 * Frames are generated here (xorshift), so the Python side can regenerate them and compare
 * the score with its own call through ctypes. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oavif_tq.h"
#include "ssimu2_hip.h"

static uint32_t xs = 2463534242u;
static uint32_t rnd(void) { xs ^= xs << 13; xs ^= xs >> 17; xs ^= xs << 5; return xs; }

typedef struct { const uint8_t* ref; uint32_t w, h; } codec_state;

/* stand-in for encode(q) -> decode: coarser quantisation for lower q */
static int codec(void* user, uint32_t q, uint8_t* out_rgb, size_t* out_size) {
    codec_state* s = (codec_state*)user;
    const int step = 1 + (int)(100 - q) / 4;
    const size_t n = (size_t)s->w * s->h * 3;
    for (size_t i = 0; i < n; ++i) {
        int v = (s->ref[i] / step) * step + step / 2;
        out_rgb[i] = (uint8_t)(v > 255 ? 255 : v);
    }
    *out_size = 1000 + 10 * q;
    return 0;
}

int main(int argc, char** argv) {
    const uint32_t w = argc > 1 ? (uint32_t)atoi(argv[1]) : 320, h = argc > 2 ? (uint32_t)atoi(argv[2]) : 200;
    const size_t n = (size_t)w * h * 3;
    uint8_t* ref = (uint8_t*)malloc(n);
    uint8_t* dist = (uint8_t*)malloc(n);
    if (!ref || !dist) return 2;
    for (size_t i = 0; i < n; ++i) {  /* smooth-ish field + noise */
        const size_t px = i / 3, x = px % w, y = px / w;
        ref[i] = (uint8_t)((x * 3 + y * 2 + (i % 3) * 40 + (rnd() & 15)) & 255);
    }
    for (size_t i = 0; i < n; ++i) {
        int v = ref[i] + (int)(rnd() % 9) - 4;
        dist[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
    ssimu2_ctx* ctx = NULL;
    int rc = ssimu2_ctx_create(0, NULL, &ctx);
    if (rc != SSIMU2_OK) {
        fprintf(stderr, "ctx_create failed rc=%d: %s\n", rc, ssimu2_last_error(NULL));
        return rc == SSIMU2_ERR_NO_DEVICE ? 77 : 1;
    }
    double pair = 0, cached = 0;
    if ((rc = ssimu2_score_rgb8(ctx, ref, dist, w, h, 3, &pair))) goto fail;
    if ((rc = ssimu2_set_reference(ctx, ref, w, h))) goto fail;
    if ((rc = ssimu2_score_against_reference(ctx, dist, &cached))) goto fail;
    if (ssimu2_score_rgb8(ctx, ref, dist, w, h, 4, &pair) != SSIMU2_ERR_UNSUPPORTED) { rc = 99; goto fail; }
    {   /* the frame as libavif hands it over: RGBA, rows 8 bytes longer than their pixels */
        const uint32_t pitch = w * 4 + 8;
        uint8_t* rgba = (uint8_t*)malloc((size_t)pitch * h);
        double strided = 0;
        if (!rgba) return 2;
        for (size_t i = 0; i < (size_t)pitch * h; ++i) rgba[i] = (uint8_t)rnd();
        for (uint32_t y = 0; y < h; ++y)
            for (uint32_t x = 0; x < w; ++x)
                memcpy(rgba + (size_t)y * pitch + (size_t)x * 4, dist + ((size_t)y * w + x) * 3, 3);
        rc = ssimu2_score_against_reference_strided(ctx, rgba, pitch, 4, &strided);
        free(rgba);
        if (rc) goto fail;
        if (strided != cached) { rc = 98; goto fail; }
    }

    {   /* round 5: the device record the context was created with, and a page-locked frame from the library (what a host
           points avifRGBImage.pixels at, io.zig:452-482) scoring like the malloc'ed one */
        ssimu2_device_info di, dq;
        double pinned_score = 0;
        void* pin = NULL;
        memset(&di, 0, sizeof di);
        di.struct_size = (uint32_t)sizeof di;
        dq = di;
        if ((rc = ssimu2_ctx_device_info(ctx, &di)) || (rc = ssimu2_query_device(0, &dq))) goto fail;
        if (strncmp(di.arch, "gfx950", 6) || di.lds_bytes_per_cu < 160u * 1024u || di.wavefront_size != 64 || !di.usable ||
            strcmp(di.pci_bus_id, dq.pci_bus_id) || di.compute_units != dq.compute_units) { rc = 97; goto fail; }
        di.struct_size = 8;   /* a caller compiled against another header */
        if (ssimu2_ctx_device_info(ctx, &di) != SSIMU2_ERR_INVALID_ARG) { rc = 96; goto fail; }
        if (ssimu2_host_alloc(ctx, 0, &pin) != SSIMU2_ERR_INVALID_ARG) { rc = 95; goto fail; }
        if ((rc = ssimu2_host_alloc(ctx, n, &pin))) goto fail;
        memcpy(pin, dist, n);
        rc = ssimu2_score_against_reference(ctx, (const uint8_t*)pin, &pinned_score);
        if (!rc) rc = ssimu2_host_free(ctx, pin);
        if (rc) goto fail;
        if (pinned_score != cached || ssimu2_host_free(ctx, NULL) != SSIMU2_OK) { rc = 94; goto fail; }
        fprintf(stderr, "device %s %s pci %s numa %d cus %u lds %u\n", dq.name, dq.arch, dq.pci_bus_id, dq.numa_node, dq.compute_units,
                dq.lds_bytes_per_cu);
    }

    oavif_tq_options o;
    oavif_tq_default_options(&o);
    oavif_tq_result res;
    codec_state cs = {ref, w, h};
    size_t last = 0;
    if ((rc = oavif_tq_search_hip(&o, ctx, ref, w, h, codec, &cs, &res, &last))) goto fail;
    printf("score %.12f %.12f %d passes %u q %u last_size %zu version %s\n", pair, cached,
           pair == cached, res.num_pass, res.q, last, ssimu2_version());
    ssimu2_ctx_destroy(ctx);
    free(ref);
    free(dist);
    return 0;
fail:
    fprintf(stderr, "failed rc=%d: %s\n", rc, ssimu2_last_error(ctx));
    ssimu2_ctx_destroy(ctx);
    return 1;
}
