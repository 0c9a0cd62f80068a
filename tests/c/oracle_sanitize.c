/* Sanitizer harness for the CPU checker (oracle/ssimu2_oracle.c), built by
 * tests/test_sanitizers.py with gcc -fsanitize=address,undefined -fno-sanitize-recover: every
 * blur mode over tiny, odd and ragged frames (heap buffers of exactly w*h*3 bytes, so a read
 * one past the frame is caught), the strided copy loop, the stage functions.  Prints one
 * checksum line; exit code 0 = no report. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int or_compute_ssimu2(const uint8_t* ref, const uint8_t* dist, uint32_t w, uint32_t h, uint32_t channels,
                      int blur_mode, double* out_score, double* avg_out, int* nscales_out);
int or_compute_ssimu2_variant(const uint8_t* ref, const uint8_t* dist, uint32_t w, uint32_t h, int blur_mode, unsigned variant,
                              double* out_score, double* avg_out, int* nscales_out);
void or_copy_rgb_pixels(const uint8_t* src, size_t row_bytes, int src_channels, int w, int h, uint8_t* dst);
void or_blur_plane(const float* in, uint32_t w, uint32_t h, int mode, float* out);
void or_downsample2(const float* in, size_t w, size_t h, float* out);
void or_linear_to_xyb(const float* lin, size_t n, float* xyb);
void or_srgb_lut(float* lut256);

static uint32_t xs = 2463534242u;
static uint32_t rnd(void) { xs ^= xs << 13; xs ^= xs >> 17; xs ^= xs << 5; return xs; }

int main(void) {
    static const int sizes[][2] = {{1, 1}, {1, 9}, {9, 1}, {2, 2}, {7, 7}, {8, 8}, {9, 17}, {16, 15}, {17, 33},
                                   {63, 65}, {64, 64}, {127, 129}, {121, 9}, {300, 5}};
    static const int modes[] = {0, 1, 2, 3, 4};
    double checksum = 0.0;
    for (size_t k = 0; k < sizeof sizes / sizeof sizes[0]; ++k) {
        const uint32_t w = (uint32_t)sizes[k][0], h = (uint32_t)sizes[k][1];
        const size_t n = (size_t)w * h * 3;
        uint8_t* ref = (uint8_t*)malloc(n);
        uint8_t* dist = (uint8_t*)malloc(n);
        for (size_t i = 0; i < n; ++i) {
            ref[i] = (uint8_t)(rnd() & 255);
            const int v = ref[i] + (int)(rnd() % 31) - 15;
            dist[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
        for (size_t m = 0; m < sizeof modes / sizeof modes[0]; ++m) {
            double score = 0.0, avg[108];
            int ns = 0;
            const int rc = or_compute_ssimu2(ref, dist, w, h, 3, modes[m], &score, avg, &ns);
            if (rc != 0 || !isfinite(score) || ns < 0 || ns > 6 || (ns == 0) != (w < 8 || h < 8)) {
                fprintf(stderr, "%ux%u mode %d: rc=%d score=%g ns=%d\n", w, h, modes[m], rc, score, ns);
                return 1;
            }
            double same = 0.0;
            if (or_compute_ssimu2(ref, ref, w, h, 3, modes[m], &same, NULL, NULL) != 0 || same != 100.0) {
                fprintf(stderr, "%ux%u mode %d: identical frames score %g\n", w, h, modes[m], same);
                return 1;
            }
            checksum += score;
        }
        /* decoded-frame copy loop: RGBA rows with padding, buffer of exactly the bytes it may read */
        {
            const size_t pitch = (size_t)w * 4 + 5;
            const size_t bytes = pitch * (h - 1) + (size_t)w * 4;
            uint8_t* rgba = (uint8_t*)malloc(bytes);
            uint8_t* out = (uint8_t*)malloc(n);
            for (size_t i = 0; i < bytes; ++i) rgba[i] = (uint8_t)(rnd() & 255);
            or_copy_rgb_pixels(rgba, pitch, 4, (int)w, (int)h, out);
            for (uint32_t y = 0; y < h; ++y)
                for (uint32_t x = 0; x < w; ++x)
                    for (int c = 0; c < 3; ++c)
                        if (out[((size_t)y * w + x) * 3 + c] != rgba[y * pitch + (size_t)x * 4 + c]) return 2;
            free(rgba);
            free(out);
        }
        /* stage functions on exact-size planes */
        {
            const size_t px = (size_t)w * h, wo = (w + 1) / 2, ho = (h + 1) / 2;
            float* a = (float*)malloc(sizeof(float) * 3 * px);
            float* b = (float*)malloc(sizeof(float) * 3 * px);
            float* d = (float*)malloc(sizeof(float) * 3 * wo * ho);
            float lut[256];
            or_srgb_lut(lut);
            for (size_t i = 0; i < 3 * px; ++i) a[i] = lut[rnd() & 255];
            or_linear_to_xyb(a, px, b);
            or_downsample2(a, w, h, d);
            for (size_t m = 0; m < sizeof modes / sizeof modes[0]; ++m) {
                if (modes[m] == 4) continue;  /* a whole-score mode, not a plane blur */
                or_blur_plane(a, w, h, modes[m], b);
                for (size_t i = 0; i < px; ++i)
                    if (!isfinite(b[i])) return 3;
            }
            checksum += d[0] + b[px - 1];
            free(a);
            free(b);
            free(d);
        }
        free(ref);
        free(dist);
    }
    /* the pin kit's stage variants (round 5): every bit alone and a few combinations on sizes that stress their index
       arithmetic -- mirrored / clamped edges on planes narrower than the kernel radius, floor-sized odd dimensions
       down to one pixel, the size test after downsampling on frames that lose a scale, XYB-domain downsampling */
    {
        static const int vs[][2] = {{17, 9}, {9, 33}, {8, 8}, {203, 101}, {31, 64}, {1, 40}, {40, 1}, {3, 3}};
        static const unsigned blur_bits[] = {0x1, 0x2, 0x4, 0x8, 0x5, 0xA, 0x9, 0x6};
        for (size_t k = 0; k < sizeof vs / sizeof vs[0]; ++k) {
            const uint32_t w = (uint32_t)vs[k][0], h = (uint32_t)vs[k][1];
            const size_t n = (size_t)w * h * 3;
            uint8_t* a = (uint8_t*)malloc(n);
            uint8_t* b = (uint8_t*)malloc(n);
            for (size_t i = 0; i < n; ++i) {
                a[i] = (uint8_t)(rnd() >> 3);
                b[i] = (uint8_t)(a[i] ^ ((rnd() & 31) == 0 ? 9 : 0));
            }
            double score = 0, avg[108];
            int ns = 0;
            for (unsigned bit = 0x10; bit <= 0x200; bit <<= 1)
                for (int mode = 0; mode <= 1; ++mode) /* stage bits on the recursion and on the FIR */
                    if (or_compute_ssimu2_variant(a, b, w, h, mode, bit, &score, avg, &ns) != 0 || !(score <= 100.0)) {
                        fprintf(stderr, "stage variant %#x mode %d failed at %ux%u\n", bit, mode, w, h);
                        return 4;
                    }
            for (size_t j = 0; j < sizeof blur_bits / sizeof blur_bits[0]; ++j)
                if (or_compute_ssimu2_variant(a, b, w, h, 4, blur_bits[j] | (j & 1 ? 0x30u : 0u), &score, avg, &ns) != 0 || !(score <= 100.0)) {
                    fprintf(stderr, "blur variant %#x failed at %ux%u\n", blur_bits[j], w, h);
                    return 4;
                }
            if (or_compute_ssimu2_variant(a, b, w, h, 1, 0x3FF, &score, avg, &ns) == 0 ||  /* clamp + mirror at once */
                or_compute_ssimu2_variant(a, b, w, h, 0, 0x1, &score, avg, &ns) == 0 ||    /* a blur bit on the recursion */
                or_compute_ssimu2_variant(a, b, w, h, 1, 0x400, &score, avg, &ns) == 0) {  /* an unknown bit */
                fprintf(stderr, "a contradictory variant was accepted at %ux%u\n", w, h);
                return 4;
            }
            checksum += score;
            free(a);
            free(b);
        }
    }
    printf("oracle_sanitize ok: checksum %.6f\n", checksum);
    return 0;
}
