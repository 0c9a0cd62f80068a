/*
 * oavif_host.c -- the drop-in, end to end, as a compiled host: what oavif's main.zig does
 * around the boundary (/root/reference/src/main.zig:37-117), in C over the two public headers.
 *
 *     oavif_host [options] <in.png|in.pam> <out.avif>
 *
 * The reference's host language is Zig and the image has no Zig toolchain, so the binding a
 * maintainer adds (oavif_amd/zig/fssimu2.zig) cannot be compiled here.
 * This file is the same host side in the language that CAN be compiled here, with nothing of
 * Python in the loop:
 *
 *   main.zig:73-91    load the image, build the scorer's RGB8 reference (Image.toRGB8)
 *   main.zig:93-100   -q: encode once
 *   main.zig:102-116  findTargetQuality -> oavif_tq_find_target_quality (include/oavif_tq.h), whose
 *                     pass (tq.zig:21-38) is the callback below: io.encodeAvifToBuffer and
 *                     decodeAvifCommon against libavif's C API, call for call (io.zig:452-482,
 *                     544-636), then the score of tq.zig:37 on the GPU through
 *                     ssimu2_score_against_reference_strided (libavif's RGB(A) rows as they are:
 *                     the copy loop of io.zig:654-663 never runs)
 *   io.zig:550-617    the source is rescaled to the encoder's depth (oavif_prescale_*) and converted to
 *                     YUV444 ONCE, not once per pass (SURVEY.md 8f rank 4 and one step further)
 *
 * libavif: the image has a libavif 1.4.1 shared library (inside Pillow's wheel) and no header.
 * The leading members of the four public structs used here are declared below from the public
 * API of libavif 1.x, the library is opened with dlopen (OAVIF_LIBAVIF names it), and the
 * documented defaults of avifEncoderCreate / avifDecoderCreate / avifImageCreate /
 * avifRGBImageSetDefaults are read back through these declarations before anything is encoded:
 * a library with another layout is refused.  PNG input goes through oavif_png_decode (the C ABI's
 * loader with io.loadPNG's output rules), PAM through the reader below (io.zig:309-406).
 *
 * stderr carries the reference's lines in the reference's formats (main.zig:78-84,95,98,102,106,116:
 * scripts/measure.py parses "N passes").  Exit code 0, or 1 with "error: <ZigErrorName>".
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <errno.h>
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include "oavif_tq.h"
#include "ssimu2_hip.h"

#define VERSION "oavif_amd-0.1 (C host)"

/* ---- libavif 1.x: the members this host touches (public API; checked at run time) --------------------- */
typedef struct { uint8_t* data; size_t size; } avifRWData;
typedef struct {
    uint32_t width, height, depth;
    int yuvFormat, yuvRange, yuvChromaSamplePosition;
    uint8_t* yuvPlanes[3];
    uint32_t yuvRowBytes[3];
    int imageOwnsYUVPlanes;
    uint8_t* alphaPlane;
    uint32_t alphaRowBytes;
    int imageOwnsAlphaPlane, alphaPremultiplied;
    avifRWData icc;
    uint16_t colorPrimaries, transferCharacteristics, matrixCoefficients;
} avifImageHead; /* avifImage continues (clli, transforms, exif, xmp, ...): never touched */
typedef struct {
    uint32_t width, height, depth;
    int format, chromaUpsampling, chromaDownsampling, avoidLibYUV, ignoreAlpha, alphaPremultiplied, isFloat;
    int maxThreads;
    uint8_t* pixels;
    uint32_t rowBytes;
} avifRGBImage;
typedef struct {
    int codecChoice, maxThreads, speed, keyframeInterval;
    uint64_t timescale;
    int repetitionCount;
    uint32_t extraLayerCount;
    int quality, qualityAlpha, minQuantizer, maxQuantizer, minQuantizerAlpha, maxQuantizerAlpha;
    int tileRowsLog2, tileColsLog2, autoTiling;
    int scalingMode[4];
} avifEncoderHead; /* avifEncoder continues (ioStats, diag, data, ...) */
typedef struct {
    int codecChoice, maxThreads, requestedSource, allowProgressive, allowIncremental, ignoreExif, ignoreXMP;
    uint32_t imageSizeLimit, imageDimensionLimit, imageCountLimit, strictFlags;
    avifImageHead* image;
} avifDecoderHead; /* avifDecoder continues */
_Static_assert(offsetof(avifImageHead, alphaPlane) == 64 && offsetof(avifImageHead, icc) == 88 &&
               offsetof(avifImageHead, colorPrimaries) == 104, "avifImage head");
_Static_assert(offsetof(avifRGBImage, pixels) == 48 && offsetof(avifRGBImage, rowBytes) == 56, "avifRGBImage");
_Static_assert(offsetof(avifEncoderHead, quality) == 32 && offsetof(avifEncoderHead, autoTiling) == 64, "avifEncoder head");
_Static_assert(offsetof(avifDecoderHead, image) == 48, "avifDecoder head");
enum { AVIF_OK = 0, AVIF_YUV444 = 1, AVIF_RGB = 0, AVIF_RGBA = 1, AVIF_ADD_IMAGE_FLAG_SINGLE = 2 };

static struct {
    const char* (*Version)(void);
    const char* (*ResultToString)(int);
    avifImageHead* (*ImageCreate)(uint32_t, uint32_t, uint32_t, int);
    void (*ImageDestroy)(avifImageHead*);
    int (*ImageSetProfileICC)(avifImageHead*, const uint8_t*, size_t);
    void (*RGBImageSetDefaults)(avifRGBImage*, const avifImageHead*);
    int (*RGBImageAllocatePixels)(avifRGBImage*);
    void (*RGBImageFreePixels)(avifRGBImage*);
    int (*ImageRGBToYUV)(avifImageHead*, const avifRGBImage*);
    int (*ImageYUVToRGB)(const avifImageHead*, avifRGBImage*);
    avifEncoderHead* (*EncoderCreate)(void);
    void (*EncoderDestroy)(avifEncoderHead*);
    int (*EncoderSetCodecSpecificOption)(avifEncoderHead*, const char*, const char*);
    int (*EncoderAddImage)(avifEncoderHead*, const avifImageHead*, uint64_t, uint32_t);
    int (*EncoderFinish)(avifEncoderHead*, avifRWData*);
    void (*RWDataFree)(avifRWData*);
    avifDecoderHead* (*DecoderCreate)(void);
    void (*DecoderDestroy)(avifDecoderHead*);
    int (*DecoderSetIOMemory)(avifDecoderHead*, const uint8_t*, size_t);
    int (*DecoderParse)(avifDecoderHead*);
    int (*DecoderNextImage)(avifDecoderHead*);
} av;

static const char* g_err = NULL; /* the reference's Zig error name of the failure (the first one: probes may run on threads) */
static char g_detail[320];
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static int fail(const char* name, const char* detail) {
    pthread_mutex_lock(&g_lock);
    if (!g_err) {
        g_err = name;
        snprintf(g_detail, sizeof g_detail, "%s", detail ? detail : "");
    }
    pthread_mutex_unlock(&g_lock);
    return -1;
}

static int load_libavif(void) {
    const char* path = getenv("OAVIF_LIBAVIF");
    void* h = dlopen(path && *path ? path : "libavif.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("libavif.so.16", RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail("LibavifUnavailable", dlerror());
#define SYM(field, name)                                                    \
    do {                                                                    \
        *(void**)(&av.field) = dlsym(h, name);                              \
        if (!av.field) return fail("LibavifUnavailable", "missing " name);  \
    } while (0)
    SYM(Version, "avifVersion"); SYM(ResultToString, "avifResultToString");
    SYM(ImageCreate, "avifImageCreate"); SYM(ImageDestroy, "avifImageDestroy");
    SYM(ImageSetProfileICC, "avifImageSetProfileICC"); SYM(RGBImageSetDefaults, "avifRGBImageSetDefaults");
    SYM(RGBImageAllocatePixels, "avifRGBImageAllocatePixels"); SYM(RGBImageFreePixels, "avifRGBImageFreePixels");
    SYM(ImageRGBToYUV, "avifImageRGBToYUV"); SYM(ImageYUVToRGB, "avifImageYUVToRGB");
    SYM(EncoderCreate, "avifEncoderCreate"); SYM(EncoderDestroy, "avifEncoderDestroy");
    SYM(EncoderSetCodecSpecificOption, "avifEncoderSetCodecSpecificOption");
    SYM(EncoderAddImage, "avifEncoderAddImage"); SYM(EncoderFinish, "avifEncoderFinish");
    SYM(RWDataFree, "avifRWDataFree"); SYM(DecoderCreate, "avifDecoderCreate");
    SYM(DecoderDestroy, "avifDecoderDestroy"); SYM(DecoderSetIOMemory, "avifDecoderSetIOMemory");
    SYM(DecoderParse, "avifDecoderParse"); SYM(DecoderNextImage, "avifDecoderNextImage");
#undef SYM
    /* layout guard: the documented defaults, read back through the declarations above */
    if (strncmp(av.Version(), "1.", 2) != 0) return fail("LibavifUnavailable", "not libavif 1.x");
    avifEncoderHead* e = av.EncoderCreate();
    if (!e) return fail("OutOfMemory", NULL);
    const int enc_ok = e->codecChoice == 0 && e->maxThreads == 1 && e->speed == -1 && e->keyframeInterval == 0 &&
                       e->timescale == 1 && e->repetitionCount == -1 && e->extraLayerCount == 0 &&
                       e->minQuantizer == 0 && e->maxQuantizer == 63 && e->minQuantizerAlpha == 0 &&
                       e->maxQuantizerAlpha == 63 && e->tileRowsLog2 == 0 && e->tileColsLog2 == 0 &&
                       e->autoTiling == 0 && e->scalingMode[0] == 1 && e->scalingMode[1] == 1 &&
                       e->scalingMode[2] == 1 && e->scalingMode[3] == 1;
    av.EncoderDestroy(e);
    avifDecoderHead* d = av.DecoderCreate();
    if (!d) return fail("OutOfMemory", NULL);
    const int dec_ok = d->maxThreads == 1 && d->imageSizeLimit == 16384u * 16384u && d->imageDimensionLimit == 32768 &&
                       d->imageCountLimit == 12 * 3600 * 60 && d->image == NULL;
    av.DecoderDestroy(d);
    avifImageHead* im = av.ImageCreate(24, 16, 10, AVIF_YUV444);
    if (!im) return fail("OutOfMemory", NULL);
    avifRGBImage rgb;
    memset(&rgb, 0xaa, sizeof rgb);
    av.RGBImageSetDefaults(&rgb, im);
    const int img_ok = im->width == 24 && im->height == 16 && im->depth == 10 && im->yuvFormat == AVIF_YUV444 &&
                       im->yuvRange == 1 && im->alphaPlane == NULL && im->icc.data == NULL && im->icc.size == 0 &&
                       im->colorPrimaries == 2 && im->transferCharacteristics == 2 && im->matrixCoefficients == 2;
    const int rgb_ok = rgb.width == 24 && rgb.height == 16 && rgb.depth == 10 && rgb.format == AVIF_RGBA &&
                       rgb.maxThreads == 1 && rgb.pixels == NULL && rgb.rowBytes == 0;
    av.ImageDestroy(im);
    if (!(enc_ok && dec_ok && img_ok && rgb_ok))
        return fail("LibavifUnavailable", "struct layout of the loaded libavif differs from the 1.x one declared here");
    return 0;
}

/* ---- options (parse_args.zig:48-63 defaults, :76-122 flags and ranges) ----------------------------------- */
typedef struct {
    int quality_alpha, speed, max_threads, tile_rows_log2, tile_cols_log2, auto_tiling;
    double score_tgt;
    int tenbit;
    const char* tune;
    double tolerance;
    int max_pass, quality /* -1 = search */, color_primaries, transfer_characteristics, matrix_coefficients;
} Options;
#define OPTIONS_DEFAULT {0, 9, 1, 0, 0, 1, 80.0, 1, "iq", 2.0, 6, -1, 2, 2, 2} /* parse_args.zig:48-63 */

static int int_arg(int* i, int argc, char** argv, long lo, long hi, const char* name, int* out) {
    if (*i >= argc || argv[*i][0] == '-') { /* parse_args.zig:126: a value starting with '-' counts as missing */
        fprintf(stderr, "Error: Missing %s value\n", name);
        return fail("MissingOptionValue", NULL);
    }
    char* end;
    errno = 0;
    const long v = strtol(argv[*i], &end, 10);
    if (*end || errno) return fail("InvalidCharacter", NULL);
    if (v < lo || v > hi) {
        fprintf(stderr, "Error: %s must be between %ld and %ld\n", name, lo, hi);
        return fail("InvalidOptionValue", NULL);
    }
    ++*i;
    *out = (int)v;
    return 0;
}
static int float_arg(int* i, int argc, char** argv, double lo, double hi, const char* name, double* out) {
    if (*i >= argc || argv[*i][0] == '-') {
        fprintf(stderr, "Error: Missing %s value\n", name);
        return fail("MissingOptionValue", NULL);
    }
    char* end;
    const double v = strtod(argv[*i], &end);
    if (*end || end == argv[*i]) return fail("InvalidCharacter", NULL);
    if (v < lo || v > hi) {
        fprintf(stderr, "Error: %s must be between %g and %g\n", name, lo, hi);
        return fail("InvalidOptionValue", NULL);
    }
    ++*i;
    *out = v;
    return 0;
}
static int bool_arg(int* i, int argc, char** argv, const char* name, int* out) {
    if (*i >= argc || argv[*i][0] == '-') {
        fprintf(stderr, "Error: Missing %s value\n", name);
        return fail("MissingOptionValue", NULL);
    }
    if (strcmp(argv[*i], "0") != 0 && strcmp(argv[*i], "1") != 0) {
        fprintf(stderr, "Error: %s must be 0 or 1\n", name);
        return fail("InvalidOptionValue", NULL);
    }
    *out = argv[(*i)++][0] == '1';
    return 0;
}

static int parse_args(Options* o, int argc, char** argv, const char** in, const char** out) {
    for (int i = 1; i < argc;) {
        const char* a = argv[i++];
#define IS(s) (strcmp(a, s) == 0)
        int rc = 0;
        if (IS("-s") || IS("--speed")) rc = int_arg(&i, argc, argv, 0, 10, "--speed", &o->speed);
        else if (IS("-t") || IS("--score-tgt")) rc = float_arg(&i, argc, argv, 30, 100, "--score-tgt", &o->score_tgt);
        else if (IS("--quality-alpha")) rc = int_arg(&i, argc, argv, 0, 99, "--quality-alpha", &o->quality_alpha);
        else if (IS("--max-threads")) rc = int_arg(&i, argc, argv, 1, 255, "--max-threads", &o->max_threads);
        else if (IS("--tile-rows-log2")) rc = int_arg(&i, argc, argv, 0, 6, "--tile-rows-log2", &o->tile_rows_log2);
        else if (IS("--tile-cols-log2")) rc = int_arg(&i, argc, argv, 0, 6, "--tile-cols-log2", &o->tile_cols_log2);
        else if (IS("--auto-tiling")) rc = bool_arg(&i, argc, argv, "--auto-tiling", &o->auto_tiling);
        else if (IS("--tune")) {
            if (i >= argc || argv[i][0] == '-') { fprintf(stderr, "Error: Missing --tune value\n"); return fail("MissingOptionValue", NULL); }
            const char* t = argv[i++];
            if (strcmp(t, "ssim") && strcmp(t, "iq") && strcmp(t, "ssimulacra2")) {
                fprintf(stderr, "Error: --tune must be one of: ssim, iq, ssimulacra2\n");
                return fail("InvalidTuneMode", NULL);
            }
            o->tune = t;
        } else if (IS("--tenbit")) rc = bool_arg(&i, argc, argv, "--tenbit", &o->tenbit);
        else if (IS("--tolerance")) rc = float_arg(&i, argc, argv, 1, 100, "--tolerance", &o->tolerance);
        else if (IS("--max-pass")) rc = int_arg(&i, argc, argv, 1, 12, "--max-pass", &o->max_pass);
        else if (IS("-q") || IS("--quality")) rc = int_arg(&i, argc, argv, 0, 100, "--quality", &o->quality);
        else if (IS("--color-primaries")) rc = int_arg(&i, argc, argv, 1, 22, "--color-primaries", &o->color_primaries);
        else if (IS("--transfer-characteristics")) rc = int_arg(&i, argc, argv, 1, 18, "--transfer-characteristics", &o->transfer_characteristics);
        else if (IS("--matrix-coefficients")) rc = int_arg(&i, argc, argv, 0, 14, "--matrix-coefficients", &o->matrix_coefficients);
        else if (!*in) *in = a;
        else if (!*out) *out = a;
        else { fprintf(stderr, "Error: Unexpected argument: %s\n", a); return fail("UnexpectedArgument", NULL); }
#undef IS
        if (rc) return rc;
    }
    return 0;
}

/* ---- image load: io.zig's Image (io.zig:42-55) ------------------------------------------------------------ */
typedef struct {
    uint32_t w, h, channels; /* 3 or 4 after load (gray is expanded: see load_pam) */
    int hbd;                 /* data is u16, full 16-bit range */
    uint8_t* data;
    uint8_t* icc;
    size_t icc_len;
} Image;

static uint8_t* read_file(const char* path, size_t* len) {
    FILE* f = fopen(path, "rb");
    if (!f) { fail("FileNotFound", path); return NULL; }
    /* Read until end of file into a growing buffer: a FIFO or another non-seekable input named x.png has no size to
       ask for (ftell gives -1 there; ADVICE r04: that became malloc(1) + fread of SIZE_MAX bytes). */
    size_t cap = 1u << 16, n = 0;
    struct stat st;
    if (fstat(fileno(f), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) cap = (size_t)st.st_size + 1;
    uint8_t* b = (uint8_t*)malloc(cap);
    while (b) {
        n += fread(b + n, 1, cap - n, f);
        if (n < cap) break; /* short read: end of file or an error */
        if (cap > SIZE_MAX / 2) { free(b); b = NULL; break; }
        uint8_t* g = (uint8_t*)realloc(b, cap * 2);
        if (!g) { free(b); b = NULL; break; }
        b = g;
        cap *= 2;
    }
    if (!b || ferror(f)) {
        fail(b ? "ReadFailed" : "OutOfMemory", path);
        fclose(f);
        free(b);
        return NULL;
    }
    fclose(f);
    *len = n;
    return b;
}

/* io.loadPAM (io.zig:309-406), the acceptance rules as oavif_amd/pam.py lists them: "P7"; the header ends at the
   first "ENDHDR\n", else at the first blank line; lines split on CR / LF, '#' comments, keys matched as prefixes,
   the first token after the key is the value; WIDTH, HEIGHT, DEPTH, MAXVAL all non-zero, MAXVAL 255, DEPTH 1..4;
   TUPLTYPE GRAYSCALE / GRAYSCALE_ALPHA / RGB / RGB_ALPHA must agree with DEPTH, BLACKANDWHITE is refused, anything
   else leaves DEPTH to decide; the raster must hold width*height*depth bytes.  Zig's error names. */
static const uint8_t* find_bytes(const uint8_t* b, size_t n, const char* pat) {
    const size_t m = strlen(pat);
    for (size_t i = 0; i + m <= n; ++i)
        if (memcmp(b + i, pat, m) == 0) return b + i;
    return NULL;
}
static int pam_value(const char* rest, char* tok, size_t cap) { /* first whitespace-separated token; 0 = none */
    while (*rest == ' ' || *rest == '\t') ++rest;
    size_t k = 0;
    while (*rest && *rest != ' ' && *rest != '\t' && k + 1 < cap) tok[k++] = *rest++;
    tok[k] = 0;
    return k > 0;
}
static int pam_usize(const char* tok, size_t* out) { /* std.fmt.parseInt(usize, tok, 10): digits, '_', a leading '+' */
    size_t v = 0;
    int digits = 0;
    if (*tok == '+') ++tok;
    for (; *tok; ++tok) {
        if (*tok == '_') continue;
        if (*tok < '0' || *tok > '9') return fail("InvalidCharacter", NULL);
        if (v > (SIZE_MAX - 9) / 10) return fail("Overflow", NULL);
        v = v * 10 + (size_t)(*tok - '0');
        digits = 1;
    }
    if (!digits) return fail("InvalidCharacter", NULL);
    *out = v;
    return 0;
}
static int load_pam(const uint8_t* b, size_t n, Image* im) {
    if (n < 3 || b[0] != 'P' || b[1] != '7') return fail("NotAPamFile", NULL);
    size_t header_end;
    const uint8_t* e = find_bytes(b, n, "ENDHDR\n");
    if (e) header_end = (size_t)(e - b) + 7;
    else if ((e = find_bytes(b, n, "\n\n")) != NULL) header_end = (size_t)(e - b) + 2;
    else return fail("HeaderNotFound", NULL);
    size_t w = 0, h = 0, depth = 0, maxval = 0;
    char tuple[64] = "UNSPECIFIED", tok[64];
    for (size_t p = 0; p < header_end;) {
        size_t q = p;
        while (q < header_end && b[q] != '\n' && b[q] != '\r') ++q;
        char line[160];
        const size_t L = q - p < sizeof line - 1 ? q - p : sizeof line - 1;
        memcpy(line, b + p, L);
        line[L] = 0;
        p = q + 1;
        if (!*line || *line == '#') continue;
        if (!strncmp(line, "WIDTH", 5)) { if (pam_value(line + 5, tok, sizeof tok) && pam_usize(tok, &w)) return -1; }
        else if (!strncmp(line, "HEIGHT", 6)) { if (pam_value(line + 6, tok, sizeof tok) && pam_usize(tok, &h)) return -1; }
        else if (!strncmp(line, "DEPTH", 5)) { if (pam_value(line + 5, tok, sizeof tok) && pam_usize(tok, &depth)) return -1; }
        else if (!strncmp(line, "MAXVAL", 6)) { if (pam_value(line + 6, tok, sizeof tok) && pam_usize(tok, &maxval)) return -1; }
        else if (!strncmp(line, "TUPLTYPE", 8)) { if (pam_value(line + 8, tok, sizeof tok)) snprintf(tuple, sizeof tuple, "%s", tok); }
        else if (!strcmp(line, "ENDHDR")) break;
    }
    if (!w || !h || !depth || !maxval) return fail("InvalidPamDimensions", NULL);
    if (maxval != 255) return fail("UnsupportedPamMaxVal", NULL);
    if (depth > 4) return fail("UnsupportedPamDepth", NULL);
    static const struct { const char* name; size_t ch; } kinds[] = {
        {"GRAYSCALE", 1}, {"GRAYSCALE_ALPHA", 2}, {"RGB", 3}, {"RGB_ALPHA", 4}};
    for (size_t k = 0; k < 4; ++k)
        if (!strcasecmp(tuple, kinds[k].name) && depth != kinds[k].ch) return fail("PamTupleMismatch", NULL);
    if (!strcasecmp(tuple, "BLACKANDWHITE")) return fail("UnsupportedPamTuple", NULL);
    if (w > 65536 || h > 65536) return fail("InsufficientDataInFile", NULL); /* no such raster fits the file anyway */
    const size_t px = w * h;
    if (px * depth > n - header_end) return fail("InsufficientDataInFile", NULL);
    /* gray (+alpha) is expanded to RGB(A): the reference hands 1- and 2-channel data to libavif as if it
       were RGB (io.zig:564), a row-stride bug this host does not reproduce */
    const uint32_t ch = depth <= 2 ? (uint32_t)depth + 2 : (uint32_t)depth;
    uint8_t* d = (uint8_t*)malloc(px * ch);
    if (!d) return fail("OutOfMemory", NULL);
    const uint8_t* s = b + header_end;
    for (size_t i = 0; i < px; ++i)
        for (uint32_t c = 0; c < ch; ++c)
            d[i * ch + c] = depth >= 3 ? s[i * depth + c] : (c < 3 ? s[i * depth] : s[i * depth + 1]);
    im->w = (uint32_t)w; im->h = (uint32_t)h; im->channels = ch; im->hbd = 0; im->data = d;
    return 0;
}

static int load_png(const uint8_t* b, size_t n, Image* im) { /* io.loadPNG (io.zig:242-307) via the C ABI */
    oavif_png_info info;
    int rc = oavif_png_info_from_memory(b, n, &info);
    if (rc) return fail(rc == OAVIF_PNG_ERR_HEADER ? "GetHeaderFailed" : rc == OAVIF_PNG_ERR_SIZE ? "ImageSizeFailed" : "DecodeFailed", NULL);
    uint8_t* d = (uint8_t*)malloc(info.data_bytes ? info.data_bytes : 1);
    uint8_t* icc = info.icc_bytes ? (uint8_t*)malloc(info.icc_bytes) : NULL;
    if (!d || (info.icc_bytes && !icc)) { free(d); free(icc); return fail("OutOfMemory", NULL); }
    rc = oavif_png_decode(b, n, d, info.data_bytes, icc, info.icc_bytes);
    if (rc) { free(d); free(icc); return fail("DecodeFailed", NULL); }
    im->w = info.width; im->h = info.height; im->channels = info.channels; im->hbd = info.hbd;
    im->data = d; im->icc = icc; im->icc_len = info.icc_bytes;
    return 0;
}

static int load_image(const char* path, Image* im) {
    const char* dot = strrchr(path, '.');
    size_t n = 0;
    if (!dot || (strcasecmp(dot, ".png") && strcasecmp(dot, ".pam"))) return fail("UnsupportedImageFormat", "this host reads PNG and PAM");
    uint8_t* b = read_file(path, &n);
    if (!b) return -1;
    const int rc = strcasecmp(dot, ".png") == 0 ? load_png(b, n, im) : load_pam(b, n, im);
    free(b);
    return rc;
}

/* ---- the pass: io.encodeAvifToBuffer + decodeAvifCommon + the score (tq.zig:21-38) ------------------------- */
typedef struct {
    const Options* o;
    const Image* src;
    const void* scaled; /* the source at the encoder's depth, computed once */
    uint32_t out_depth;
    avifImageHead* image; /* ... and as the encoder takes it (YUV444), converted once: make_source_image() */
    ssimu2_ctx* scorer; /* made when the first score needs it (ensure_scorer) */
    int device, blur;
    const uint8_t* rgb8; /* e.rgb: the scorer's reference */
    /* EncBuffer (main.zig:11-19): the AVIF bytes of the LAST probe */
    uint8_t* buf;
    size_t buf_size;
    int buf_q;
    double encode_ms, decode_ms, score_ms;
    /* decodeAvifCommon's RGB buffer (io.zig:452-482) in page-locked memory of the scorer (ssimu2_host_alloc), used by
       every pass once the context exists: libavif converts straight into it and the upload is one DMA.  NULL = the
       context is not up yet (the first pass's CPU half runs during HIP start-up), or OAVIF_HOST_PINNED=0. */
    uint8_t* pinned;
} EncCtx;

#include <time.h>
static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

/* io.zig:550-623, hoisted out of the pass loop: the avifImage of the source -- created, tagged (CICP, ICC) and
   converted to YUV444 -- is made ONCE.  The reference rebuilds it on every pass (avifImageCreate +
   avifImageRGBToYUV over the whole frame, 15 % of a 4K encode at speed 9) although only `quality` changes between
   passes (io.zig:625).  Same planes, same bitstream. */
static int make_source_image(EncCtx* e) {
    const Options* o = e->o;
    const Image* s = e->src;
    int r;
    avifImageHead* image = av.ImageCreate(s->w, s->h, e->out_depth, AVIF_YUV444);
    if (!image) return fail("OutOfMemory", NULL);
    image->colorPrimaries = (uint16_t)o->color_primaries;
    image->transferCharacteristics = (uint16_t)o->transfer_characteristics;
    image->matrixCoefficients = (uint16_t)o->matrix_coefficients;
    if (s->icc && av.ImageSetProfileICC(image, s->icc, s->icc_len) != AVIF_OK) {
        av.ImageDestroy(image);
        return fail("SetICCProfileFailed", NULL);
    }
    avifRGBImage rgb;
    memset(&rgb, 0, sizeof rgb);
    av.RGBImageSetDefaults(&rgb, image);
    rgb.format = s->channels == 4 ? AVIF_RGBA : AVIF_RGB;
    rgb.pixels = (uint8_t*)(uintptr_t)e->scaled;
    rgb.rowBytes = s->w * s->channels * (e->out_depth > 8 ? 2u : 1u);
    rgb.depth = e->out_depth;
    if ((r = av.ImageRGBToYUV(image, &rgb)) != AVIF_OK) {
        av.ImageDestroy(image);
        return fail("ConvertFailed", av.ResultToString(r));
    }
    e->image = image;
    return 0;
}

static int encode_to_buffer(EncCtx* e, uint32_t q, uint8_t** out, size_t* out_size) { /* io.zig:619-635 */
    const Options* o = e->o;
    avifEncoderHead* enc = NULL;
    avifRWData output = {NULL, 0};
    int rc = -1, r;
    if (!e->image && make_source_image(e)) return -1;
    enc = av.EncoderCreate();
    if (!enc) { fail("OutOfMemory", NULL); goto done; }
    enc->qualityAlpha = o->quality_alpha; /* copyToEncoder (parse_args.zig:65-74) */
    enc->speed = o->speed;
    enc->maxThreads = o->max_threads;
    enc->tileRowsLog2 = o->tile_rows_log2;
    enc->tileColsLog2 = o->tile_cols_log2;
    enc->autoTiling = o->auto_tiling;
    if (av.EncoderSetCodecSpecificOption(enc, "tune", o->tune) != AVIF_OK) { fail("InvalidCodecOption", NULL); goto done; }
    enc->quality = (int)q; /* io.zig:625-626 */
    enc->qualityAlpha = o->quality_alpha;
    if ((r = av.EncoderAddImage(enc, e->image, 1, AVIF_ADD_IMAGE_FLAG_SINGLE)) != AVIF_OK) { fail("AddImageFailed", av.ResultToString(r)); goto done; }
    if ((r = av.EncoderFinish(enc, &output)) != AVIF_OK) { fail("FinishFailed", av.ResultToString(r)); goto done; }
    *out = (uint8_t*)malloc(output.size);
    if (!*out) { fail("OutOfMemory", NULL); goto done; }
    memcpy(*out, output.data, output.size);
    *out_size = output.size;
    rc = 0;
done:
    if (output.data) av.RWDataFree(&output);
    if (enc) av.EncoderDestroy(enc);
    return rc;
}

/* One pass (computeScoreAtQuality, tq.zig:21-38) in two halves: the CPU half (encode at q, decode to libavif's
   8-bit RGB(A) rows) touches nothing shared but the read-only source image, so the probes of a speculative wave
   run it on threads and the FIRST pass of a run runs it while the scorer is still starting up; the GPU half
   scores those rows on a given context. */
typedef struct { avifDecoderHead* dec; avifRGBImage rgb; int caller_pixels; } Decoded;
static void decoded_free(Decoded* d) {
    if (d->rgb.pixels && !d->caller_pixels) av.RGBImageFreePixels(&d->rgb);
    if (d->dec) av.DecoderDestroy(d->dec);
    memset(d, 0, sizeof *d);
}
static uint8_t* pinned_frame(ssimu2_ctx* scorer, const Image* src) { /* room for RGBA rows; NULL = use libavif's own */
    void* p = NULL;
    const char* v = getenv("OAVIF_HOST_PINNED");
    if (v && !strcmp(v, "0")) return NULL;
    if (ssimu2_host_alloc(scorer, (size_t)src->w * src->h * 4, &p) != SSIMU2_OK) return NULL; /* not fatal */
    return (uint8_t*)p;
}
static int pass_cpu(const EncCtx* e, uint32_t q, uint8_t** out_avif, size_t* out_size, Decoded* d, double times_ms[3],
                    uint8_t* pinned) {
    uint8_t* avif = NULL;
    size_t avif_size = 0;
    memset(d, 0, sizeof *d);
    double t0 = now_ms();
    if (encode_to_buffer((EncCtx*)(uintptr_t)e, q, &avif, &avif_size)) return -1; /* e->image exists: read-only use */
    double t1 = now_ms();
    /* decodeAvifCommon(avif, use_8bit = true) (io.zig:452-482) */
    int rc = -1, r;
    d->dec = av.DecoderCreate();
    if (!d->dec) { fail("OutOfMemory", NULL); goto done; }
    if (av.DecoderSetIOMemory(d->dec, avif, avif_size) != AVIF_OK) { fail("SetIOFailed", NULL); goto done; }
    if ((r = av.DecoderParse(d->dec)) != AVIF_OK) { fail("ParseFailed", av.ResultToString(r)); goto done; }
    if ((r = av.DecoderNextImage(d->dec)) != AVIF_OK) { fail("DecodeImageFailed", av.ResultToString(r)); goto done; }
    const avifImageHead* img = d->dec->image;
    if (!img || img->width != e->src->w || img->height != e->src->h) { fail("DecodeImageFailed", "unexpected frame size"); goto done; }
    av.RGBImageSetDefaults(&d->rgb, img);
    d->rgb.depth = 8;                                        /* io.zig:470-471 */
    d->rgb.format = img->alphaPlane ? AVIF_RGBA : AVIF_RGB;  /* io.zig:473 */
    if (pinned) { /* the caller's buffer in place of avifRGBImageAllocatePixels: tight rows, as libavif would lay them */
        d->rgb.pixels = pinned;
        d->rgb.rowBytes = img->width * (d->rgb.format == AVIF_RGBA ? 4u : 3u);
        d->caller_pixels = 1;
    } else if (av.RGBImageAllocatePixels(&d->rgb) != AVIF_OK) { fail("AllocatePixelsFailed", NULL); goto done; }
    if ((r = av.ImageYUVToRGB(img, &d->rgb)) != AVIF_OK) { fail("ConvertToRGBFailed", av.ResultToString(r)); goto done; }
    times_ms[0] = t1 - t0; times_ms[1] = now_ms() - t1;
    *out_avif = avif; *out_size = avif_size;
    avif = NULL;
    rc = 0;
done:
    if (rc) decoded_free(d);
    free(avif);
    return rc;
}
static int pass_score(ssimu2_ctx* scorer, Decoded* d, double* out_score, double times_ms[3]) {
    /* tq.zig:37, without the copy loop of io.zig:654-663: libavif's rows as they are */
    const double t = now_ms();
    const int r = ssimu2_score_against_reference_strided(scorer, d->rgb.pixels, d->rgb.rowBytes,
                                                         d->rgb.format == AVIF_RGBA ? 4 : 3, out_score);
    times_ms[2] = now_ms() - t;
    decoded_free(d);
    return r == SSIMU2_OK ? 0 : fail("ScorerFailed", ssimu2_last_error(scorer));
}

static void phase(const char* what);
static int ensure_scorer(EncCtx* e) { /* ssimu2_prefetch started this at process start; the first score waits for it */
    if (e->scorer) return 0;
    int rc = ssimu2_ctx_create(e->device, NULL, &e->scorer);
    if (rc != SSIMU2_OK) return fail(rc == SSIMU2_ERR_NO_DEVICE ? "NoDevice" : "ScorerFailed", ssimu2_last_error(NULL));
    phase("scorer context created");
    if ((rc = ssimu2_ctx_set_blur(e->scorer, e->blur)) || (rc = ssimu2_set_reference(e->scorer, e->rgb8, e->src->w, e->src->h)))
        return fail("ScorerFailed", ssimu2_last_error(e->scorer));
    phase("reference uploaded and cached");
    e->pinned = pinned_frame(e->scorer, e->src);
    return 0;
}

static int probe(void* user, uint32_t q, double* out_score) { /* the sequential search's pass */
    EncCtx* e = (EncCtx*)user;
    uint8_t* avif = NULL;
    size_t n = 0;
    double t[3] = {0, 0, 0};
    Decoded d;
    if (pass_cpu(e, q, &avif, &n, &d, t, e->pinned)) return -1;
    if (e->scorer == NULL) phase("first probe encoded and decoded");
    if (ensure_scorer(e) || pass_score(e->scorer, &d, out_score, t)) { decoded_free(&d); free(avif); return -1; }
    e->encode_ms += t[0]; e->decode_ms += t[1]; e->score_ms += t[2];
    free(e->buf); /* tq.zig:31-35: the buffer of this probe replaces the previous one */
    e->buf = avif; e->buf_size = n; e->buf_q = (int)q;
    return 0;
}

/* ---- the probes of one search fanned over scorer contexts and host threads (BASELINE
   configs[2]): oavif_tq_find_target_quality_speculative asks for waves of quantizers, each probe of a wave runs
   the pass above on its own context (= its own HIP stream) and thread.  OAVIF_PROBE_FANOUT=N, as the Python
   mirror; the result is the sequential search's (include/oavif_tq.h). ------------------------------------------ */
typedef struct {
    EncCtx* e;
    const uint8_t* rgb8;
    int blur;
    uint32_t fan;
    ssimu2_ctx* ctx[OAVIF_TQ_MAX_FANOUT];
    int have_ref[OAVIF_TQ_MAX_FANOUT]; /* a context gets the reference when a wave first uses it */
    uint8_t* pinned[OAVIF_TQ_MAX_FANOUT]; /* ... and its page-locked frame buffer (EncCtx.pinned) */
    struct { int q; uint8_t* b; size_t n; } kept[OAVIF_TQ_MAX_PASS * OAVIF_TQ_MAX_FANOUT]; /* every probe's bytes: any may be the answer */
    int nkept;
} Spec;
typedef struct { Spec* s; uint32_t slot, q; double score; int rc; } SpecJob;

static void* spec_job(void* arg) {
    SpecJob* j = (SpecJob*)arg;
    Spec* s = j->s;
    j->rc = -1;
    if (!s->have_ref[j->slot]) { /* a slot is used by one thread at a time */
        if (ssimu2_ctx_set_blur(s->ctx[j->slot], s->blur) || ssimu2_set_reference(s->ctx[j->slot], s->rgb8, s->e->src->w, s->e->src->h)) {
            fail("ScorerFailed", ssimu2_last_error(s->ctx[j->slot]));
            return NULL;
        }
        s->have_ref[j->slot] = 1;
    }
    if (!s->pinned[j->slot]) s->pinned[j->slot] = j->slot == 0 && s->e->pinned ? s->e->pinned : pinned_frame(s->ctx[j->slot], s->e->src);
    uint8_t* avif = NULL;
    size_t n = 0;
    double t[3] = {0, 0, 0};
    Decoded d;
    if (pass_cpu(s->e, j->q, &avif, &n, &d, t, s->pinned[j->slot])) return NULL;
    if (pass_score(s->ctx[j->slot], &d, &j->score, t)) { free(avif); return NULL; }
    pthread_mutex_lock(&g_lock);
    s->e->encode_ms += t[0]; s->e->decode_ms += t[1]; s->e->score_ms += t[2];
    if (s->nkept < (int)(sizeof s->kept / sizeof s->kept[0])) {
        s->kept[s->nkept].q = (int)j->q; s->kept[s->nkept].b = avif; s->kept[s->nkept].n = n;
        ++s->nkept;
        avif = NULL;
    }
    pthread_mutex_unlock(&g_lock);
    free(avif);
    j->rc = 0;
    return NULL;
}

static int spec_batch(void* user, const uint32_t* qs, uint32_t n, double* out_scores) {
    Spec* s = (Spec*)user;
    SpecJob jobs[OAVIF_TQ_MAX_FANOUT];
    pthread_t th[OAVIF_TQ_MAX_FANOUT];
    int started[OAVIF_TQ_MAX_FANOUT] = {0};
    if (n > s->fan) return fail("SearchFailed", "wave larger than the fan-out");
    for (uint32_t i = 0; i < n; ++i) {
        jobs[i].s = s; jobs[i].slot = i; jobs[i].q = qs[i]; jobs[i].score = 0; jobs[i].rc = -1;
        if (i + 1 == n) spec_job(&jobs[i]); /* the last probe of a wave runs on the calling thread */
        else started[i] = pthread_create(&th[i], NULL, spec_job, &jobs[i]) == 0;
        if (i + 1 < n && !started[i]) spec_job(&jobs[i]);
    }
    int rc = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (started[i]) pthread_join(th[i], NULL);
        if (jobs[i].rc) rc = -1;
        out_scores[i] = jobs[i].score;
    }
    return rc;
}

static int write_file(const char* path, const uint8_t* b, size_t n) {
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(b, 1, n, f) != n) { if (f) fclose(f); return fail("WriteFailed", path); }
    fclose(f);
    return 0;
}

typedef struct { /* everything run() allocates, released on every path */
    Image src;
    uint8_t* rgb8;
    void* scaled;
    EncCtx e;
    uint8_t* once;
    Spec spec;
} Run;

static void run_free(Run* r) {
    for (uint32_t i = 1; i < OAVIF_TQ_MAX_FANOUT; ++i)
        if (r->spec.ctx[i]) {
            ssimu2_host_free(r->spec.ctx[i], r->spec.pinned[i]);
            ssimu2_ctx_destroy(r->spec.ctx[i]);
        }
    for (int k = 0; k < r->spec.nkept; ++k) free(r->spec.kept[k].b);
    if (r->e.scorer) {
        ssimu2_host_free(r->e.scorer, r->e.pinned ? r->e.pinned : r->spec.pinned[0]);
        ssimu2_ctx_destroy(r->e.scorer);
    }
    if (r->e.image) av.ImageDestroy(r->e.image);
    free(r->e.buf);
    free(r->once);
    if (r->scaled != r->src.data) free(r->scaled);
    if (r->rgb8 != r->src.data) free(r->rgb8);
    free(r->src.data);
    free(r->src.icc);
}

static double g_t0;
static void phase(const char* what) { /* OAVIF_HOST_TIMES: where a one-image run spends its wall time */
    if (getenv("OAVIF_HOST_TIMES")) fprintf(stderr, "  [%7.1f ms] %s\n", now_ms() - g_t0, what);
}

static int run_inner(Run* r, int argc, char** argv) {
    Options o = OPTIONS_DEFAULT;
    const char *in = NULL, *out = NULL;
    if (parse_args(&o, argc, argv, &in, &out)) return -1;
    if (!in || !out) return fail("MissingInputOrOutput", NULL);
    const int device = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : 0;
    if (o.quality < 0) ssimu2_prefetch(device); /* the HIP start-up runs behind the image load and the first encode */
    phase("arguments parsed, scorer start-up begun in the background");
    if (load_libavif()) return -1;
    phase("libavif opened and checked");
    Image* src = &r->src;
    if (load_image(in, src)) return -1;
    phase("image loaded");
    struct stat st;
    memset(&st, 0, sizeof st);
    stat(in, &st);
    fprintf(stderr, "Read %ux%u, %s, %d-bit, %lld bytes\n", src->w, src->h, src->channels > 3 ? "RGBA" : "RGB",
            src->hbd ? 16 : 8, (long long)st.st_size);
    const size_t px = (size_t)src->w * src->h, nsamp = px * src->channels;
    /* e.rgb: Image.toRGB8 (io.zig:57-133) unless the source already is RGB8 (main.zig:86) */
    r->rgb8 = src->data;
    if (src->channels != 3 || src->hbd) {
        r->rgb8 = (uint8_t*)malloc(px * 3);
        if (!r->rgb8) return fail("OutOfMemory", NULL);
        for (size_t i = 0; i < px; ++i)
            for (int c = 0; c < 3; ++c)
                r->rgb8[i * 3 + c] = src->hbd ? (uint8_t)(((const uint16_t*)src->data)[i * src->channels + c] >> 8)
                                              : src->data[i * src->channels + c];
    }
    /* io.zig:546; a libaom without high-bit-depth support (this image's) cannot write 10-bit: probed once */
    uint32_t out_depth = (o.tenbit || src->hbd) ? 10 : 8;
    const char* depth_note = NULL;
    if (out_depth == 10) {
        Options po = o;
        po.speed = 10;
        uint16_t tiny[16 * 16 * 3];
        for (int i = 0; i < 16 * 16 * 3; ++i) tiny[i] = 512;
        Image ti = {16, 16, 3, 0, (uint8_t*)tiny, NULL, 0};
        EncCtx pe;
        memset(&pe, 0, sizeof pe);
        pe.o = &po; pe.src = &ti; pe.scaled = tiny; pe.out_depth = 10; pe.buf_q = -1;
        uint8_t* pb = NULL;
        size_t pn = 0;
        const int refused = encode_to_buffer(&pe, 50, &pb, &pn);
        if (pe.image) av.ImageDestroy(pe.image);
        if (refused) {
            out_depth = 8;
            depth_note = "note: the reference would write 10-bit here (--tenbit 1 / 16-bit source, io.zig:546-548); this "
                         "image's libaom has no high-bit-depth support, so the bitstream is 8-bit";
        }
        free(pb);
        g_err = NULL;
        g_detail[0] = 0;
    }
    /* io.zig:566-617, hoisted out of the pass loop */
    r->scaled = src->data;
    if (!src->hbd && out_depth == 10) {
        r->scaled = malloc(nsamp * 2);
        if (!r->scaled) return fail("OutOfMemory", NULL);
        oavif_prescale_8_to_10(src->data, nsamp, (uint16_t*)r->scaled);
    } else if (src->hbd && out_depth == 10) {
        r->scaled = malloc(nsamp * 2);
        if (!r->scaled) return fail("OutOfMemory", NULL);
        oavif_prescale_16_to_10((const uint16_t*)src->data, nsamp, (uint16_t*)r->scaled);
    } else if (src->hbd) {
        r->scaled = malloc(nsamp);
        if (!r->scaled) return fail("OutOfMemory", NULL);
        oavif_prescale_16_to_8((const uint16_t*)src->data, nsamp, (uint8_t*)r->scaled);
    }
    EncCtx* e = &r->e;
    e->o = &o; e->src = src; e->scaled = r->scaled; e->out_depth = out_depth; e->buf_q = -1;
    if (o.quality >= 0) { /* main.zig:93-100 */
        fprintf(stderr, "Encoding [q%d, speed %d, %u-bit]\n", o.quality, o.speed, out_depth);
        size_t n = 0;
        if (encode_to_buffer(e, (uint32_t)o.quality, &r->once, &n) || write_file(out, r->once, n)) return -1;
        fprintf(stderr, "Compressed to %zu bytes (%.3f bpp)\n", n, n * 8.0 / (double)px);
        if (depth_note) fprintf(stderr, "%s\n", depth_note);
        return 0;
    }
    if (o.score_tgt == floor(o.score_tgt)) /* Zig's {} on an f64 prints 80 for 80.0 */
        fprintf(stderr, "Searching [tgt %.0f±%.1f, speed %d, %u-bit]\n", o.score_tgt, o.tolerance, o.speed, out_depth);
    else
        fprintf(stderr, "Searching [tgt %g±%.1f, speed %d, %u-bit]\n", o.score_tgt, o.tolerance, o.speed, out_depth);
    /* the blur of the search path: the published recursion unless OAVIF_SSIMU2_BLUR says otherwise (as the
       Zig shim's `blur` and the Python mirror: INTEGRATION.md section 2e) */
    const char* bm = getenv("OAVIF_SSIMU2_BLUR");
    int mode = SSIMU2_BLUR_RECURSIVE;
    if (bm && !strcmp(bm, "fir")) mode = SSIMU2_BLUR_FIR;
    else if (bm && (!strcmp(bm, "recursive_fma") || !strcmp(bm, "iir_fma"))) mode = SSIMU2_BLUR_RECURSIVE_FMA;
    e->device = device; e->blur = mode; e->rgb8 = r->rgb8;
    int rc;
    if (make_source_image(e)) return -1;
    phase("source converted to YUV444");
    /* The scorer context is NOT created here: the HIP runtime started initialising in the background at process
       start (ssimu2_prefetch) and takes 0.1-0.25 s; the CPU half of the first pass -- its quantizer does not
       depend on any score (tq.zig:136) -- runs meanwhile, and the first score waits for whatever is left. */
    oavif_tq_options to = {o.score_tgt, o.tolerance, (uint32_t)o.max_pass};
    oavif_tq_result res;
    int spec_used = 0;
    oavif_tq_spec_stats spec_stats = {0, 0, 0};
    const int fan = getenv("OAVIF_PROBE_FANOUT") ? atoi(getenv("OAVIF_PROBE_FANOUT")) : 1;
    if (fan > 1) { /* not a CLI flag: the option surface stays the reference's (parse_args.zig:76-122) */
        Spec* sp = &r->spec; /* the source image exists before the first wave: the threads only read it */
        sp->e = e; sp->rgb8 = r->rgb8; sp->blur = mode;
        sp->fan = fan > OAVIF_TQ_MAX_FANOUT ? OAVIF_TQ_MAX_FANOUT : (uint32_t)fan;
        if (ensure_scorer(e)) return -1; /* the fan-out needs its contexts up front */
        sp->ctx[0] = e->scorer;
        sp->have_ref[0] = 1;
        for (uint32_t i = 1; i < sp->fan; ++i) /* contexts are made here, on one thread; each is one HIP stream + scratch */
            if (ssimu2_ctx_create(device, NULL, &sp->ctx[i]) != SSIMU2_OK) return fail("ScorerFailed", ssimu2_last_error(NULL));
        oavif_tq_spec_options so = OAVIF_TQ_SPEC_OPTIONS_INIT(sp->fan, 1);
        oavif_tq_spec_stats sst;
        rc = oavif_tq_find_target_quality_speculative(&to, &so, spec_batch, sp, &res, &sst);
        if (rc) return g_err ? -1 : fail("SearchFailed", NULL);
        for (int k = 0; k < sp->nkept; ++k)
            if (sp->kept[k].q == (int)res.q && !e->buf) { /* EncBuffer: here the bytes of the chosen q, if it was probed */
                e->buf = sp->kept[k].b; e->buf_size = sp->kept[k].n; e->buf_q = sp->kept[k].q;
                sp->kept[k].b = NULL;
            }
        spec_used = 1; spec_stats = sst;
    } else {
        rc = oavif_tq_find_target_quality(&to, probe, e, &res);
        if (rc) return g_err ? -1 : fail("SearchFailed", NULL);
    }
    phase("search done");
    fprintf(stderr, "Found q%u (score %.2f, %u passes)\n", res.q, res.score, res.num_pass);
    if (e->buf_q == (int)res.q) { /* main.zig:109-113 */
        if (write_file(out, e->buf, e->buf_size)) return -1;
    } else {
        free(e->buf);
        e->buf = NULL;
        if (encode_to_buffer(e, res.q, &e->buf, &e->buf_size) || write_file(out, e->buf, e->buf_size)) return -1;
    }
    fprintf(stderr, "Compressed to %zu bytes (%.3f bpp)\n", e->buf_size, e->buf_size * 8.0 / (double)px);
    if (depth_note) fprintf(stderr, "%s\n", depth_note);
    if (getenv("OAVIF_HOST_TIMES") && spec_used)
        fprintf(stderr, "speculative: %u waves, %u probes issued, %u cache hits\n", spec_stats.waves,
                spec_stats.probes_issued, spec_stats.cache_hits);
    if (getenv("OAVIF_HOST_TIMES")) /* not one of the reference's lines: only on request */
        fprintf(stderr, "times: encode %.1f ms, decode %.1f ms, upload+score %.2f ms over %u passes\n", e->encode_ms,
                e->decode_ms, e->score_ms, res.num_pass);
    return 0;
}

static int run(int argc, char** argv) {
    Run r;
    memset(&r, 0, sizeof r);
    g_t0 = now_ms();
    const int rc = run_inner(&r, argc, argv);
    phase("output written");
    run_free(&r);
    phase("contexts and buffers released");
    return rc;
}

/* parse_args.zig:180-238: the usage text with the defaults of the options struct filled in */
static void print_usage(void) {
    const Options d = OPTIONS_DEFAULT;
    fprintf(stderr,
            "\nusage:  oavif [options] <in> <out.avif>\n\noptions:\n"
            " -h, --help\n    show this help\n"
            " -v, --version\n    show version information\n"
            " -s, --speed u8\n    encoder speed (0..10) [%d]\n"
            " -t, --score-tgt f64\n    target SSIMULACRA2 score (0..100) [%.0f]\n"
            " --quality-alpha u8\n    quality factor for alpha (0..100=lossless) [%d]\n"
            " --max-threads u8\n    maximum number of threads to use (1..255) [%d]\n"
            " --tile-rows-log2 u8\n    tile rows log2 (0..6) [%d]\n"
            " --tile-cols-log2 u8\n    tile columns log2 (0..6) [%d]\n"
            " --auto-tiling 0/1\n    enable automatic tiling [%d]\n"
            " --tune str\n    libaom tuning mode (ssim, iq, ssimulacra2) [%s]\n"
            " --tenbit 0/1\n    force 10-bit AVIF output [%d]\n"
            " --tolerance f64\n    target quality error tolerance (1..100) [%.0f]\n"
            " --max-pass u8\n    maximum search passes (1..12) [%d]\n"
            " -q, --quality u8\n    quantizer (0..100), bypasses search\n"
            " --color-primaries u8\n    color primaries (1..22) [%d]\n"
            " --transfer-characteristics u8\n    transfer characteristics (1..18) [%d]\n"
            " --matrix-coefficients u8\n    matrix coefficients (0..14) [%d]"
            "\n\n\x1b[37mInput image formats: PNG, PAM, JPEG, WebP, or AVIF\x1b[0m\n",
            d.speed, d.score_tgt, d.quality_alpha, d.max_threads, d.tile_rows_log2, d.tile_cols_log2, d.auto_tiling, d.tune,
            d.tenbit, d.tolerance, d.max_pass, d.color_primaries, d.transfer_characteristics, d.matrix_coefficients);
}

/* io.printVersion (io.zig:14-39) for what this host links: itself, the scorer library, libavif and its codecs */
static void print_version(void) {
    fprintf(stderr, "oavif %s\nscorer %s\n", VERSION, ssimu2_version());
    if (load_libavif() == 0) fprintf(stderr, "libavif %s\n", av.Version());
}

int main(int argc, char** argv) {
    fprintf(stderr, "\x1b[31moavif\x1b[0m | %s\n", VERSION);
    /* main.zig:46-61: -h/--help and -v/--version count only while they are the leading arguments */
    int show_help = 0, show_version = 0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--help") || !strcmp(argv[i], "-h")) show_help = 1;
        else if (!strcmp(argv[i], "--version") || !strcmp(argv[i], "-v")) show_version = 1;
        else break;
    }
    if (show_help) { print_usage(); return 0; }
    if (show_version) { print_version(); return 0; }
    int rc = 0;
    if (run(argc, argv)) {
        fprintf(stderr, "error: %s%s%s\n", g_err ? g_err : "Unexpected", g_detail[0] ? ": " : "", g_detail);
        rc = 1;
    }
    /* Everything this process owns is released and its output is on disk.  The HIP runtime's exit handlers then
       spend another 50-100 ms unloading code objects and tearing the device context down -- a third of a 1080p
       run -- for memory the kernel reclaims anyway: leave without them (OAVIF_HOST_ATEXIT=1 keeps them, for
       leak checkers). */
    fflush(NULL);
    if (!getenv("OAVIF_HOST_ATEXIT")) _exit(rc);
    return rc;
}
