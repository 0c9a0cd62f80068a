/*
 * oracle/ssimu2_oracle.c -- CPU restatement of the SSIMULACRA2 score that oavif's
 * target-quality search computes once per pass.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oavif_amd/ (the product) may link, load or
 * call this file.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * use it, and only as the checker / the timed CPU baseline.
 *
 * What it restates
 * ----------------
 * The reference calls the scorer at exactly one place:
 *     /root/reference/src/tq.zig:37
 *         return try fssimu2.computeSsimu2(allocator, e.rgb, decoded_rgb, e.w, e.h, 3, null);
 * `fssimu2` is a third-party Zig package pinned by URL + hash at
 * /root/reference/build.zig.zon:7-10 (tag 0.1.1).  Its source is NOT in /root/reference
 * and NOT on this machine (no vendored copy, no package cache, no network), and the
 * reference has no tests, golden vectors or fixtures for this path.
 *
 *     ==> PARITY UNPINNED against fssimu2 0.1.1. <==
 *
 * What is restated instead is the *published* SSIMULACRA2 v2.1 algorithm (libjxl
 * tools/ssimulacra2.cc + lib/jxl/gauss_blur.cc + lib/jxl/enc_xyb.cc), written from the
 * public description of that algorithm, which is the algorithm BASELINE.json's
 * north_star lists stage by stage (sRGB->linear->XYB, 6-scale 2x downsample, separable
 * Gaussian blur, SSIM + edge-difference maps, 108-weight reduction).  Input contract is
 * the reference's own: two tightly packed 8-bit interleaved RGB buffers of w*h*3 bytes
 * (/root/reference/src/main.zig:86, /root/reference/src/io.zig:647-663).
 *
 * Two blur formulations, selectable per call, proven equivalent in exact arithmetic by
 * tests/test_oracle.py:
 *   OR_BLUR_IIR (0)  libjxl's recursive Gaussian (Charalampidis 2016, sigma 1.5, three
 *                    undamped 2nd-order sections fed by in[n-N-1]+in[n+N-1], N=5, zero
 *                    outside the image), fp32 state, horizontal pass then vertical.
 *   OR_BLUR_FIR (1)  the same operator written as what it is in exact arithmetic: a
 *                    symmetric 9-tap FIR (taps -4..4) with zero padding, fp32.
 * The fp32 recursion is NOT numerically equivalent: its sections are undamped, so
 * rounding error random-walks along every row and column (measured rms 7e-7 vs 1.4e-8
 * for the FIR on a 512x512 plane) and, through the max(0, .) in the maps, biases scores
 * by ~0.01-0.1 points near 80 and up to ~1.4 points near 95.  That noise is a property
 * of one fp32 evaluation order (libjxl's own SIMD targets differ from each other in it),
 * not of the algorithm, so the FIR form is the oracle's PRIMARY mode and the one the HIP
 * path implements (LDS rows with halo) and is compared with; the IIR mode is kept to
 * quantify the gap (tests/test_oracle.py, DESIGN.md "Oracle").
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; -fopenmp variant for timing).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define OR_NUM_SCALES 6
#define OR_BLUR_IIR 0
#define OR_BLUR_FIR 1
#define OR_BLUR_IIR_FMA 3 /* OR_BLUR_IIR with the recursion's multiply-subtract fused (what an FMA
                             target of libjxl's MulAdd evaluates): a second, equally legitimate
                             fp32 evaluation order, kept only to measure how far two recursive
                             implementations differ from EACH OTHER (tests/tools/) */
#define OR_BLUR_FIR_PRODFIRST 4 /* OR_BLUR_FIR with the product planes x*x, y*y, x*y rounded to fp32
                                   FIRST and then blurred like any plane -- the published order of
                                   operations (multiply, then blur) with the FIR in place of the
                                   recursion.  Measures what fusing the products into the pair sums
                                   (OR_BLUR_FIR, the kernels' contract) moves: tests/test_oracle.py */
#define OR_BLUR_EXACT 2 /* the same 9-tap operator accumulated in fp64, result rounded to fp32:
                          what both fp32 forms approximate (evidence for DESIGN.md 2.1) */

/* ---- stage variants (the pin kit: tests/golden/pin_kit, scripts/pin_blur_mode.py) -------------
 * fssimu2's source is not available, so WHERE it might differ from the published algorithm is not
 * known either.  Besides the blur's evaluation order (the modes above), these bits switch single
 * stages to the cheap, plausible alternatives a re-implementation could have taken; every one is
 * OFF in the oracle proper (or_compute_ssimu2) and none has a counterpart in the HIP kernels.  A
 * maintainer who can run fssimu2 scores the kit's pairs once and scripts/pin_blur_mode.py names the
 * nearest variant -- i.e. the STAGE that differs -- instead of "no mode matches".
 *   blur      EDGE_CLAMP / EDGE_MIRROR  samples outside the plane replicate the edge / mirror about
 *                                       it instead of being zero (FIR family only)
 *             GAUSS9 / GAUSS11          a true sampled sigma-1.5 Gaussian truncated at radius 4 / 5 and
 *                                       normalised, in place of the recursion's impulse response
 *   pyramid   DOWNSAMPLE_XYB            scale s+1 = 2x2 average of scale s's XYB planes (not of linear
 *                                       light, re-converted)
 *             DOWNSAMPLE_FLOOR          odd sizes drop the last row / column (floor(w/2)) instead of
 *                                       replicating it (ceil(w/2))
 *             SIZE_TEST_AFTER           a scale is scored only if ITS size is >= 8 (the published loop
 *                                       tests the size before downsampling)
 *   colour    SRGB_POWF                 the transfer function in fp32 (powf) instead of a table from fp64
 *             CBRT_LIBM                 libm's cbrtf instead of or_cbrtf
 *   maps      SUMS_F32                  the per-plane sums accumulated in fp32
 * The generic convolution the blur variants use multiplies first and blurs then (published order). */
#define OR_VAR_EDGE_CLAMP 0x1u
#define OR_VAR_EDGE_MIRROR 0x2u
#define OR_VAR_GAUSS9 0x4u
#define OR_VAR_GAUSS11 0x8u
#define OR_VAR_DOWNSAMPLE_XYB 0x10u
#define OR_VAR_DOWNSAMPLE_FLOOR 0x20u
#define OR_VAR_SIZE_TEST_AFTER 0x40u
#define OR_VAR_SRGB_POWF 0x80u
#define OR_VAR_CBRT_LIBM 0x100u
#define OR_VAR_SUMS_F32 0x200u
#define OR_VAR_BLUR_MASK (OR_VAR_EDGE_CLAMP | OR_VAR_EDGE_MIRROR | OR_VAR_GAUSS9 | OR_VAR_GAUSS11)
#define OR_VAR_ALL 0x3FFu

/* ---- constants of the published algorithm ------------------------------------- */

static const float kC2 = 0.0009f;

/* opsin absorbance (libjxl opsin_params.h), rows L, M, S */
static const float kM00 = 0.30f, kM01 = 0.622f, kM02 = 0.078f;
static const float kM10 = 0.23f, kM11 = 0.692f, kM12 = 0.078f;
static const float kM20 = 0.24342268924547819f, kM21 = 0.20476744424496821f,
                   kM22 = 0.55180986650955360f; /* 1 - kM20 - kM21 */
static const float kOpsinBias = 0.0037930732552754493f;

/* 108 trained weights; for a six-scale image index = ((channel*6 + scale)*2 + norm)*3 +
   {ssim, artifact, detail_lost} (see or_score_from_averages for the running index). */
static const double kWeights[108] = {
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

/* ---- recursive Gaussian coefficients (Charalampidis 2016; libjxl gauss_blur.cc) -- */

typedef struct {
    int radius;      /* N = 5 for sigma 1.5 */
    float n2[3];     /* input gains of the three sections */
    float d1[3];     /* -2 cos(omega_k) */
    double n2d[3], d1d[3];
    float fir[5];    /* equivalent FIR taps |d| = 0..4 (fp32 of the exact value) */
    double fird[5];
} or_gauss;

static void inv3x3(double m[9]) {
    double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7],
           i = m[8];
    double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    double det = a * A + b * B + c * C;
    double r[9] = {A, -(b * i - c * h), b * f - c * e,
                   B, a * i - c * g, -(a * f - c * d),
                   C, -(a * h - b * g), a * e - b * d};
    for (int k = 0; k < 9; ++k) m[k] = r[k] / det;
}

void or_gauss_create(double sigma, or_gauss* rg) {
    const double kPi = 3.141592653589793238;
    const double radius = round(3.2795 * sigma + 0.2546);
    const double pi_div_2r = kPi / (2.0 * radius);
    const double omega[3] = {pi_div_2r, 3.0 * pi_div_2r, 5.0 * pi_div_2r};
    const double p_1 = +1.0 / tan(0.5 * omega[0]);
    const double p_3 = -1.0 / tan(0.5 * omega[1]);
    const double p_5 = +1.0 / tan(0.5 * omega[2]);
    const double r_1 = +p_1 * p_1 / sin(omega[0]);
    const double r_3 = -p_3 * p_3 / sin(omega[1]);
    const double r_5 = +p_5 * p_5 / sin(omega[2]);
    const double neg_half_sigma2 = -0.5 * sigma * sigma;
    double rho[3];
    for (int i = 0; i < 3; ++i) rho[i] = exp(neg_half_sigma2 * omega[i] * omega[i]) / radius;
    const double D_13 = p_1 * r_3 - r_1 * p_3;
    const double D_35 = p_3 * r_5 - r_3 * p_5;
    const double D_51 = p_5 * r_1 - r_5 * p_1;
    const double zeta_15 = D_35 / D_13;
    const double zeta_35 = D_51 / D_13;
    double A[9] = {p_1, p_3, p_5, r_1, r_3, r_5, zeta_15, zeta_35, 1.0};
    inv3x3(A);
    const double gamma[3] = {1.0, radius * radius - sigma * sigma,
                             zeta_15 * rho[0] + zeta_35 * rho[1] + rho[2]};
    double beta[3];
    for (int i = 0; i < 3; ++i)
        beta[i] = A[3 * i] * gamma[0] + A[3 * i + 1] * gamma[1] + A[3 * i + 2] * gamma[2];
    rg->radius = (int)radius;
    for (int i = 0; i < 3; ++i) {
        rg->n2d[i] = -beta[i] * cos(omega[i] * (radius + 1.0));
        rg->d1d[i] = -2.0 * cos(omega[i]);
        rg->n2[i] = (float)rg->n2d[i];
        rg->d1[i] = (float)rg->d1d[i];
    }
    /* Impulse response of the three sections driven by in[n-N-1]+in[n+N-1]:
       w(d) = sum_k n2_k/sin(w_k) * sin(w_k (d+N)), d in [-N+1, N-1]; zero elsewhere. */
    for (int d = 0; d < 5; ++d) {
        double w = 0.0;
        for (int k = 0; k < 3; ++k)
            w += rg->n2d[k] / sin(omega[k]) * sin(omega[k] * (d + radius));
        rg->fird[d] = w;
        rg->fir[d] = (float)w;
    }
}

/* exported for tests: taps as fp64 and fp32 */
void or_gauss_taps(double* taps5_f64, float* taps5_f32, double* n2_3, double* d1_3) {
    or_gauss rg;
    or_gauss_create(1.5, &rg);
    for (int i = 0; i < 5; ++i) {
        taps5_f64[i] = rg.fird[i];
        taps5_f32[i] = rg.fir[i];
    }
    for (int i = 0; i < 3; ++i) {
        n2_3[i] = rg.n2d[i];
        d1_3[i] = rg.d1d[i];
    }
}

/* 1-D recursive pass over a strided line (scalar form of libjxl FastGaussian1D). */
static void iir_line(const or_gauss* rg, const float* in, ptrdiff_t n_in, ptrdiff_t stride_in,
                     float* out, ptrdiff_t stride_out, int fused) {
    const ptrdiff_t N = rg->radius;
    const float n2_1 = rg->n2[0], n2_3 = rg->n2[1], n2_5 = rg->n2[2];
    const float d1_1 = rg->d1[0], d1_3 = rg->d1[1], d1_5 = rg->d1[2];
    float prev_1 = 0, prev_3 = 0, prev_5 = 0, prev2_1 = 0, prev2_3 = 0, prev2_5 = 0;
    for (ptrdiff_t n = -N + 1; n < n_in; ++n) {
        const ptrdiff_t left = n - N - 1, right = n + N - 1;
        const float lv = left >= 0 ? in[left * stride_in] : 0.0f;
        const float rv = right < n_in ? in[right * stride_in] : 0.0f;
        const float sum = lv + rv;
        float o1 = sum * n2_1, o3 = sum * n2_3, o5 = sum * n2_5;
        o1 = o1 - prev2_1;
        o3 = o3 - prev2_3;
        o5 = o5 - prev2_5;
        if (fused) {
            o1 = fmaf(-d1_1, prev_1, o1);
            o3 = fmaf(-d1_3, prev_3, o3);
            o5 = fmaf(-d1_5, prev_5, o5);
        } else {
            o1 = o1 - d1_1 * prev_1;
            o3 = o3 - d1_3 * prev_3;
            o5 = o5 - d1_5 * prev_5;
        }
        prev2_1 = prev_1; prev2_3 = prev_3; prev2_5 = prev_5;
        prev_1 = o1; prev_3 = o3; prev_5 = o5;
        if (n >= 0) out[n * stride_out] = o1 + o3 + o5;
    }
}

static void fir_line(const or_gauss* rg, const float* in, ptrdiff_t n_in, ptrdiff_t stride_in,
                     float* out, ptrdiff_t stride_out) {
    const float w0 = rg->fir[0], w1 = rg->fir[1], w2 = rg->fir[2], w3 = rg->fir[3],
                w4 = rg->fir[4];
#define AT(i) (((i) >= 0 && (i) < n_in) ? in[(i) * stride_in] : 0.0f)
    /* Operation order is part of the contract with the HIP kernels (they evaluate the
       same mul + four fused multiply-adds), so the two blurs agree bit for bit. */
    for (ptrdiff_t n = 0; n < n_in; ++n) {
        if (stride_in == 1 && stride_out == 1 && n == 4 && n_in > 8) {
            /* interior of a contiguous line: no bounds tests, vectorisable; same operations */
            for (; n < n_in - 4; ++n) {
                float acc = w0 * in[n];
                acc = fmaf(w1, in[n - 1] + in[n + 1], acc);
                acc = fmaf(w2, in[n - 2] + in[n + 2], acc);
                acc = fmaf(w3, in[n - 3] + in[n + 3], acc);
                acc = fmaf(w4, in[n - 4] + in[n + 4], acc);
                out[n] = acc;
            }
        }
        float acc = w0 * AT(n);
        acc = fmaf(w1, AT(n - 1) + AT(n + 1), acc);
        acc = fmaf(w2, AT(n - 2) + AT(n + 2), acc);
        acc = fmaf(w3, AT(n - 3) + AT(n + 3), acc);
        acc = fmaf(w4, AT(n - 4) + AT(n + 4), acc);
        out[n * stride_out] = acc;
    }
#undef AT
}

/* Vertical 9-tap of a whole plane, row by row (cache-friendly and vectorisable): out[y][x] is
   exactly what fir_line gives on column x -- rows outside the plane are the zero padding. */
static void fir_columns(const or_gauss* rg, const float* in, size_t w, size_t h, float* out) {
    const float w0 = rg->fir[0], w1 = rg->fir[1], w2 = rg->fir[2], w3 = rg->fir[3],
                w4 = rg->fir[4];
    float* zero = (float*)calloc(w, sizeof(float));
#pragma omp parallel for schedule(static)
    for (ptrdiff_t y = 0; y < (ptrdiff_t)h; ++y) {
        const float* r[9];
        for (int d = -4; d <= 4; ++d) {
            const ptrdiff_t yy = y + d;
            r[d + 4] = (yy >= 0 && yy < (ptrdiff_t)h) ? in + (size_t)yy * w : zero;
        }
        float* o = out + (size_t)y * w;
        for (size_t x = 0; x < w; ++x) {
            float acc = w0 * r[4][x];
            acc = fmaf(w1, r[3][x] + r[5][x], acc);
            acc = fmaf(w2, r[2][x] + r[6][x], acc);
            acc = fmaf(w3, r[1][x] + r[7][x], acc);
            acc = fmaf(w4, r[0][x] + r[8][x], acc);
            o[x] = acc;
        }
    }
    free(zero);
}

/* Horizontal 9-tap of the product plane a*b without materialising it: the centre tap is
   a*b, every symmetric pair sum is fma(a[-d], b[-d], a[+d]*b[+d]) (one rounding fewer than
   two products and an add, and one operation fewer: the HIP kernels do exactly this). */
static void fir_line_prod(const or_gauss* rg, const float* a, const float* b, ptrdiff_t n_in,
                          float* out) {
    const float w0 = rg->fir[0], w1 = rg->fir[1], w2 = rg->fir[2], w3 = rg->fir[3],
                w4 = rg->fir[4];
#define A(i) (((i) >= 0 && (i) < n_in) ? a[i] : 0.0f)
#define B(i) (((i) >= 0 && (i) < n_in) ? b[i] : 0.0f)
#define PAIR(d) fmaf(A(n - (d)), B(n - (d)), A(n + (d)) * B(n + (d)))
    for (ptrdiff_t n = 0; n < n_in; ++n) {
        if (n == 4 && n_in > 8) {
            /* interior: no bounds tests, vectorisable; same operations */
#define PAIRI(d) fmaf(a[n - (d)], b[n - (d)], a[n + (d)] * b[n + (d)])
            for (; n < n_in - 4; ++n) {
                float acc = w0 * (a[n] * b[n]);
                acc = fmaf(w1, PAIRI(1), acc);
                acc = fmaf(w2, PAIRI(2), acc);
                acc = fmaf(w3, PAIRI(3), acc);
                acc = fmaf(w4, PAIRI(4), acc);
                out[n] = acc;
            }
#undef PAIRI
        }
        float acc = w0 * (A(n) * B(n));
        acc = fmaf(w1, PAIR(1), acc);
        acc = fmaf(w2, PAIR(2), acc);
        acc = fmaf(w3, PAIR(3), acc);
        acc = fmaf(w4, PAIR(4), acc);
        out[n] = acc;
    }
#undef PAIR
#undef A
#undef B
}

/* 2-D blur of the product plane a*b (FIR mode: products formed inside the horizontal pass,
   see fir_line_prod; IIR mode: the published form, product plane materialised first). */
static void blur_plane_prod(const or_gauss* rg, int mode, const float* a, const float* b, size_t w,
                            size_t h, float* prod_tmp, float* tmp, float* out);

/* fp64 evaluation of the 9-tap operator on a strided line of doubles */
static void fir_line_f64(const or_gauss* rg, const double* in, ptrdiff_t n_in, ptrdiff_t stride_in,
                         double* out, ptrdiff_t stride_out) {
#define AT(i) (((i) >= 0 && (i) < n_in) ? in[(i) * stride_in] : 0.0)
    for (ptrdiff_t n = 0; n < n_in; ++n) {
        double acc = rg->fird[0] * AT(n);
        for (int d = 1; d <= 4; ++d) acc += rg->fird[d] * (AT(n - d) + AT(n + d));
        out[n * stride_out] = acc;
    }
#undef AT
}

/* 2-D blur in fp64 of an fp32 plane (or of the product a*b when b != NULL), rounded once */
static void blur_plane_exact(const or_gauss* rg, const float* a, const float* b, size_t w, size_t h,
                             float* out) {
    const size_t n = w * h;
    double* p = (double*)malloc(sizeof(double) * n);
    double* t = (double*)malloc(sizeof(double) * n);
    for (size_t i = 0; i < n; ++i) p[i] = b ? (double)a[i] * (double)b[i] : (double)a[i];
    for (size_t y = 0; y < h; ++y) fir_line_f64(rg, p + y * w, w, 1, t + y * w, 1);
    for (size_t x = 0; x < w; ++x) fir_line_f64(rg, t + x, h, w, p + x, w);
    for (size_t i = 0; i < n; ++i) out[i] = (float)p[i];
    free(p);
    free(t);
}

/* 2-D blur of one w*h plane: horizontal into tmp, vertical into out. */
static void blur_plane(const or_gauss* rg, int mode, const float* in, size_t w, size_t h,
                       float* tmp, float* out) {
#pragma omp parallel for schedule(static)
    for (ptrdiff_t y = 0; y < (ptrdiff_t)h; ++y) {
        if (mode == OR_BLUR_IIR || mode == OR_BLUR_IIR_FMA)
            iir_line(rg, in + y * w, w, 1, tmp + y * w, 1, mode == OR_BLUR_IIR_FMA);
        else fir_line(rg, in + y * w, w, 1, tmp + y * w, 1);
    }
    if (mode == OR_BLUR_IIR || mode == OR_BLUR_IIR_FMA) {
#pragma omp parallel for schedule(static)
        for (ptrdiff_t x = 0; x < (ptrdiff_t)w; ++x)
            iir_line(rg, tmp + x, h, w, out + x, w, mode == OR_BLUR_IIR_FMA);
    } else {
        fir_columns(rg, tmp, w, h, out);
    }
}

static void blur_plane_prod(const or_gauss* rg, int mode, const float* a, const float* b, size_t w,
                            size_t h, float* prod_tmp, float* tmp, float* out) {
    if (mode == OR_BLUR_IIR || mode == OR_BLUR_IIR_FMA || mode == OR_BLUR_FIR_PRODFIRST) {
        const size_t n = w * h;
#pragma omp parallel for schedule(static)
        for (ptrdiff_t i = 0; i < (ptrdiff_t)n; ++i) prod_tmp[i] = a[i] * b[i];
        blur_plane(rg, mode == OR_BLUR_FIR_PRODFIRST ? OR_BLUR_FIR : mode, prod_tmp, w, h, tmp, out);
        return;
    }
#pragma omp parallel for schedule(static)
    for (ptrdiff_t y = 0; y < (ptrdiff_t)h; ++y)
        fir_line_prod(rg, a + y * w, b + y * w, w, tmp + y * w);
    fir_columns(rg, tmp, w, h, out);
}

/* ---- blur variants: a plain fp32 convolution with selectable taps and edge rule ----------------- */
typedef struct {
    int radius;     /* taps -radius..radius, symmetric */
    float tap[8];   /* tap[d], d = 0..radius */
    int edge;       /* 0 zero, 1 clamp, 2 mirror (about the edge sample: -1 -> 1) */
} or_conv;

static void conv_setup(const or_gauss* rg, unsigned variant, or_conv* cv) {
    memset(cv, 0, sizeof *cv);
    cv->edge = (variant & OR_VAR_EDGE_CLAMP) ? 1 : (variant & OR_VAR_EDGE_MIRROR) ? 2 : 0;
    if (variant & (OR_VAR_GAUSS9 | OR_VAR_GAUSS11)) {
        cv->radius = (variant & OR_VAR_GAUSS11) ? 5 : 4;
        double t[8], sum = 0.0;
        for (int d = 0; d <= cv->radius; ++d) {
            t[d] = exp(-(double)(d * d) / (2.0 * 1.5 * 1.5));
            sum += d ? 2.0 * t[d] : t[d];
        }
        for (int d = 0; d <= cv->radius; ++d) cv->tap[d] = (float)(t[d] / sum);
    } else {
        cv->radius = 4;
        for (int d = 0; d <= 4; ++d) cv->tap[d] = rg->fir[d];
    }
}

static inline float conv_at(const float* in, ptrdiff_t i, ptrdiff_t n, ptrdiff_t stride, int edge) {
    if (i >= 0 && i < n) return in[i * stride];
    if (edge == 0) return 0.0f;
    if (edge == 1) return in[(i < 0 ? 0 : n - 1) * stride];
    if (n == 1) return in[0];
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i; /* mirror without repeating the edge sample */
    return in[i * stride];
}

static void conv_line(const or_conv* cv, const float* in, ptrdiff_t n, ptrdiff_t stride_in, float* out,
                      ptrdiff_t stride_out) {
    for (ptrdiff_t i = 0; i < n; ++i) {
        float acc = cv->tap[0] * conv_at(in, i, n, stride_in, cv->edge);
        for (int d = 1; d <= cv->radius; ++d)
            acc = fmaf(cv->tap[d], conv_at(in, i - d, n, stride_in, cv->edge) + conv_at(in, i + d, n, stride_in, cv->edge), acc);
        out[i * stride_out] = acc;
    }
}

static void conv_plane(const or_conv* cv, const float* in, size_t w, size_t h, float* tmp, float* out) {
#pragma omp parallel for schedule(static)
    for (ptrdiff_t y = 0; y < (ptrdiff_t)h; ++y) conv_line(cv, in + y * w, (ptrdiff_t)w, 1, tmp + y * w, 1);
#pragma omp parallel for schedule(static)
    for (ptrdiff_t x = 0; x < (ptrdiff_t)w; ++x) conv_line(cv, tmp + x, (ptrdiff_t)h, (ptrdiff_t)w, out + x, (ptrdiff_t)w);
}

/* exported for tests: blur one plane */
void or_blur_plane(const float* in, uint32_t w, uint32_t h, int mode, float* out) {
    or_gauss rg;
    or_gauss_create(1.5, &rg);
    float* tmp = (float*)malloc(sizeof(float) * (size_t)w * h);
    blur_plane(&rg, mode, in, w, h, tmp, out);
    free(tmp);
}

/* exported for tests: blur the product plane a*b the way the score does in `mode` */
void or_blur_product(const float* a, const float* b, uint32_t w, uint32_t h, int mode, float* out) {
    or_gauss rg;
    or_gauss_create(1.5, &rg);
    const size_t n = (size_t)w * h;
    float* tmp = (float*)malloc(sizeof(float) * n);
    float* prod = (float*)malloc(sizeof(float) * n);
    if (mode == OR_BLUR_EXACT) blur_plane_exact(&rg, a, b, w, h, out);
    else blur_plane_prod(&rg, mode, a, b, w, h, prod, tmp, out);
    free(tmp);
    free(prod);
}

/* ---- colour: 8-bit sRGB -> linear -> XYB ------------------------------------------ */

void or_srgb_lut(float* lut256) {
    for (int i = 0; i < 256; ++i) {
        double v = (double)i / 255.0;
        double l = v <= 0.04045 ? v / 12.92 : pow((v + 0.055) / 1.055, 2.4);
        lut256[i] = (float)l;
    }
}

/* interleaved RGB8 -> three linear planes (`fp32_pow`: OR_VAR_SRGB_POWF, the curve evaluated in fp32) */
static void rgb8_to_linear(const uint8_t* rgb, size_t n, float* lin /* 3*n */, int fp32_pow) {
    float lut[256];
    or_srgb_lut(lut);
    if (fp32_pow)
        for (int i = 0; i < 256; ++i) {
            const float v = (float)i / 255.0f;
            lut[i] = v <= 0.04045f ? v / 12.92f : powf((v + 0.055f) / 1.055f, 2.4f);
        }
#pragma omp parallel for schedule(static)
    for (ptrdiff_t i = 0; i < (ptrdiff_t)n; ++i) {
        lin[i] = lut[rgb[3 * i]];
        lin[n + i] = lut[rgb[3 * i + 1]];
        lin[2 * n + i] = lut[rgb[3 * i + 2]];
    }
}

/* Cube root from IEEE mul/fma only (no libm, no division), so that a GPU evaluating the
   same sequence returns the same bits: bit-trick seed for y = x^(-1/3) (3 % off), one
   third-order step y <- y (1 + e/3 + 2e^2/9 + 14e^3/81), e = 1 - x y^3 (the series of
   (1-e)^(-1/3)), c = x y^2, one Newton step on c with the residual from one fma.  17
   operations.  Measured against cbrt() in fp64 over [0.0037, 1.01] and 1e-30..1e30: max error
   0.76 ulp, 91 % correctly rounded (tests/test_oracle.py) -- the same class as libm's cbrtf.
   Why it matters: the SSIM map cancels (sigma terms ~1e-5 out of values ~0.25), so a 1-ulp
   difference here moves the score by ~1e-3 (DESIGN.md "Arithmetic contract"). */
float or_cbrtf(float x) {
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

/* linear RGB planes -> "positive XYB" planes (ToXYB then MakePositiveXYB).  Operation
   order (which products are fused) is fixed here and mirrored by the HIP kernels. */
static void linear_to_xyb_impl(const float* lin, size_t n, float* xyb, int libm);
void or_linear_to_xyb(const float* lin, size_t n, float* xyb) { linear_to_xyb_impl(lin, n, xyb, 0); }
static void linear_to_xyb_impl(const float* lin, size_t n, float* xyb, int libm) {
    const float cb = libm ? cbrtf(kOpsinBias) : or_cbrtf(kOpsinBias);
#pragma omp parallel for schedule(static)
    for (ptrdiff_t i = 0; i < (ptrdiff_t)n; ++i) {
        const float r = lin[i], g = lin[n + i], b = lin[2 * n + i];
        float l = fmaf(kM00, r, fmaf(kM01, g, fmaf(kM02, b, kOpsinBias)));
        float m = fmaf(kM10, r, fmaf(kM11, g, fmaf(kM12, b, kOpsinBias)));
        float s = fmaf(kM20, r, fmaf(kM21, g, fmaf(kM22, b, kOpsinBias)));
        l = l < 0.0f ? 0.0f : l;
        m = m < 0.0f ? 0.0f : m;
        s = s < 0.0f ? 0.0f : s;
        l = (libm ? cbrtf(l) : or_cbrtf(l)) - cb; /* libm: OR_VAR_CBRT_LIBM */
        m = (libm ? cbrtf(m) : or_cbrtf(m)) - cb;
        s = (libm ? cbrtf(s) : or_cbrtf(s)) - cb;
        const float X = 0.5f * (l - m), Y = 0.5f * (l + m), B = s;
        xyb[2 * n + i] = (B - Y) + 0.55f;
        xyb[i] = fmaf(X, 14.0f, 0.42f);
        xyb[n + i] = Y + 0.01f;
    }
}

/* 2x2 box average of linear planes, edge pixels replicated (Downsample(in,2,2)) */
static void downsample2_impl(const float* in, size_t w, size_t h, float* out, int floor_dims);
void or_downsample2(const float* in, size_t w, size_t h, float* out) { downsample2_impl(in, w, h, out, 0); }
/* `floor_dims`: OR_VAR_DOWNSAMPLE_FLOOR, an odd last row / column is dropped instead of replicated */
static void downsample2_impl(const float* in, size_t w, size_t h, float* out, int floor_dims) {
    const size_t ow = floor_dims && w > 1 ? w / 2 : (w + 1) / 2, oh = floor_dims && h > 1 ? h / 2 : (h + 1) / 2;
    for (int c = 0; c < 3; ++c) {
        const float* pin = in + (size_t)c * w * h;
        float* pout = out + (size_t)c * ow * oh;
#pragma omp parallel for schedule(static)
        for (ptrdiff_t oy = 0; oy < (ptrdiff_t)oh; ++oy) {
            for (size_t ox = 0; ox < ow; ++ox) {
                float sum = 0.0f;
                for (size_t iy = 0; iy < 2; ++iy)
                    for (size_t ix = 0; ix < 2; ++ix) {
                        size_t x = ox * 2 + ix, y = (size_t)oy * 2 + iy;
                        if (x > w - 1) x = w - 1;
                        if (y > h - 1) y = h - 1;
                        sum += pin[y * w + x];
                    }
                pout[oy * ow + ox] = sum * 0.25f;
            }
        }
    }
}

/* ---- maps ----------------------------------------------------------------------- */

static double tothe4th(double x) { x *= x; x *= x; return x; }

static void ssim_map(const float* m1, const float* m2, const float* s11, const float* s22,
                     const float* s12, size_t n, double* plane_avg /* 6 */) {
    const double one_per_pixels = 1.0 / (double)n;
    for (int c = 0; c < 3; ++c) {
        double sum0 = 0.0, sum1 = 0.0;
        const float *a = m1 + c * n, *b = m2 + c * n, *p11 = s11 + c * n, *p22 = s22 + c * n,
                    *p12 = s12 + c * n;
#pragma omp parallel for schedule(static) reduction(+ : sum0, sum1)
        for (ptrdiff_t i = 0; i < (ptrdiff_t)n; ++i) {
            const float mu1 = a[i], mu2 = b[i];
            const float mu11 = mu1 * mu1, mu22 = mu2 * mu2, mu12 = mu1 * mu2;
            const float dm = mu1 - mu2;
            const float num_m = fmaf(-dm, dm, 1.0f);
            const float num_s = fmaf(2.0f, p12[i] - mu12, kC2);
            const float denom_s = ((p11[i] - mu11) + (p22[i] - mu22)) + kC2;
            double d = 1.0 - (double)((num_m * num_s) / denom_s);
            d = d > 0.0 ? d : 0.0;
            sum0 += d;
            sum1 += tothe4th(d);
        }
        plane_avg[c * 2] = one_per_pixels * sum0;
        plane_avg[c * 2 + 1] = sqrt(sqrt(one_per_pixels * sum1));
    }
}

static void edge_diff_map(const float* img1, const float* mu1, const float* img2,
                          const float* mu2, size_t n, double* plane_avg /* 12 */) {
    const double one_per_pixels = 1.0 / (double)n;
    for (int c = 0; c < 3; ++c) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        const float *r1 = img1 + c * n, *rm1 = mu1 + c * n, *r2 = img2 + c * n,
                    *rm2 = mu2 + c * n;
#pragma omp parallel for schedule(static) reduction(+ : s0, s1, s2, s3)
        for (ptrdiff_t i = 0; i < (ptrdiff_t)n; ++i) {
            const double d1 = (1.0 + fabs((double)(r2[i] - rm2[i]))) /
                                  (1.0 + fabs((double)(r1[i] - rm1[i]))) -
                              1.0;
            const double artifact = d1 > 0.0 ? d1 : 0.0;
            s0 += artifact;
            s1 += tothe4th(artifact);
            const double detail_lost = d1 < 0.0 ? -d1 : 0.0;
            s2 += detail_lost;
            s3 += tothe4th(detail_lost);
        }
        plane_avg[c * 4] = one_per_pixels * s0;
        plane_avg[c * 4 + 1] = sqrt(sqrt(one_per_pixels * s1));
        plane_avg[c * 4 + 2] = one_per_pixels * s2;
        plane_avg[c * 4 + 3] = sqrt(sqrt(one_per_pixels * s3));
    }
}

/* OR_VAR_SUMS_F32: both maps and their running sums in fp32, pixel order, one thread (deterministic) */
static void maps_f32(const float* img1, const float* mu1, const float* img2, const float* mu2, const float* s11,
                     const float* s22, const float* s12, size_t n, double* plane_avg /* 18 */) {
    const float inv = 1.0f / (float)n;
    for (int c = 0; c < 3; ++c) {
        float sum0 = 0.f, sum1 = 0.f, e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
        const float *a = mu1 + c * n, *b = mu2 + c * n, *p11 = s11 + c * n, *p22 = s22 + c * n, *p12 = s12 + c * n;
        const float *r1 = img1 + c * n, *r2 = img2 + c * n;
        for (size_t i = 0; i < n; ++i) {
            const float m1 = a[i], m2 = b[i];
            const float dm = m1 - m2;
            const float num_m = 1.0f - dm * dm;
            const float num_s = 2.0f * (p12[i] - m1 * m2) + kC2;
            const float denom_s = (p11[i] - m1 * m1) + (p22[i] - m2 * m2) + kC2;
            float d = 1.0f - (num_m * num_s) / denom_s;
            d = d > 0.0f ? d : 0.0f;
            sum0 += d;
            sum1 += (d * d) * (d * d);
            const float d1 = (1.0f + fabsf(r2[i] - m2)) / (1.0f + fabsf(r1[i] - m1)) - 1.0f;
            const float art = d1 > 0.0f ? d1 : 0.0f, det = d1 < 0.0f ? -d1 : 0.0f;
            e0 += art;
            e1 += (art * art) * (art * art);
            e2 += det;
            e3 += (det * det) * (det * det);
        }
        plane_avg[c * 2] = (double)(inv * sum0);
        plane_avg[c * 2 + 1] = (double)sqrtf(sqrtf(inv * sum1));
        plane_avg[6 + c * 4] = (double)(inv * e0);
        plane_avg[6 + c * 4 + 1] = (double)sqrtf(sqrtf(inv * e1));
        plane_avg[6 + c * 4 + 2] = (double)(inv * e2);
        plane_avg[6 + c * 4 + 3] = (double)sqrtf(sqrtf(inv * e3));
    }
}

/* 108 averages -> score.  avg layout: [scale][18] = 6 ssim (c*2+n) then 12 edge (c*4+k). */
double or_score_from_averages(const double* avg /* nscales*18 */, int nscales) {
    /* The published Score() walks `for c: for scale < scales.size(): for n` with a
       running weight index, so an image too small for all six scales consumes the
       weights contiguously (no gaps for the missing scales).  Kept as published. */
    double ssim = 0.0;
    size_t i = 0;
    for (int c = 0; c < 3; ++c) {
        for (int scale = 0; scale < nscales; ++scale) {
            const double* a = avg + scale * 18;
            for (int n = 0; n < 2; ++n) {
                ssim += kWeights[i++] * fabs(a[c * 2 + n]);
                ssim += kWeights[i++] * fabs(a[6 + c * 4 + n]);
                ssim += kWeights[i++] * fabs(a[6 + c * 4 + n + 2]);
            }
        }
    }
    ssim = ssim * 0.9562382616834844;
    ssim = 2.326765642916932 * ssim - 0.020884521182843837 * ssim * ssim +
           6.248496625763138e-05 * ssim * ssim * ssim;
    if (ssim > 0.0) ssim = 100.0 - 10.0 * pow(ssim, 0.6276336467831387);
    else ssim = 100.0;
    return ssim;
}

/* /root/reference/src/io.zig:654-663 -- decodeAvifToRgb's copy of libavif's decoded rows
   (RGB or RGBA, `row_bytes` apart) into the tight RGB8 buffer handed to the scorer.  Scalar and
   single-threaded like the reference's loop; the checker (and CPU timing beside) the device
   unpack behind ssimu2_score_against_reference_strided. */
void or_copy_rgb_pixels(const uint8_t* src, size_t row_bytes, int src_channels, int w, int h,
                        uint8_t* dst) {
    for (int y = 0; y < h; ++y) {
        const uint8_t* src_row = src + (size_t)y * row_bytes;
        for (int x = 0; x < w; ++x) {
            const size_t si = (size_t)x * (size_t)src_channels;
            const size_t di = ((size_t)y * (size_t)w + (size_t)x) * 3;
            dst[di + 0] = src_row[si + 0];
            dst[di + 1] = src_row[si + 1];
            dst[di + 2] = src_row[si + 2];
        }
    }
}

/* thread count of the OpenMP build (the environment variable is read only once per process,
   and another library may have initialised the OpenMP runtime first) */
int or_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

void or_weights(double* out108) { memcpy(out108, kWeights, sizeof(kWeights)); }

/* ---- the entry point: same contract as the call at tq.zig:37 ---------------------- */
/* returns 0 on success; *out_score is the SSIMULACRA2 score; avg_out (optional) gets
   6*18 doubles (unused scales zero), nscales_out (optional) the scales evaluated. */
static int compute_core(const uint8_t* ref, const uint8_t* dist, uint32_t w, uint32_t h, uint32_t channels,
                        int blur_mode, unsigned variant, double* out_score, double* avg_out, int* nscales_out);

int or_compute_ssimu2(const uint8_t* ref, const uint8_t* dist, uint32_t w, uint32_t h,
                      uint32_t channels, int blur_mode, double* out_score, double* avg_out,
                      int* nscales_out) {
    return compute_core(ref, dist, w, h, channels, blur_mode, 0u, out_score, avg_out, nscales_out);
}

/* The pin kit's entry point: the score with the stages named in `variant` (OR_VAR_* bits) switched to their
   alternatives.  variant = 0 is or_compute_ssimu2.  Blur variants need an FIR-family `blur_mode` (-5 otherwise). */
int or_compute_ssimu2_variant(const uint8_t* ref, const uint8_t* dist, uint32_t w, uint32_t h, int blur_mode,
                              unsigned variant, double* out_score, double* avg_out, int* nscales_out) {
    if (variant & ~OR_VAR_ALL) return -5;
    if ((variant & OR_VAR_BLUR_MASK) && blur_mode != OR_BLUR_FIR && blur_mode != OR_BLUR_FIR_PRODFIRST) return -5;
    if ((variant & OR_VAR_EDGE_CLAMP) && (variant & OR_VAR_EDGE_MIRROR)) return -5;
    if ((variant & OR_VAR_GAUSS9) && (variant & OR_VAR_GAUSS11)) return -5;
    return compute_core(ref, dist, w, h, 3, blur_mode, variant, out_score, avg_out, nscales_out);
}

static int compute_core(const uint8_t* ref, const uint8_t* dist, uint32_t w, uint32_t h, uint32_t channels,
                        int blur_mode, unsigned variant, double* out_score, double* avg_out, int* nscales_out) {
    if (!ref || !dist || !out_score) return -1;
    if (channels != 3) return -2;
    if (w == 0 || h == 0) return -3;
    or_gauss rg;
    or_gauss_create(1.5, &rg);
    size_t n = (size_t)w * h;
    float* lin1 = (float*)malloc(sizeof(float) * 3 * n);
    float* lin2 = (float*)malloc(sizeof(float) * 3 * n);
    float* img1 = (float*)malloc(sizeof(float) * 3 * n);
    float* img2 = (float*)malloc(sizeof(float) * 3 * n);
    float* mul = (float*)malloc(sizeof(float) * 3 * n);
    float* s11 = (float*)malloc(sizeof(float) * 3 * n);
    float* s22 = (float*)malloc(sizeof(float) * 3 * n);
    float* s12 = (float*)malloc(sizeof(float) * 3 * n);
    float* mu1 = (float*)malloc(sizeof(float) * 3 * n);
    float* mu2 = (float*)malloc(sizeof(float) * 3 * n);
    float* tmp = (float*)malloc(sizeof(float) * n);
    float* dtmp = (float*)malloc(sizeof(float) * 3 * ((n + 3) / 4 + w + h + 4));
    if (!lin1 || !lin2 || !img1 || !img2 || !mul || !s11 || !s22 || !s12 || !mu1 || !mu2 ||
        !tmp || !dtmp) {
        free(lin1); free(lin2); free(img1); free(img2); free(mul); free(s11); free(s22);
        free(s12); free(mu1); free(mu2); free(tmp); free(dtmp);
        return -4;
    }
    rgb8_to_linear(ref, n, lin1, (variant & OR_VAR_SRGB_POWF) != 0);
    rgb8_to_linear(dist, n, lin2, (variant & OR_VAR_SRGB_POWF) != 0);
    or_conv cv;
    conv_setup(&rg, variant, &cv);
    const int libm_cbrt = (variant & OR_VAR_CBRT_LIBM) != 0, floor_dims = (variant & OR_VAR_DOWNSAMPLE_FLOOR) != 0;

    double avg[OR_NUM_SCALES * 18];
    memset(avg, 0, sizeof(avg));
    int nscales = 0;
    size_t cw = w, ch = h;
    for (int scale = 0; scale < OR_NUM_SCALES; ++scale) {
        if (cw < 8 || ch < 8) break;
        if (scale) {
            const size_t ow = floor_dims && cw > 1 ? cw / 2 : (cw + 1) / 2, oh = floor_dims && ch > 1 ? ch / 2 : (ch + 1) / 2;
            if ((variant & OR_VAR_SIZE_TEST_AFTER) && (ow < 8 || oh < 8)) break;
            if (variant & OR_VAR_DOWNSAMPLE_XYB) { /* the previous scale's XYB planes averaged, no re-conversion */
                downsample2_impl(img1, cw, ch, dtmp, floor_dims);
                memcpy(img1, dtmp, sizeof(float) * 3 * ow * oh);
                downsample2_impl(img2, cw, ch, dtmp, floor_dims);
                memcpy(img2, dtmp, sizeof(float) * 3 * ow * oh);
            } else {
                downsample2_impl(lin1, cw, ch, dtmp, floor_dims);
                memcpy(lin1, dtmp, sizeof(float) * 3 * ow * oh);
                downsample2_impl(lin2, cw, ch, dtmp, floor_dims);
                memcpy(lin2, dtmp, sizeof(float) * 3 * ow * oh);
            }
            cw = ow;
            ch = oh;
            /* libjxl tests the size BEFORE downsampling for this scale; the downsampled
               image is scored even if it is now smaller than 8. */
        }
        n = cw * ch;
        if (!scale || !(variant & OR_VAR_DOWNSAMPLE_XYB)) {
            linear_to_xyb_impl(lin1, n, img1, libm_cbrt);
            linear_to_xyb_impl(lin2, n, img2, libm_cbrt);
        }
        for (int c = 0; c < 3; ++c) {
            const float *a = img1 + c * n, *b = img2 + c * n;
            float* m = mul + c * n;
            if (variant & OR_VAR_BLUR_MASK) { /* generic convolution: products rounded first, then blurred */
                const float* pl[3][2] = {{a, a}, {b, b}, {a, b}};
                float* outp[3] = {s11 + c * n, s22 + c * n, s12 + c * n};
                for (int k = 0; k < 3; ++k) {
                    for (size_t i = 0; i < n; ++i) m[i] = pl[k][0][i] * pl[k][1][i];
                    conv_plane(&cv, m, cw, ch, tmp, outp[k]);
                }
                conv_plane(&cv, a, cw, ch, tmp, mu1 + c * n);
                conv_plane(&cv, b, cw, ch, tmp, mu2 + c * n);
                continue;
            }
            if (blur_mode == OR_BLUR_EXACT) {
                blur_plane_exact(&rg, a, a, cw, ch, s11 + c * n);
                blur_plane_exact(&rg, b, b, cw, ch, s22 + c * n);
                blur_plane_exact(&rg, a, b, cw, ch, s12 + c * n);
                blur_plane_exact(&rg, a, NULL, cw, ch, mu1 + c * n);
                blur_plane_exact(&rg, b, NULL, cw, ch, mu2 + c * n);
                continue;
            }
            blur_plane_prod(&rg, blur_mode, a, a, cw, ch, m, tmp, s11 + c * n);
            blur_plane_prod(&rg, blur_mode, b, b, cw, ch, m, tmp, s22 + c * n);
            blur_plane_prod(&rg, blur_mode, a, b, cw, ch, m, tmp, s12 + c * n);
            const int plain_mode = blur_mode == OR_BLUR_FIR_PRODFIRST ? OR_BLUR_FIR : blur_mode;
            blur_plane(&rg, plain_mode, a, cw, ch, tmp, mu1 + c * n);
            blur_plane(&rg, plain_mode, b, cw, ch, tmp, mu2 + c * n);
        }
        if (variant & OR_VAR_SUMS_F32) {
            maps_f32(img1, mu1, img2, mu2, s11, s22, s12, n, avg + scale * 18);
        } else {
            ssim_map(mu1, mu2, s11, s22, s12, n, avg + scale * 18);
            edge_diff_map(img1, mu1, img2, mu2, n, avg + scale * 18 + 6);
        }
        ++nscales;
    }
    *out_score = or_score_from_averages(avg, nscales);
    if (avg_out) memcpy(avg_out, avg, sizeof(avg));
    if (nscales_out) *nscales_out = nscales;
    free(lin1); free(lin2); free(img1); free(img2); free(mul); free(s11); free(s22);
    free(s12); free(mu1); free(mu2); free(tmp); free(dtmp);
    return 0;
}
