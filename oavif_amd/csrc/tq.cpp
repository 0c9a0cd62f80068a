// Target-quality search: host-side fp64 control logic behind include/oavif_tq.h.
//
// Behavioural restatement of /root/reference/src/tq.zig:40-210 (the reference's Zig is the
// specification; every decision that can change the chosen quantizer is cited).  The pass
// itself (encode -> decode -> score, tq.zig:21-38) is injected, so the same code runs with
// a scripted score table (tests), a CPU codec + the HIP scorer (product), or anything else.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/oavif_tq.h"

namespace {

// Zig's @round rounds half away from zero == C round() (tq.zig:42,109,114,116).
inline double zig_round(double v) { return std::round(v); }

inline uint32_t clamp_round_0_100(double r) {
    // @intFromFloat(std.math.clamp(@round(r), 0, 100))   tq.zig:109,114,116
    double v = zig_round(r);
    if (!(v >= 0.0)) v = 0.0;  // also maps NaN to 0 instead of Zig's undefined behaviour
    if (v > 100.0) v = 100.0;
    return (uint32_t)v;
}

// tq.zig:45-51: the line through points [0] and [1] of the score-sorted history.
bool linear_interpolate(const std::vector<double>& s, const std::vector<double>& q, double target,
                        double* out) {
    if (s.size() < 2) return false;
    if (s[1] == s[0]) return false;
    const double t = (target - s[0]) / (s[1] - s[0]);
    *out = q[0] + (q[1] - q[0]) * t;
    return true;
}

// tq.zig:53-71: parabola q(score) through points [0],[1],[2] of the score-sorted history.
bool quadratic_interpolate(const std::vector<double>& s, const std::vector<double>& q,
                           double target, double* out) {
    if (s.size() < 3) return false;
    const double x0 = s[0], x1 = s[1], x2 = s[2];
    const double y0 = q[0], y1 = q[1], y2 = q[2];
    const double denom = (x0 - x1) * (x0 - x2) * (x1 - x2);
    if (std::fabs(denom) < 0.001) return false;
    const double a = (x2 * (y1 - y0) + x1 * (y0 - y2) + x0 * (y2 - y1)) / denom;
    const double b = (x2 * x2 * (y0 - y1) + x1 * x1 * (y2 - y0) + x0 * x0 * (y1 - y2)) / denom;
    const double c =
        (x1 * x2 * (x1 - x2) * y0 + x2 * x0 * (x2 - x0) * y1 + x0 * x1 * (x0 - x1) * y2) / denom;
    *out = a * target * target + b * target + c;
    return true;
}

uint32_t interpolate_quantizer(uint32_t lo, uint32_t hi, const oavif_tq_pass* hist, uint32_t n,
                               double target) {
    const uint32_t binary = (lo + hi) / 2;  // @divFloor on unsigned     tq.zig:80
    if (n == 0) return binary;              //                          tq.zig:82-83
    // copy + stable sort ascending by score (std.mem.sort is stable)   tq.zig:85-93
    std::vector<oavif_tq_pass> sorted(hist, hist + n);
    std::stable_sort(sorted.begin(), sorted.end(),
                     [](const oavif_tq_pass& l, const oavif_tq_pass& r) { return l.score < r.score; });
    std::vector<double> scores(n), quals(n);
    for (uint32_t i = 0; i < n; ++i) {
        scores[i] = sorted[i].score;
        quals[i] = (double)sorted[i].q;
    }
    uint32_t pred = binary;
    double r;
    if (n == 1) {
        pred = binary;                                               // tq.zig:106
    } else if (n == 2) {
        if (linear_interpolate(scores, quals, target, &r)) pred = clamp_round_0_100(r);
    } else {
        // the interpolants use the LOWEST-scoring two / three probes    tq.zig:112-118
        if (quadratic_interpolate(scores, quals, target, &r)) pred = clamp_round_0_100(r);
        else if (linear_interpolate(scores, quals, target, &r)) pred = clamp_round_0_100(r);
    }
    // std.math.clamp(pred, lo, hi)                                     tq.zig:121
    if (pred < lo) pred = lo;
    if (pred > hi) pred = hi;
    return pred;
}

}  // namespace

extern "C" {

void oavif_tq_default_options(oavif_tq_options* o) {
    if (!o) return;
    o->score_tgt = 80.0;  // parse_args.zig:55
    o->tolerance = 2.0;   // parse_args.zig:58
    o->max_pass = 6;      // parse_args.zig:59
}

uint32_t oavif_tq_predict_q_from_score(double tgt) {
    const double q = 6.83 * std::exp(0.0282 * tgt);         // tq.zig:41
    return (uint32_t)std::fmin(100.0, zig_round(q));        // tq.zig:42
}

uint32_t oavif_tq_interpolate_quantizer(uint32_t lo_bound, uint32_t hi_bound,
                                        const oavif_tq_pass* history, uint32_t history_len,
                                        double target) {
    return interpolate_quantizer(lo_bound, hi_bound, history, history_len, target);
}

int oavif_tq_find_target_quality(const oavif_tq_options* o, oavif_tq_probe_fn probe, void* user,
                                 oavif_tq_result* out) {
    if (!o || !probe || !out) return SSIMU2_ERR_INVALID_ARG;
    if (o->max_pass < 1 || o->max_pass > OAVIF_TQ_MAX_PASS) return SSIMU2_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    out->buf_q = -1;

    oavif_tq_pass* hist = out->history;
    uint32_t n = 0;
    uint32_t lo = 0, hi = 100;  // tq.zig:132-133
    uint32_t q = 0;
    double score = 0.0;

    for (uint32_t pass = 0; pass < o->max_pass; ++pass) {
        q = pass == 0 ? oavif_tq_predict_q_from_score(o->score_tgt)
                      : interpolate_quantizer(lo, hi, hist, n, o->score_tgt);  // tq.zig:136-139

        // quantizer already probed: stop before scoring                  tq.zig:141-148
        bool seen = false;
        for (uint32_t i = 0; i < n; ++i) seen |= hist[i].q == q;
        if (seen) break;

        const int rc = probe(user, q, &score);  // computeScoreAtQuality  tq.zig:150
        if (rc != 0) return rc;
        out->num_pass += 1;                     // tq.zig:29
        out->buf_q = (int32_t)q;                // tq.zig:34
        hist[n].q = q;
        hist[n].score = score;
        ++n;
        out->history_len = n;

        const double abs_err = std::fabs(score - o->score_tgt);
        if (pass == 0) {  // bound the range from the first error        tq.zig:155-165
            const uint32_t err_bound = (uint32_t)(std::ceil(abs_err) * 4.0);
            if (score - o->score_tgt > 0) {
                hi = q;
                lo = q > err_bound ? q - err_bound : 0;
            } else {  // note: score == target lands here
                lo = q;
                hi = std::min<uint32_t>(100u, q + err_bound);
            }
        }

        if (abs_err < o->tolerance) {  // accepted as is, above OR below target   tq.zig:167-168
            out->q = q;
            out->score = score;
            return SSIMU2_OK;
        }

        if (pass > 0) {  // tq.zig:171-176
            if (score > o->score_tgt) hi = q;
            else lo = q;
        }

        // u32 arithmetic: hi == 0 wraps, as the ReleaseFast build the reference's CI ships
        // (.github/workflows/ci.yml:51-52) does                           tq.zig:179
        if (lo >= hi - 1u) break;
    }

    // final pick                                                          tq.zig:183-209
    bool have_best = false;
    uint32_t best_q = 0, highest_q = 0;
    double best_score = 0.0, highest_score = 0.0;
    for (uint32_t i = 0; i < n; ++i) {
        const oavif_tq_pass& h = hist[i];
        if (h.score >= o->score_tgt && (!have_best || h.q < best_q)) {
            have_best = true;
            best_q = h.q;
            best_score = h.score;
        }
        // compares max(score, 0) but stores the raw score (kept asymmetric) tq.zig:193-196
        if (std::fmax(h.score, 0.0) >= highest_score) {
            highest_score = h.score;
            highest_q = h.q;
        }
    }
    if (have_best) {
        out->q = best_q;
        out->score = best_score;
    } else {
        out->q = highest_q;
        out->score = highest_score;
    }
    return SSIMU2_OK;
}

namespace {

// ---- speculative probe fan-out -------------------------------------------------------------
struct Spec {
    const oavif_tq_options* o;
    uint32_t fanout;        // probes per wave from the second wave on
    uint32_t first_fanout;  // ... of the first wave
    oavif_tq_batch_probe_fn batch;
    void* user;
    bool known[101];
    double score[101];
    oavif_tq_spec_stats stats;
};

constexpr int kSimStop = 0x7157;  // private return code: the simulated search asked for a new q

struct Sim {
    const Spec* s;
    uint32_t q_miss;
    double hyp;
    int32_t next;
};

static int sim_probe(void* p, uint32_t q, double* out_score) {
    Sim* m = (Sim*)p;
    if (q <= 100 && m->s->known[q]) {
        *out_score = m->s->score[q];
        return 0;
    }
    if (q == m->q_miss) {
        *out_score = m->hyp;
        return 0;
    }
    m->next = (int32_t)q;
    return kSimStop;
}

// Score the search may see at `q`: between two probed quantizers, the line through them;
// otherwise the model behind the first guess (tq.zig:41 inverted: score = ln(q / 6.83) / 0.0282)
// shifted to pass through the nearest probe.
static double estimate_score(const Spec& s, uint32_t q) {
    int below = -1, above = -1;
    for (int k = (int)q - 1; k >= 0 && below < 0; --k)
        if (s.known[k]) below = k;
    for (int k = (int)q + 1; k <= 100 && above < 0; ++k)
        if (s.known[k]) above = k;
    if (below >= 0 && above >= 0) {
        const double t = ((double)q - below) / ((double)above - below);
        return s.score[below] + (s.score[above] - s.score[below]) * t;
    }
    auto model = [](double qq) { return std::log(std::fmax(qq, 1.0) / 6.83) / 0.0282; };
    const int near = below >= 0 ? below : above;
    if (near < 0) return model((double)q);
    return model((double)q) + (s.score[near] - model((double)near));
}

// Candidates for the pass after `q_miss`: what the search would ask for next if q_miss scored
// est +- (tolerance + 0.5 + k), k = 0, 1, 2, ...  (inside the tolerance the search ends).
static void add_candidates(const Spec& s, uint32_t q_miss, uint32_t* wave, uint32_t* n, uint32_t fanout) {
    const double est = estimate_score(s, q_miss);
    for (int k = 0; k < 24 && *n < fanout; ++k) {
        for (int sign = +1; sign >= -1 && *n < fanout; sign -= 2) {
            Sim m{&s, q_miss, est + sign * (s.o->tolerance + 0.5 + k), -1};
            // hypothetical scores relative to the TARGET as well: the pass-0 bounds depend on
            // |score - target| only, and the estimate may be far off for unusual content
            for (int rel = 0; rel < 2 && *n < fanout; ++rel) {
                if (rel == 1) m.hyp = s.o->score_tgt + sign * (s.o->tolerance + 0.5 + k);
                m.next = -1;
                oavif_tq_result tmp;
                const int rc = oavif_tq_find_target_quality(s.o, sim_probe, &m, &tmp);
                if (rc != kSimStop || m.next < 0 || m.next > 100) continue;
                bool dup = false;
                for (uint32_t i = 0; i < *n; ++i) dup |= wave[i] == (uint32_t)m.next;
                if (!dup) wave[(*n)++] = (uint32_t)m.next;
            }
        }
    }
}

static int replay_probe(void* p, uint32_t q, double* out_score) {
    Spec* s = (Spec*)p;
    if (q > 100) return SSIMU2_ERR_INVALID_ARG;  // unreachable: every proposal is clamped to 0..100
    if (s->known[q]) {
        s->stats.cache_hits += 1;
        *out_score = s->score[q];
        return 0;
    }
    uint32_t wave[OAVIF_TQ_MAX_FANOUT];
    double sc[OAVIF_TQ_MAX_FANOUT];
    uint32_t n = 0;
    wave[n++] = q;
    const uint32_t fanout = s->stats.waves == 0 ? s->first_fanout : s->fanout;
    if (fanout > 1) add_candidates(*s, q, wave, &n, fanout);
    const int rc = s->batch(s->user, wave, n, sc);
    if (rc != 0) return rc;
    s->stats.waves += 1;
    s->stats.probes_issued += n;
    for (uint32_t i = 0; i < n; ++i) {
        s->known[wave[i]] = true;
        s->score[wave[i]] = sc[i];
    }
    *out_score = s->score[q];
    return 0;
}

struct HipPass {
    ssimu2_ctx* scorer;
    oavif_tq_codec_fn codec;
    void* user;
    uint8_t* decoded;
    size_t last_size;
};

static int hip_probe(void* p, uint32_t q, double* out_score) {
    HipPass* s = (HipPass*)p;
    size_t sz = 0;
    const int rc = s->codec(s->user, q, s->decoded, &sz);  // tq.zig:24,26 (CPU, unchanged)
    if (rc != 0) return rc;
    s->last_size = sz;                                     // tq.zig:35
    return ssimu2_score_against_reference(s->scorer, s->decoded, out_score);  // tq.zig:37
}
}  // namespace

int oavif_tq_find_target_quality_speculative(const oavif_tq_options* o,
                                             const oavif_tq_spec_options* so,
                                             oavif_tq_batch_probe_fn batch, void* user,
                                             oavif_tq_result* out, oavif_tq_spec_stats* stats) {
    if (!o || !so || !batch || !out) return SSIMU2_ERR_INVALID_ARG;
    // ABI guard (include/oavif_tq.h): exactly this header's tag | size.  Round 2's untagged {max_fanout,
    // first_wave_fanout} cannot produce it, so such a caller gets an error instead of a misparsed fan-out
    // ({8, 1} used to read as "size 8, max_fanout 1" -- a silent sequential search).
    if (so->struct_size != (OAVIF_TQ_SPEC_OPTIONS_TAG | (uint32_t)sizeof(oavif_tq_spec_options)))
        return SSIMU2_ERR_INVALID_ARG;
    const uint32_t first_wave = so->first_wave_fanout;
    if (so->max_fanout < 1 || so->max_fanout > OAVIF_TQ_MAX_FANOUT) return SSIMU2_ERR_INVALID_ARG;
    if (first_wave > so->max_fanout) return SSIMU2_ERR_INVALID_ARG;
    Spec s{};
    s.o = o;
    s.fanout = so->max_fanout;
    s.first_fanout = first_wave ? first_wave : so->max_fanout;
    s.batch = batch;
    s.user = user;
    const int rc = oavif_tq_find_target_quality(o, replay_probe, &s, out);
    if (stats) *stats = s.stats;
    return rc;
}

int oavif_tq_search_hip(const oavif_tq_options* o, ssimu2_ctx* scorer, const uint8_t* ref_rgb,
                        uint32_t w, uint32_t h, oavif_tq_codec_fn codec, void* user,
                        oavif_tq_result* out, size_t* out_last_avif_size) {
    if (!o || !scorer || !ref_rgb || !codec || !out || w == 0 || h == 0)
        return SSIMU2_ERR_INVALID_ARG;
    int rc = ssimu2_set_reference(scorer, ref_rgb, w, h);  // e.rgb is fixed for the search
    if (rc != 0) return rc;
    std::vector<uint8_t> decoded;
    try {
        decoded.resize((size_t)w * h * 3);
    } catch (const std::bad_alloc&) {
        return SSIMU2_ERR_OOM;
    }
    HipPass pass{scorer, codec, user, decoded.data(), 0};
    rc = oavif_tq_find_target_quality(o, hip_probe, &pass, out);
    if (out_last_avif_size) *out_last_avif_size = pass.last_size;
    return rc;
}

// io.zig:566-617 as functions of the source alone (see the header): computed once per search.
void oavif_prescale_8_to_10(const uint8_t* src, size_t n, uint16_t* dst) {
    uint16_t lut[256];
    for (unsigned v = 0; v < 256; ++v) lut[v] = (uint16_t)((v * 1023u + 127u) / 255u);  // io.zig:572
    for (size_t i = 0; i < n; ++i) dst[i] = lut[src[i]];
}

void oavif_prescale_16_to_10(const uint16_t* src, size_t n, uint16_t* dst) {
    for (size_t i = 0; i < n; ++i) dst[i] = (uint16_t)(src[i] >> 6);  // io.zig:587
}

void oavif_prescale_16_to_8(const uint16_t* src, size_t n, uint8_t* dst) {
    for (size_t i = 0; i < n; ++i) dst[i] = (uint8_t)(src[i] >> 8);  // io.zig:602
}

}  // extern "C"
