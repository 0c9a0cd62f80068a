// Sanitizer harness for the host-side search code (oavif_amd/csrc/tq.cpp), CPU only.
// Built by tests/test_sanitizers.py with g++ -fsanitize=address,undefined (and once more with
// -fsanitize=thread): tq.cpp + this file, the two scorer entry points tq.cpp calls replaced by
// the deterministic stand-ins below (no HIP, no GPU).  Exit code 0 = every check held and no
// sanitizer report was printed (the build uses -fno-sanitize-recover).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "oavif_tq.h"
#include "ssimu2_hip.h"

// ---- stand-ins for the scorer half of a pass (ssimu2_hip.h) ---------------------------------
struct ssimu2_ctx {
    std::vector<uint8_t> ref;
    uint32_t w = 0, h = 0;
};

extern "C" int ssimu2_set_reference(ssimu2_ctx* c, const uint8_t* ref, uint32_t w, uint32_t h) {
    if (!c || !ref) return SSIMU2_ERR_INVALID_ARG;
    c->ref.assign(ref, ref + (size_t)w * h * 3);
    c->w = w;
    c->h = h;
    return SSIMU2_OK;
}

extern "C" int ssimu2_score_against_reference(ssimu2_ctx* c, const uint8_t* dist, double* out) {
    if (!c || !dist || !out || c->ref.empty()) return SSIMU2_ERR_INVALID_ARG;
    double se = 0.0;
    for (size_t i = 0; i < c->ref.size(); ++i) {
        const double d = (double)c->ref[i] - (double)dist[i];
        se += d * d;
    }
    *out = 100.0 - 3.0 * std::sqrt(se / (double)c->ref.size());
    return SSIMU2_OK;
}

// ---- helpers -------------------------------------------------------------------------------
static uint64_t g_state = 88172645463325252ull;
static uint32_t rnd() {
    g_state ^= g_state << 13;
    g_state ^= g_state >> 7;
    g_state ^= g_state << 17;
    return (uint32_t)(g_state >> 11);
}
static double frand() { return (rnd() & 0xFFFFFF) / 16777216.0; }

#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            std::exit(1);                                                    \
        }                                                                    \
    } while (0)

struct Table {
    double score[101];
    int calls;
    int fail_at;  // probe number that returns an error (-1: never)
};

static int table_probe(void* user, uint32_t q, double* out) {
    Table* t = (Table*)user;
    if (q > 100) return 1234;  // the search must never ask for this
    if (t->fail_at >= 0 && t->calls == t->fail_at) return 77;
    ++t->calls;
    *out = t->score[q];
    return 0;
}

static int table_batch(void* user, const uint32_t* qs, uint32_t n, double* out) {
    Table* t = (Table*)user;
    std::vector<std::thread> th;  // the probes of a wave run concurrently in the product
    std::vector<int> rc(n, 0);
    for (uint32_t i = 0; i < n; ++i)
        th.emplace_back([&, i] {
            if (qs[i] > 100) rc[i] = 1234; else out[i] = t->score[qs[i]];
        });
    for (auto& x : th) x.join();
    t->calls += (int)n;
    for (uint32_t i = 0; i < n; ++i)
        if (rc[i]) return rc[i];
    return 0;
}

static void fill_table(Table* t, int kind) {
    // kind 0: smooth monotone; 1: noisy; 2: flat plateaus; 3: decreasing; 4: constants / extremes
    double base = 20.0 + 40.0 * frand(), gain = 0.2 + 0.8 * frand();
    for (int q = 0; q <= 100; ++q) {
        double s = base + gain * q;
        if (kind == 1) s += 6.0 * (frand() - 0.5);
        if (kind == 2) s = std::floor(s / 7.0) * 7.0;
        if (kind == 3) s = base + gain * (100 - q);
        if (kind == 4) s = (rnd() & 1) ? 100.0 : -250.0 * frand();
        t->score[q] = s;
    }
    t->calls = 0;
    t->fail_at = -1;
}

struct Codec {
    const uint8_t* ref;
    size_t n;
    int calls;
};
static int codec(void* user, uint32_t q, uint8_t* out_rgb, size_t* out_size) {
    Codec* c = (Codec*)user;
    const int step = 1 + (int)(100 - q) / 3;
    for (size_t i = 0; i < c->n; ++i) {
        const int v = (c->ref[i] / step) * step + step / 2;
        out_rgb[i] = (uint8_t)(v > 255 ? 255 : v);
    }
    *out_size = 1000 + 10 * q;
    ++c->calls;
    return q == 13 ? 55 : 0;  // one quantizer's encode fails: the error must propagate
}

int main(int argc, char** argv) {
    // 1. the search over probe tables: every option corner, every kind of table
    long searches = 0;
    const int iters = argc > 1 ? std::atoi(argv[1]) : 1000;
    for (int iter = 0; iter < iters; ++iter) {
        Table t;
        fill_table(&t, iter % 5);
        oavif_tq_options o;
        oavif_tq_default_options(&o);
        CHECK(o.score_tgt == 80.0 && o.tolerance == 2.0 && o.max_pass == 6);
        o.score_tgt = 30.0 + 70.0 * frand();
        o.tolerance = 1.0 + (iter % 7 == 0 ? 99.0 * frand() : 4.0 * frand());
        o.max_pass = 1 + rnd() % OAVIF_TQ_MAX_PASS;
        if (iter % 97 == 0) o.score_tgt = 100.0;
        if (iter % 89 == 0) o.score_tgt = 30.0;
        oavif_tq_result r;
        std::memset(&r, 0xAB, sizeof r);
        int rc = oavif_tq_find_target_quality(&o, table_probe, &t, &r);
        CHECK(rc == 0);
        CHECK(r.q <= 100 && r.num_pass >= 1 && r.num_pass <= o.max_pass);
        CHECK(r.history_len == r.num_pass && (int)r.num_pass == t.calls);
        CHECK(r.buf_q >= 0 && (uint32_t)r.buf_q == r.history[r.history_len - 1].q);
        for (uint32_t i = 0; i < r.history_len; ++i) CHECK(r.history[i].q <= 100);
        ++searches;
        // the speculative search returns the same result for every fan-out
        for (uint32_t fan : {1u, 2u, 5u, (uint32_t)OAVIF_TQ_MAX_FANOUT}) {
            Table t2 = t;
            t2.calls = 0;
            oavif_tq_spec_options so = OAVIF_TQ_SPEC_OPTIONS_INIT(fan, (uint32_t)(iter % 2 ? 1 : 0));  // every other search: the first wave alone
            oavif_tq_result r2;
            oavif_tq_spec_stats st;
            std::memset(&r2, 0xCD, sizeof r2);
            rc = oavif_tq_find_target_quality_speculative(&o, &so, table_batch, &t2, &r2, &st);
            CHECK(rc == 0);
            CHECK(r2.q == r.q && r2.num_pass == r.num_pass && r2.buf_q == r.buf_q && r2.history_len == r.history_len);
            CHECK(std::memcmp(&r2.score, &r.score, sizeof(double)) == 0);
            for (uint32_t i = 0; i < r.history_len; ++i)
                CHECK(r2.history[i].q == r.history[i].q &&
                      std::memcmp(&r2.history[i].score, &r.history[i].score, sizeof(double)) == 0);
            CHECK(st.probes_issued == (uint32_t)t2.calls && st.waves >= 1 && st.waves <= r.num_pass);
            ++searches;
        }
        // a probe error aborts the search with the probe's code
        Table t3 = t;
        t3.calls = 0;
        t3.fail_at = (int)(rnd() % r.num_pass);
        rc = oavif_tq_find_target_quality(&o, table_probe, &t3, &r);
        CHECK(rc == 77);
    }
    // 2. argument errors
    {
        oavif_tq_options o;
        oavif_tq_default_options(&o);
        oavif_tq_result r;
        Table t;
        fill_table(&t, 0);
        CHECK(oavif_tq_find_target_quality(nullptr, table_probe, &t, &r) != 0);
        CHECK(oavif_tq_find_target_quality(&o, nullptr, &t, &r) != 0);
        CHECK(oavif_tq_find_target_quality(&o, table_probe, &t, nullptr) != 0);
        oavif_tq_spec_options so = OAVIF_TQ_SPEC_OPTIONS_INIT(0, 0);
        oavif_tq_spec_stats st;
        CHECK(oavif_tq_find_target_quality_speculative(&o, &so, table_batch, &t, &r, &st) != 0);
        so.max_fanout = OAVIF_TQ_MAX_FANOUT + 1;
        CHECK(oavif_tq_find_target_quality_speculative(&o, &so, table_batch, &t, &r, &st) != 0);
        // the ABI guard: only this header's tag | size is accepted.  A bare size (12, or the 8 an earlier
        // library took for "stops before first_wave_fanout") and round 2's untagged {max_fanout,
        // first_wave_fanout} -- whose legal fan-outs 8 and 12 used to read as sizes -- are refused
        so.max_fanout = 4;
        const uint32_t good = so.struct_size;
        CHECK(good == (OAVIF_TQ_SPEC_OPTIONS_TAG | 12u));
        CHECK(oavif_tq_find_target_quality_speculative(&o, &so, table_batch, &t, &r, &st) == 0);
        for (uint32_t bad : {4u, 8u, 12u, 64u, OAVIF_TQ_SPEC_OPTIONS_TAG | 8u, OAVIF_TQ_SPEC_OPTIONS_TAG | 16u, OAVIF_TQ_SPEC_OPTIONS_TAG}) {
            so.struct_size = bad;
            CHECK(oavif_tq_find_target_quality_speculative(&o, &so, table_batch, &t, &r, &st) != 0);
        }
        for (uint32_t old_fan = 1; old_fan <= OAVIF_TQ_MAX_FANOUT; ++old_fan)
            for (uint32_t old_first = 0; old_first <= old_fan; ++old_first) {
                const uint32_t old_layout[3] = {old_fan, old_first, 0xFFFFFFFFu};   // what a round-2 caller's memory holds
                CHECK(oavif_tq_find_target_quality_speculative(&o, (const oavif_tq_spec_options*)old_layout, table_batch, &t, &r, &st) != 0);
            }
        so.struct_size = good;
        o.max_pass = 0;
        CHECK(oavif_tq_find_target_quality(&o, table_probe, &t, &r) != 0);
        o.max_pass = OAVIF_TQ_MAX_PASS + 1;
        CHECK(oavif_tq_find_target_quality(&o, table_probe, &t, &r) != 0);
    }
    // 3. interpolation on degenerate histories (equal scores, equal q, NaN-free extremes)
    for (int iter = 0; iter < 20000; ++iter) {
        oavif_tq_pass hist[OAVIF_TQ_MAX_PASS];
        const uint32_t n = rnd() % (OAVIF_TQ_MAX_PASS + 1);
        for (uint32_t i = 0; i < n; ++i) {
            hist[i].q = rnd() % 101;
            hist[i].score = (iter % 3 == 0) ? 50.0 : (iter % 3 == 1 ? std::floor(100.0 * frand()) : -100.0 + 200.0 * frand());
        }
        uint32_t lo = rnd() % 101, hi = rnd() % 101;
        if (lo > hi) std::swap(lo, hi);
        const uint32_t q = oavif_tq_interpolate_quantizer(lo, hi, n ? hist : nullptr, n, 100.0 * frand());
        CHECK(q >= lo && q <= hi);
    }
    for (int t = -50; t <= 250; ++t) CHECK(oavif_tq_predict_q_from_score((double)t) <= 100);
    // 4. the HIP-bound search entry point over the stand-in scorer, incl. a codec failure
    {
        const uint32_t w = 67, h = 41;
        std::vector<uint8_t> ref((size_t)w * h * 3);
        for (auto& b : ref) b = (uint8_t)(rnd() & 255);
        ssimu2_ctx ctx;
        Codec c = {ref.data(), ref.size(), 0};
        oavif_tq_options o;
        oavif_tq_default_options(&o);
        for (double tgt : {35.0, 60.0, 80.0, 95.0}) {
            o.score_tgt = tgt;
            oavif_tq_result r;
            size_t last = 0;
            const int rc = oavif_tq_search_hip(&o, &ctx, ref.data(), w, h, codec, &c, &r, &last);
            CHECK(rc == 0 || rc == 55);
            if (rc == 0) CHECK(last == 1000 + 10 * (size_t)r.buf_q && r.num_pass >= 1);
        }
        oavif_tq_result r;
        CHECK(oavif_tq_search_hip(&o, nullptr, ref.data(), w, h, codec, &c, &r, nullptr) != 0);
        CHECK(oavif_tq_search_hip(&o, &ctx, nullptr, w, h, codec, &c, &r, nullptr) != 0);
        CHECK(oavif_tq_search_hip(&o, &ctx, ref.data(), w, h, nullptr, &c, &r, nullptr) != 0);
    }
    // 5. pre-scaling loops on exact-size heap buffers (ASan catches a one-off)
    {
        for (size_t n : {(size_t)0, (size_t)1, (size_t)255, (size_t)65537}) {
            std::vector<uint8_t> s8(n), d8(n);
            std::vector<uint16_t> s16(n), d16(n);
            for (size_t i = 0; i < n; ++i) {
                s8[i] = (uint8_t)i;
                s16[i] = (uint16_t)(i * 257u);
            }
            oavif_prescale_8_to_10(s8.data(), n, d16.data());
            for (size_t i = 0; i < n; ++i) CHECK(d16[i] == (uint16_t)((s8[i] * 1023u + 127u) / 255u));
            oavif_prescale_16_to_10(s16.data(), n, d16.data());
            for (size_t i = 0; i < n; ++i) CHECK(d16[i] == (uint16_t)(s16[i] >> 6));
            oavif_prescale_16_to_8(s16.data(), n, d8.data());
            for (size_t i = 0; i < n; ++i) CHECK(d8[i] == (uint8_t)(s16[i] >> 8));
        }
    }
    std::printf("tq_sanitize ok: %ld searches\n", searches);
    return 0;
}
