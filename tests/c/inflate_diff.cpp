// Differential + robustness harness for oavif_amd/csrc/inflate_fast.h against zlib, CPU only (built by
// tests/test_sanitizers.py with -fsanitize=address,undefined).  Valid streams -- every zlib level and strategy over
// random, structured, degenerate and PNG-like data of many sizes -- are decoded through output strips of awkward
// sizes (the resumption path) and must reproduce zlib's bytes exactly.  Corrupted and truncated streams must end in
// an error, or in exactly the bytes zlib gives: never a crash, an out-of-bounds access or a hang.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <vector>

#include "../../oavif_amd/csrc/inflate_fast.h"

typedef std::vector<uint8_t> Bytes;
static uint32_t rs = 2463534242u;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 17; rs ^= rs << 5; return rs; }

// raw deflate of `src` with the given parameters
static Bytes deflate_raw(const Bytes& src, int level, int strategy, int mem) {
    z_stream z;
    memset(&z, 0, sizeof z);
    if (deflateInit2(&z, level, Z_DEFLATED, -15, mem, strategy) != Z_OK) abort();
    Bytes out(src.size() * 2 + 1024);  // deflateBound undershoots Z_FIXED on noise
    z.next_in = const_cast<uint8_t*>(src.data());
    z.avail_in = (uInt)src.size();
    z.next_out = out.data();
    z.avail_out = (uInt)out.size();
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) abort();
    out.resize(z.total_out);
    deflateEnd(&z);
    return out;
}

// zlib's verdict on a raw stream: 0 = ok (out filled), else error
static int zlib_inflate_raw(const Bytes& comp, size_t cap, Bytes& out) {
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, -15) != Z_OK) abort();
    out.assign(cap + 1, 0);
    z.next_in = const_cast<uint8_t*>(comp.data());
    z.avail_in = (uInt)comp.size();
    z.next_out = out.data();
    z.avail_out = (uInt)out.size();
    const int rc = inflate(&z, Z_FINISH);
    const size_t n = z.total_out;
    inflateEnd(&z);
    if (rc != Z_STREAM_END) return -1;
    out.resize(n);
    return 0;
}

// the fast decoder through a sliding strip buffer, as png_ingest.cpp drives it; cap = most bytes to accept
static int fast_inflate(const Bytes& comp, size_t strip, size_t cap, Bytes& out) {
    Bytes in(comp);
    in.insert(in.end(), finf::kPad, 0);
    static finf::Stream s;
    s.init(in.data(), comp.size());
    Bytes buf(finf::kWindow + strip + finf::kOutMargin);
    size_t have = 0;  // valid bytes at the start of buf (history)
    out.clear();
    for (int guard = 0; guard < 1 << 24; ++guard) {
        uint8_t* o = buf.data() + have;
        const finf::Result r = finf::run(s, buf.data(), o, buf.data() + buf.size());
        const size_t got = (size_t)(o - (buf.data() + have));
        out.insert(out.end(), buf.data() + have, o);
        if (out.size() > cap) return -2;  // more than the caller would take
        have += got;
        if (r == finf::kError) return -1;
        if (r == finf::kDone) return s.overrun() ? -1 : 0;
        if (got == 0 && have < buf.size() - finf::kOutMargin) return -3;  // no progress although there is room
        if (have > finf::kWindow) {  // slide: keep the last kWindow bytes
            memmove(buf.data(), buf.data() + have - finf::kWindow, finf::kWindow);
            have = finf::kWindow;
        }
    }
    return -4;
}

static Bytes make_data(int kind, size_t n) {
    Bytes d(n);
    switch (kind) {
        case 0: for (auto& v : d) v = (uint8_t)rnd(); break;                                      // noise
        case 1: for (size_t i = 0; i < n; ++i) d[i] = (uint8_t)(i * 7 + (i >> 8)); break;          // ramps
        case 2: memset(d.data(), 0, n); break;                                                     // zeros (distance 1 runs)
        case 3: for (size_t i = 0; i < n; ++i) d[i] = (uint8_t)((rnd() & 7) == 0 ? rnd() : d[i >= 3 ? i - 3 : 0]); break;  // period 3
        case 4: for (size_t i = 0; i < n; ++i) d[i] = (uint8_t)("the quick brown fox "[i % 20] + ((rnd() & 63) == 0)); break;
        case 5: for (size_t i = 0; i < n; ++i) d[i] = (uint8_t)((rnd() % 5 == 0 ? rnd() : 0) & 15); break;       // few symbols
        default: for (size_t i = 0; i < n; ++i) d[i] = (uint8_t)(d[i >= 4 ? i - 4 : 0] + (int)(rnd() % 5) - 2); break;  // PNG-like residuals
    }
    return d;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 60;
    static const size_t sizes[] = {0, 1, 2, 7, 257, 258, 259, 4096, 32767, 32768, 32769, 65535, 65536, 70000, 300000};
    static const size_t strips[] = {300, 1000, 4093, 65536, 262144};
    long valid = 0, corrupt = 0, both_ok = 0;
    for (int r = 0; r < rounds; ++r) {
        const size_t n = r < 15 ? sizes[r] : (size_t)(rnd() % 200000);
        const Bytes src = make_data(r % 7, n);
        static const int levels[] = {0, 1, 3, 6, 9};
        static const int strategies[] = {Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED};
        const Bytes comp = deflate_raw(src, levels[(r / 7) % 5], strategies[r % 5], 1 + (int)(rnd() % 9));
        for (size_t strip : strips) {
            Bytes out;
            const int rc = fast_inflate(comp, strip, src.size(), out);
            if (rc != 0 || out != src) {
                fprintf(stderr, "MISMATCH valid stream: round %d size %zu strip %zu rc %d out %zu\n", r, n, strip, rc, out.size());
                return 1;
            }
            ++valid;
        }
        // corruptions: flips, truncations, garbage tails
        for (int c = 0; c < 40 && !comp.empty(); ++c) {
            Bytes bad(comp);
            const int what = (int)(rnd() % 4);
            if (what == 0) bad[rnd() % bad.size()] ^= (uint8_t)(1u << (rnd() % 8));
            else if (what == 1) bad.resize(rnd() % bad.size());
            else if (what == 2) for (int k = 0; k < 4; ++k) bad[rnd() % bad.size()] = (uint8_t)rnd();
            else bad.insert(bad.begin() + (long)(rnd() % bad.size()), (uint8_t)rnd());
            Bytes zout, fout;
            const size_t cap = src.size() + 70000;
            const int zrc = zlib_inflate_raw(bad, cap, zout);
            const int frc = fast_inflate(bad, strips[c % 5], cap, fout);
            ++corrupt;
            if (zrc == 0 && frc == 0) {
                ++both_ok;
                if (zout != fout) {
                    fprintf(stderr, "MISMATCH both decoders accept a corrupted stream but disagree: round %d case %d\n", r, c);
                    return 1;
                }
            } else if (zrc == 0 && frc != 0 && frc != -2) {
                // zlib accepts what this decoder refuses: only legitimate for streams zlib is lenient about
                // (none known); report it
                fprintf(stderr, "NOTE zlib accepts, fast refuses (rc %d): round %d case %d\n", frc, r, c);
                return 1;
            } else if (zrc != 0 && frc == 0) {
                // the other direction: a stream zlib -- and so libspng, the reference's PNG decoder --
                // reports as a decode failure must not turn into pixels here (e.g. incomplete Huffman codes)
                fprintf(stderr, "NOTE fast accepts, zlib refuses: round %d case %d\n", r, c);
                return 1;
            }
        }
    }
    // a stored block whose header is cut by the end of input.  LEN = 0xFFFF reads the zero padding
    // as NLEN = 0x0000 and passes the complement test; the decoder must report the truncation by position instead of
    // copying 65535 bytes from behind its input.  Every prefix of a level-0 stream's last block header, the 4-byte
    // stream of the finding itself, and a hand-made incomplete literal/length code.
    {
        long cut = 0;
        Bytes src65535 = make_data(0, 65535);
        const Bytes whole = deflate_raw(src65535, 0, Z_DEFAULT_STRATEGY, 8);  // 01 FF FF 00 00 <65535 bytes>
        for (size_t keep = 0; keep < 12 && keep < whole.size(); ++keep) {
            Bytes bad(whole.begin(), whole.begin() + (long)keep), zout, fout;
            const int zrc = zlib_inflate_raw(bad, 70000, zout);
            for (size_t strip : strips) {
                const int frc = fast_inflate(bad, strip, 70000, fout);
                if (zrc == 0 || frc == 0) {
                    fprintf(stderr, "MISMATCH stored stream cut after %zu bytes: zlib %d fast %d\n", keep, zrc, frc);
                    return 1;
                }
                ++cut;
            }
        }
        Bytes two = make_data(0, 65535 + 300);  // two stored blocks: cut inside the SECOND header as well
        const Bytes whole2 = deflate_raw(two, 0, Z_DEFAULT_STRATEGY, 8);
        for (size_t keep = 65535 + 5; keep <= 65535 + 5 + 6 && keep < whole2.size(); ++keep) {
            Bytes bad(whole2.begin(), whole2.begin() + (long)keep), zout, fout;
            const int zrc = zlib_inflate_raw(bad, 140000, zout);
            const int frc = fast_inflate(bad, 4093, 140000, fout);
            if (zrc == 0 || frc == 0) {
                fprintf(stderr, "MISMATCH second stored header cut at %zu: zlib %d fast %d\n", keep, zrc, frc);
                return 1;
            }
            ++cut;
        }
        static const uint8_t finding[4] = {0x01, 0xFF, 0xFF, 0x00};
        for (size_t keep = 1; keep <= 4; ++keep) {
            Bytes bad(finding, finding + keep), zout, fout;
            if (zlib_inflate_raw(bad, 70000, zout) == 0 || fast_inflate(bad, 65536, 70000, fout) == 0) {
                fprintf(stderr, "MISMATCH the 4-byte stored header of ADVICE r04 (first %zu bytes) is accepted\n", keep);
                return 1;
            }
            ++cut;
        }
        printf("inflate_diff: %ld truncated stored headers refused by both decoders\n", cut);
    }
    printf("inflate_diff ok: %ld valid decodes identical to the source, %ld corrupted streams (%ld accepted by both, same bytes)\n",
           valid, corrupt, both_ok);
    return 0;
}
