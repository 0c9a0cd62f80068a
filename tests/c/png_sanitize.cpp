// Sanitizer + mutation harness for the native PNG ingest (oavif_amd/csrc/png_ingest.cpp), CPU only.
// Built by tests/test_sanitizers.py with g++ -fsanitize=address,undefined: png_ingest.cpp + this
// file + zlib.  It writes a few valid PNGs of every colour type / depth / interlace (stored zlib
// blocks, filter 0..4 cycling), checks that they decode, then feeds the decoder thousands of
// corruptions of them -- byte flips with and without the chunk CRC repaired, truncations, length
// fields blown up -- and requires an error code or a clean decode every time: no crash, no
// out-of-bounds access, no sanitizer report (-fno-sanitize-recover).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <vector>

#include "oavif_tq.h"

typedef std::vector<uint8_t> Bytes;

static void put32(Bytes& b, uint32_t v) {
    for (int s = 24; s >= 0; s -= 8) b.push_back((uint8_t)(v >> s));
}
static void chunk(Bytes& out, const char* type, const Bytes& data) {
    put32(out, (uint32_t)data.size());
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put32(out, (uint32_t)crc32(0L, out.data() + at, (uInt)(4 + data.size())));
}
static uint32_t rnd_state = 12345u;
static uint32_t rnd() { return rnd_state = rnd_state * 1664525u + 1013904223u; }

static Bytes make_png(uint32_t w, uint32_t h, int ctype, int depth, int interlace) {
    static int made = 0;
    const bool low_entropy = (made++ & 1) != 0;
    const int samples = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4;
    Bytes png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A}, d;
    put32(d, w);
    put32(d, h);
    d.push_back((uint8_t)depth);
    d.push_back((uint8_t)ctype);
    d.push_back(0);
    d.push_back(0);
    d.push_back((uint8_t)interlace);
    chunk(png, "IHDR", d);
    if (ctype == 3) {
        Bytes pl(3 * (1u << depth));
        for (auto& v : pl) v = (uint8_t)rnd();
        chunk(png, "PLTE", pl);
        chunk(png, "tRNS", Bytes(2, 128));
    }
    static const int ax[7] = {0, 4, 0, 2, 0, 1, 0}, ay[7] = {0, 0, 4, 0, 2, 0, 1}, adx[7] = {8, 8, 4, 4, 2, 2, 1},
                     ady[7] = {8, 8, 8, 4, 4, 2, 2};
    Bytes raw;
    int f = 0;
    for (int k = 0; k < (interlace ? 7 : 1); ++k) {
        const uint32_t pw = interlace ? (w > (uint32_t)ax[k] ? (w - ax[k] + adx[k] - 1) / adx[k] : 0) : w;
        const uint32_t ph = interlace ? (h > (uint32_t)ay[k] ? (h - ay[k] + ady[k] - 1) / ady[k] : 0) : h;
        if (!pw || !ph) continue;
        const size_t rb = ((size_t)pw * samples * depth + 7) / 8;
        for (uint32_t y = 0; y < ph; ++y) {
            raw.push_back((uint8_t)(f++ % 5));  // any filter type: the bytes are random anyway
            // every other file: few distinct bytes in runs, so that the deflate stream is made of matches (the
            // match / window paths of the inflater), not of literals alone
            for (size_t i = 0; i < rb; ++i) raw.push_back((uint8_t)(low_entropy ? ((rnd() >> 13) % 7 == 0 ? (rnd() >> 11) & 3 : (raw.empty() ? 0 : raw.back())) : rnd() >> 11));
        }
    }
    uLongf cap = compressBound((uLong)raw.size());
    Bytes comp(cap);
    compress2(comp.data(), &cap, raw.data(), (uLong)raw.size(), low_entropy ? 9 : 1);
    comp.resize(cap);
    chunk(png, "IDAT", comp);
    chunk(png, "IEND", Bytes());
    return png;
}

static int decode(const Bytes& png) {
    oavif_png_info info;
    int rc = oavif_png_info_from_memory(png.data(), png.size(), &info);
    if (rc) return rc;
    if (info.data_bytes > (64u << 20)) return OAVIF_PNG_ERR_SIZE;  // a corrupted IHDR may ask for gigabytes
    std::vector<uint16_t> out((info.data_bytes + 1) / 2);
    Bytes icc(info.icc_bytes);
    return oavif_png_decode(png.data(), png.size(), (uint8_t*)out.data(), info.data_bytes, icc.data(), icc.size());
}

// repair the CRC of the chunk that contains byte `pos`, so the corruption reaches the decoder proper
static void fix_crc(Bytes& png, size_t pos) {
    size_t at = 8;
    while (at + 12 <= png.size()) {
        const uint32_t len = (uint32_t)png[at] << 24 | png[at + 1] << 16 | png[at + 2] << 8 | png[at + 3];
        if ((size_t)len > png.size() - at - 12) return;
        if (pos >= at + 4 && pos < at + 8 + len) {
            const uint32_t c = (uint32_t)crc32(0L, png.data() + at + 4, 4 + len);
            for (int k = 0; k < 4; ++k) png[at + 8 + len + k] = (uint8_t)(c >> (24 - 8 * k));
            return;
        }
        at += 12 + len;
    }
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 300;
    static const int kinds[][2] = {{0, 1}, {0, 4}, {0, 8}, {0, 16}, {2, 8}, {2, 16}, {3, 2}, {3, 8}, {4, 8}, {4, 16}, {6, 8}, {6, 16}};
    std::vector<Bytes> seeds;
    for (auto& k : kinds)
        for (int il = 0; il < 2; ++il) seeds.push_back(make_png(1 + rnd() % 40, 1 + rnd() % 40, k[0], k[1], il));
    for (const Bytes& s : seeds)
        if (decode(s) != OAVIF_PNG_OK) {
            printf("a valid seed failed to decode\n");
            return 1;
        }
    // lying headers: the IHDR of a small valid file rewritten to claim a huge image, CRC
    // repaired.  The info call itself must refuse what the IDAT bytes cannot inflate to (1032 : 1), so no
    // caller sizes gigabytes from it; the decode is called directly with a small buffer as well.
    {
        static const uint32_t dims[][2] = {{60000, 60000}, {0x7FFFFFFF, 2}, {2, 0x7FFFFFFF}, {1u << 20, 1u << 20}};
        for (const Bytes& s : seeds)
            for (auto& d : dims) {
                Bytes m = s;
                for (int k = 0; k < 4; ++k) m[16 + k] = (uint8_t)(d[0] >> (24 - 8 * k)), m[20 + k] = (uint8_t)(d[1] >> (24 - 8 * k));
                fix_crc(m, 16);
                oavif_png_info info;
                if (oavif_png_info_from_memory(m.data(), m.size(), &info) == OAVIF_PNG_OK) {
                    printf("a header claiming %u x %u over %zu bytes of file was accepted\n", d[0], d[1], m.size());
                    return 1;
                }
                uint16_t small[64];
                if (oavif_png_decode(m.data(), m.size(), (uint8_t*)small, sizeof small, nullptr, 0) == OAVIF_PNG_OK) {
                    printf("a lying header decoded\n");
                    return 1;
                }
            }
    }
    // a 40 x 40 RGB8 file whose IDAT is a zlib header + a stored block header cut after LEN and
    // the first byte of NLEN (78 01 | 01 FF FF 00): the zero padding behind the input completed NLEN, and the inflater
    // copied 65535 bytes from behind the IDAT bytes into the caller's pixels.  Every prefix of such an IDAT must be
    // refused, with nothing read outside the file (ASan).
    {
        static const uint8_t idat[] = {0x78, 0x01, 0x01, 0xFF, 0xFF, 0x00, 0x00};
        for (size_t keep = 2; keep <= sizeof idat; ++keep) {
            Bytes png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A}, d;
            put32(d, 40);
            put32(d, 40);
            d.push_back(8);
            d.push_back(2);
            d.push_back(0);
            d.push_back(0);
            d.push_back(0);
            chunk(png, "IHDR", d);
            chunk(png, "IDAT", Bytes(idat, idat + keep));
            chunk(png, "IEND", Bytes());
            std::vector<uint8_t> px(40 * 40 * 3);
            if (oavif_png_decode(png.data(), png.size(), px.data(), px.size(), nullptr, 0) == OAVIF_PNG_OK) {
                printf("a stored block cut inside its header (IDAT of %zu bytes) decoded\n", keep);
                return 1;
            }
        }
    }
    long ok = 0, rejected = 0;
    for (int r = 0; r < rounds; ++r)
        for (const Bytes& s : seeds) {
            Bytes m = s;
            const int kind = rnd() % 4;
            if (kind == 0) {
                m.resize(rnd() % (m.size() + 1));  // truncation
            } else {
                const int flips = 1 + rnd() % 3;
                for (int f = 0; f < flips; ++f) {
                    const size_t pos = 8 + rnd() % (m.size() - 8);
                    m[pos] ^= (uint8_t)(1u << (rnd() % 8));
                    if (kind >= 2) fix_crc(m, pos);  // let it through the CRC check
                }
            }
            (decode(m) == OAVIF_PNG_OK ? ok : rejected)++;
        }
    printf("png_sanitize ok: %zu seeds, %ld mutants decoded, %ld rejected\n", seeds.size(), ok, rejected);
    return 0;
}
