// A resumable DEFLATE (RFC 1951) decoder for the PNG ingest (png_ingest.cpp), written for speed on a 64-bit host.
//
// Why it exists: after the SSE2 row filters, loading a PNG is zlib's inflate and little else (a 1920x1080 RGB8
// file: 39 of 47 ms), and the batch driver of BASELINE configs[3] loads one PNG per image.  This decoder keeps the
// properties the loader needs from zlib -- bounded memory whatever the header claims (it writes into the caller's
// strip buffer and stops when that is full, resuming where it stopped), an error instead of a crash on any
// malformed stream -- and drops what it does not need (no dictionary, no gzip, no flush modes), which lets the hot
// loop be what fast inflaters are made of: a 64-bit bit buffer refilled eight bytes at a time, one table look-up
// per literal / length / distance with the extra-bit counts and bases packed in the entry, and eight-byte match
// copies.  Input is ONE contiguous buffer followed by at least kPad zero bytes (the caller concatenates the IDAT
// chunks), so a refill never reads outside the buffer: it stops once `in` has passed the real end (at most 7 bytes into
// the padding), after which the bit count runs negative and the block ends in an error; a stream that ends exactly
// inside the padding is detected afterwards (`overrun()`), as a truncated stream.
//
// Checked against zlib itself (tests/c/inflate_diff.cpp, run by tests/test_sanitizers.py under ASan + UBSan):
// every level and strategy of deflate over random, structured and degenerate data, decoded through strips of
// every awkward size, must give zlib's bytes; corrupted and truncated streams must give an error or zlib's bytes.
// Host code; no HIP.
#pragma once

#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace finf {

constexpr size_t kPad = 16;         // zero bytes the caller keeps behind the input
constexpr size_t kWindow = 32768;   // history a match may reach back into: the caller keeps that much before `out`
constexpr size_t kOutMargin = 258 + 16;  // a match (258) + its 7 bytes of copy overshoot, or three literals  // run() returns kNeedOutput when fewer bytes than this are free

enum Result { kDone = 0, kNeedOutput = 1, kError = -1 };

// table entry: bits 0-7 = bits to consume for the code itself (the subtable entry holds the whole code's length),
// bits 8-11 = extra bits, bits 12-15 = kind, bits 16-31 = literal / base value / subtable offset
constexpr uint32_t kLit = 1u << 12, kLen = 2u << 12, kEob = 3u << 12, kSub = 4u << 12, kBad = 0;  // kind 0 = unused code
constexpr int kLitBits = 11, kDistBits = 8, kPreBits = 7;
constexpr int kLitSize = (1 << kLitBits) + 600, kDistSize = (1 << kDistBits) + 420;   // primary + room for subtables

struct Stream {
    const uint8_t* in_begin = nullptr;
    const uint8_t* in = nullptr;
    const uint8_t* in_end = nullptr;   // real end of input; kPad zero bytes follow
    uint64_t bitbuf = 0;
    int bitcnt = 0;
    int state = 0;                     // 0 block header, 1 stored, 2 huffman, 3 done
    bool final_block = false;
    uint32_t stored_left = 0;
    uint64_t produced = 0;             // bytes written so far over all calls (bounds match distances)
    uint32_t lit[kLitSize];
    uint32_t dist[kDistSize];

    void init(const uint8_t* data, size_t n) {
        in_begin = in = data;
        in_end = data + n;
        bitbuf = 0;
        bitcnt = 0;
        state = 0;
        final_block = false;
        stored_left = 0;
        produced = 0;
    }
    // bytes of input really consumed (whole bytes still in the bit buffer are not)
    size_t consumed() const { return (size_t)(in - in_begin) - (size_t)(bitcnt >> 3); }
    bool overrun() const { return consumed() > (size_t)(in_end - in_begin); }
};

inline uint64_t load64(const uint8_t* p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;  // little-endian host (x86-64)
}

// Canonical Huffman decode table: primary table of `tbits` bits, subtables for longer codes.  `payload(sym)`
// supplies the payload of a symbol.  Returns false for an over-subscribed code, and for an incomplete one under
// zlib's rule (inftrees.c): the code-length alphabet (`strict`) must be complete; a literal/length or distance code
// may be incomplete only when it has no codes longer than one bit (the single-code distance tree deflate writes for
// a block with one distance, or no distance codes at all) -- its unused slots stay kBad, an error only if the
// stream reaches one.  A corrupted IDAT that zlib / libspng report as a decode failure fails here too.
template <typename Payload>
inline bool build_table(uint32_t* table, int table_cap, int tbits, const uint8_t* lens, int nsyms, bool strict, Payload payload) {
    int count[16] = {0};
    for (int i = 0; i < nsyms; ++i) ++count[lens[i]];
    count[0] = 0;
    int left = 1, maxlen = 0;
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - count[l];
        if (left < 0) return false;  // over-subscribed
        if (count[l]) maxlen = l;
    }
    if (left > 0 && maxlen > 0 && (strict || maxlen != 1)) return false;  // incomplete set
    int offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + count[l];
    uint16_t sorted[320];
    for (int i = 0; i < nsyms; ++i)
        if (lens[i]) sorted[offs[lens[i]]++] = (uint16_t)i;
    const int primary = 1 << tbits;
    for (int i = 0; i < primary; ++i) table[i] = kBad;
    int next_sub = primary;
    // codes in canonical order; `code` is kept bit-reversed implicitly by reversing on insert
    uint32_t code = 0;
    int idx = 0;
    // first pass: how many bits each subtable needs = the longest code sharing its primary prefix
    int sub_bits_of[1 << kLitBits];  // tbits <= kLitBits
    for (int i = 0; i < primary; ++i) sub_bits_of[i] = 0;
    {
        uint32_t c = 0;
        for (int l = 1; l <= 15; ++l) {
            for (int k = 0; k < count[l]; ++k, ++c) {
                if (l > tbits) {
                    uint32_t rev = 0;
                    for (int b = 0; b < l; ++b) rev |= ((c >> (l - 1 - b)) & 1u) << b;
                    const int p = (int)(rev & (uint32_t)(primary - 1));
                    if (l - tbits > sub_bits_of[p]) sub_bits_of[p] = l - tbits;
                }
            }
            c <<= 1;
        }
    }
    for (int l = 1; l <= 15; ++l) {
        for (int k = 0; k < count[l]; ++k, ++code, ++idx) {
            const int sym = sorted[idx];
            uint32_t rev = 0;
            for (int b = 0; b < l; ++b) rev |= ((code >> (l - 1 - b)) & 1u) << b;
            const uint32_t e = payload(sym) | (uint32_t)l;
            if (l <= tbits) {
                for (uint32_t i = rev; i < (uint32_t)primary; i += 1u << l) table[i] = e;
            } else {
                const int p = (int)(rev & (uint32_t)(primary - 1));
                const int sb = sub_bits_of[p];
                if ((table[p] >> 12 & 0xf) != 4) {  // open the subtable of this prefix
                    if (next_sub + (1 << sb) > table_cap) return false;
                    table[p] = kSub | ((uint32_t)next_sub << 16) | (uint32_t)sb << 8 | (uint32_t)tbits;
                    for (int i = 0; i < (1 << sb); ++i) table[next_sub + i] = kBad;
                    next_sub += 1 << sb;
                }
                const int base = (int)(table[p] >> 16);
                for (uint32_t i = rev >> tbits; i < (1u << sb); i += 1u << (l - tbits)) table[base + i] = e;
            }
        }
        code <<= 1;
    }
    return true;
}

constexpr uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
constexpr uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
constexpr uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
constexpr uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t litlen_payload(int sym) {
    if (sym < 256) return kLit | ((uint32_t)sym << 16);
    if (sym == 256) return kEob;
    if (sym > 285) return kBad | 0xf00u;  // 286, 287: never valid; kind 0 with a non-zero length is still "bad"
    return kLen | ((uint32_t)kLenBase[sym - 257] << 16) | ((uint32_t)kLenExtra[sym - 257] << 8);
}
inline uint32_t dist_payload(int sym) {
    if (sym > 29) return kBad | 0xf00u;
    return kLen | ((uint32_t)kDistBase[sym] << 16) | ((uint32_t)kDistExtra[sym] << 8);
}

#define FINF_REFILL(s)                                           \
    do {                                                         \
        if ((s).in <= (s).in_end) {                              \
            (s).bitbuf |= load64((s).in) << (s).bitcnt;          \
            (s).in += (63 - (s).bitcnt) >> 3;                    \
            (s).bitcnt |= 56;                                    \
        }                                                        \
    } while (0)

inline bool read_block_header(Stream& s) {
    FINF_REFILL(s);
    s.final_block = s.bitbuf & 1;
    const int type = (int)((s.bitbuf >> 1) & 3);
    s.bitbuf >>= 3;
    s.bitcnt -= 3;
    if (type == 0) {  // stored: skip to a byte boundary, LEN, NLEN
        const int skip = s.bitcnt & 7;
        s.bitbuf >>= skip;
        s.bitcnt -= skip;
        FINF_REFILL(s);
        const uint32_t len = (uint32_t)(s.bitbuf & 0xffff), nlen = (uint32_t)((s.bitbuf >> 16) & 0xffff);
        if ((len ^ 0xffff) != nlen) return false;
        s.bitbuf >>= 32;
        s.bitcnt -= 32;
        if (s.bitcnt < 0) return false;  // the header itself ran past the end of input
        // hand the whole bytes still in the bit buffer back to the byte stream
        s.in -= s.bitcnt >> 3;
        s.bitbuf = 0;
        s.bitcnt = 0;
        // LEN / NLEN were (partly) the zero padding: 0xFFFF / 0x0000 passes the complement test, so a stream cut
        // inside a stored header must be caught by position (the refill above is skipped once `in` is past the end)
        if (s.in > s.in_end) return false;
        s.stored_left = len;
        s.state = 1;
        return true;
    }
    uint8_t lens[320];
    int nlit, ndist;
    if (type == 1) {
        nlit = 288;
        ndist = 32;  // the fixed distance code is the complete 5-bit one; its symbols 30 and 31 are invalid (dist_payload)
        for (int i = 0; i < 144; ++i) lens[i] = 8;
        for (int i = 144; i < 256; ++i) lens[i] = 9;
        for (int i = 256; i < 280; ++i) lens[i] = 7;
        for (int i = 280; i < 288; ++i) lens[i] = 8;
        for (int i = 0; i < 32; ++i) lens[288 + i] = 5;
    } else if (type == 2) {
        nlit = (int)(s.bitbuf & 31) + 257;
        ndist = (int)((s.bitbuf >> 5) & 31) + 1;
        const int npre = (int)((s.bitbuf >> 10) & 15) + 4;
        s.bitbuf >>= 14;
        s.bitcnt -= 14;
        if (nlit > 286 || ndist > 30) return false;
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t pre[19] = {0};
        FINF_REFILL(s);
        for (int i = 0; i < npre; ++i) {
            if (i == 14) FINF_REFILL(s);  // 19 x 3 = 57 bits: one refill does not cover them all
            pre[order[i]] = (uint8_t)(s.bitbuf & 7);
            s.bitbuf >>= 3;
            s.bitcnt -= 3;
        }
        uint32_t ptab[(1 << kPreBits) + 8];
        if (!build_table(ptab, (1 << kPreBits) + 8, kPreBits, pre, 19, true, [](int sym) { return kLit | ((uint32_t)sym << 16); }))
            return false;
        int i = 0;
        while (i < nlit + ndist) {
            FINF_REFILL(s);
            const uint32_t e = ptab[s.bitbuf & ((1u << kPreBits) - 1)];
            if ((e >> 12 & 0xf) != 1) return false;
            s.bitbuf >>= e & 0xff;
            s.bitcnt -= (int)(e & 0xff);
            const int sym = (int)(e >> 16);
            if (sym < 16) {
                lens[i++] = (uint8_t)sym;
                continue;
            }
            int rep, val = 0;
            if (sym == 16) {
                if (i == 0) return false;
                val = lens[i - 1];
                rep = 3 + (int)(s.bitbuf & 3);
                s.bitbuf >>= 2;
                s.bitcnt -= 2;
            } else if (sym == 17) {
                rep = 3 + (int)(s.bitbuf & 7);
                s.bitbuf >>= 3;
                s.bitcnt -= 3;
            } else {
                rep = 11 + (int)(s.bitbuf & 127);
                s.bitbuf >>= 7;
                s.bitcnt -= 7;
            }
            if (i + rep > nlit + ndist || s.bitcnt < 0) return false;
            while (rep--) lens[i++] = (uint8_t)val;
        }
        if (s.bitcnt < 0) return false;
        if (lens[256] == 0) return false;  // no end-of-block code
        memmove(lens + 288, lens + nlit, (size_t)ndist);  // distances behind a fixed offset
        for (int k = nlit; k < 288; ++k) lens[k] = 0;
    } else {
        return false;
    }
    if (s.bitcnt < 0) return false;
    if (!build_table(s.lit, kLitSize, kLitBits, lens, type == 1 ? 288 : nlit, false, litlen_payload)) return false;
    if (!build_table(s.dist, kDistSize, kDistBits, lens + 288, ndist, false, dist_payload)) return false;
    s.state = 2;
    return true;
}

// Decode into [out, out_end); `out_begin` is the oldest byte a match may copy from that is still in the caller's
// buffer (the caller keeps at least min(kWindow, produced) bytes before `out`).  Returns kNeedOutput with fewer
// than kOutMargin bytes free -- the caller takes what is there, slides its window and calls again --, kDone at the
// end of the final block, kError on a malformed stream.
inline Result run(Stream& s, const uint8_t* out_begin, uint8_t*& out, uint8_t* out_end) {
    for (;;) {
        if (s.state == 3) return kDone;
        if (s.state == 0) {
            if (!read_block_header(s)) return kError;
            continue;
        }
        if (s.state == 1) {  // stored
            while (s.stored_left) {
                size_t n = s.stored_left;
                if ((size_t)(out_end - out) < n) n = (size_t)(out_end - out);
                if (n == 0) return kNeedOutput;
                const size_t avail = s.in < s.in_end ? (size_t)(s.in_end - s.in) : 0;
                if (avail < n) return kError;  // truncated
                memcpy(out, s.in, n);
                out += n;
                s.in += n;
                s.produced += n;
                s.stored_left -= (uint32_t)n;
            }
            s.state = s.final_block ? 3 : 0;
            continue;
        }
        // Huffman block
        uint64_t bitbuf = s.bitbuf;
        int bitcnt = s.bitcnt;
        const uint8_t* in = s.in;
        const uint8_t* const in_end = s.in_end;
        uint8_t* o = out;
        const uint32_t* const lit = s.lit;
        const uint32_t* const dist = s.dist;
        Result res = kError;
#define REFILL()                                  \
    do {                                          \
        if (in <= in_end) {                       \
            bitbuf |= load64(in) << bitcnt;       \
            in += (63 - bitcnt) >> 3;             \
            bitcnt |= 56;                         \
        }                                         \
    } while (0)
        for (;;) {
            if ((size_t)(out_end - o) < kOutMargin) {
                res = kNeedOutput;
                break;
            }
            REFILL();  // >= 56 bits: a litlen code (15) + extra (5) + a distance code (15) + extra (13) = 48
            uint32_t e = lit[bitbuf & ((1u << kLitBits) - 1)];
            if ((e >> 12 & 0xf) == 4) e = lit[(e >> 16) + ((bitbuf >> kLitBits) & ((1u << (e >> 8 & 0xf)) - 1))];
            const uint32_t kind = e >> 12 & 0xf;
            bitbuf >>= e & 0xff;
            bitcnt -= (int)(e & 0xff);
            if (kind == 1) {
                *o++ = (uint8_t)(e >> 16);
                // a second literal from the bits already there (no refill needed: 15 more bits at most)
                uint32_t e2 = lit[bitbuf & ((1u << kLitBits) - 1)];
                if ((e2 >> 12 & 0xf) == 1) {
                    bitbuf >>= e2 & 0xff;
                    bitcnt -= (int)(e2 & 0xff);
                    *o++ = (uint8_t)(e2 >> 16);
                    e2 = lit[bitbuf & ((1u << kLitBits) - 1)];  // and a third: 3 x 15 = 45 <= 56 bits
                    if ((e2 >> 12 & 0xf) == 1) {
                        bitbuf >>= e2 & 0xff;
                        bitcnt -= (int)(e2 & 0xff);
                        *o++ = (uint8_t)(e2 >> 16);
                    }
                }
                continue;
            }
            if (kind == 2) {
                const uint32_t lx = e >> 8 & 0xf;
                const uint32_t length = (e >> 16) + (uint32_t)(bitbuf & ((1u << lx) - 1));
                bitbuf >>= lx;
                bitcnt -= (int)lx;
                uint32_t d = dist[bitbuf & ((1u << kDistBits) - 1)];
                if ((d >> 12 & 0xf) == 4) d = dist[(d >> 16) + ((bitbuf >> kDistBits) & ((1u << (d >> 8 & 0xf)) - 1))];
                if ((d >> 12 & 0xf) != 2) break;  // unused distance code
                bitbuf >>= d & 0xff;
                bitcnt -= (int)(d & 0xff);
                const uint32_t dx = d >> 8 & 0xf;
                const uint32_t distance = (d >> 16) + (uint32_t)(bitbuf & ((1u << dx) - 1));
                bitbuf >>= dx;
                bitcnt -= (int)dx;
                if (bitcnt < 0) break;  // consumed bits that were never loaded (past the end of input)
                if (distance > (uint64_t)(o - out) + s.produced || distance > (size_t)(o - out_begin)) break;  // before the start
                const uint8_t* src = o - distance;
                uint8_t* const end = o + length;
                if (distance >= 8) {
                    do {  // may write up to 7 bytes past `end`: inside kOutMargin
                        memcpy(o, src, 8);
                        o += 8;
                        src += 8;
                    } while (o < end);
                } else if (distance == 1) {
                    memset(o, *src, length);
                } else {
                    do *o++ = *src++;
                    while (o < end);
                }
                o = end;
                continue;
            }
            if (kind == 3) {
                if (bitcnt < 0) break;
                res = kDone;  // end of block
                break;
            }
            break;  // unused litlen code
        }
#undef REFILL
        if (bitcnt < 0) res = kError;
        s.produced += (uint64_t)(o - out);
        out = o;
        s.bitbuf = bitbuf;
        s.bitcnt = bitcnt < 0 ? 0 : bitcnt;
        s.in = in;
        if (res == kError) return kError;
        if (res == kNeedOutput) return kNeedOutput;
        s.state = s.final_block ? 3 : 0;  // end of block
    }
}

#undef FINF_REFILL

}  // namespace finf
