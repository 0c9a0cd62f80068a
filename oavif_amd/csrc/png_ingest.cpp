// PNG ingest with the output rules of oavif's loader.
//
//   /root/reference/src/io.zig:242-307  loadPNG: libspng, decode flags 0, output format chosen as
//       bit depth 16               -> SPNG_FMT_RGBA16  (host-endian u16, 4 channels, hbd = true)
//       8-bit truecolour           -> SPNG_FMT_RGB8    (3 channels)
//       everything else            -> SPNG_FMT_RGBA8   (4 channels: gray / gray+alpha / palette /
//                                     RGBA; sub-byte gray scaled to 8 bits)
//   and the iCCP profile, decompressed, handed on unchanged (io.zig:261-268).
//   Decode flags are 0 (io.zig:285), i.e. no SPNG_DECODE_TRNS: a tRNS chunk is NOT applied, files
//   without an alpha channel come out opaque (alpha 255 / 65535).  libspng is not importable here and
//   the reference holds no PNG fixture, so this reading of its flags is unpinned.
//
// libspng is not in this image and is third-party anyway; this is the PNG specification
// (signature, chunk CRCs, IHDR / PLTE / tRNS / iCCP / IDAT, zlib inflate, the five row filters,
// Adam7) written against zlib alone.  Critical chunks with a bad CRC fail the decode, ancillary
// ones are skipped (libspng's defaults).  No gamma handling (flags 0).  Host-side code: no HIP.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <new>
#include <vector>

#include "../../include/oavif_tq.h"
#include "inflate_fast.h"

namespace {

inline uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
inline uint16_t be16(const uint8_t* p) { return (uint16_t)(p[0] << 8 | p[1]); }

struct Span {
    const uint8_t* p;
    size_t n;
};

struct Png {
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    Span plte{nullptr, 0}, iccp{nullptr, 0};
    std::vector<Span> idat;
    int samples() const { return ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4; }
    int bits_per_pixel() const { return samples() * depth; }
    bool hbd() const { return depth == 16; }
    uint32_t out_channels() const { return hbd() ? 4u : (ctype == 2 ? 3u : 4u); }
};

// chunk walk; returns an oavif_png error code
int parse(const uint8_t* buf, size_t len, Png& png) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (len < 8 + 25 || memcmp(buf, sig, 8) != 0) return OAVIF_PNG_ERR_HEADER;
    size_t pos = 8;
    bool first = true, seen_idat = false, idat_closed = false, end = false;
    while (!end) {
        if (len - pos < 12) return first ? OAVIF_PNG_ERR_HEADER : OAVIF_PNG_ERR_DECODE;
        const uint32_t clen = be32(buf + pos);
        const uint8_t* type = buf + pos + 4;
        if (clen > 0x7FFFFFFFu || (size_t)clen > len - pos - 12) return first ? OAVIF_PNG_ERR_HEADER : OAVIF_PNG_ERR_DECODE;
        const uint8_t* data = type + 4;
        const bool critical = !(type[0] & 0x20);
        const bool crc_ok = (uint32_t)crc32(crc32(0L, type, 4), data, clen) == be32(data + clen);
        pos += 12 + (size_t)clen;
        if (first) {
            if (memcmp(type, "IHDR", 4) != 0 || clen != 13 || !crc_ok) return OAVIF_PNG_ERR_HEADER;
            png.w = be32(data);
            png.h = be32(data + 4);
            png.depth = data[8];
            png.ctype = data[9];
            png.interlace = data[12];
            if (png.w == 0 || png.h == 0 || png.w > 0x7FFFFFFFu || png.h > 0x7FFFFFFFu) return OAVIF_PNG_ERR_HEADER;
            if (data[10] != 0 || data[11] != 0 || png.interlace > 1) return OAVIF_PNG_ERR_HEADER;
            const int d = png.depth;
            bool ok = false;
            switch (png.ctype) {
                case 0: ok = d == 1 || d == 2 || d == 4 || d == 8 || d == 16; break;
                case 3: ok = d == 1 || d == 2 || d == 4 || d == 8; break;
                case 2: case 4: case 6: ok = d == 8 || d == 16; break;
                default: ok = false;
            }
            if (!ok) return OAVIF_PNG_ERR_HEADER;
            first = false;
            continue;
        }
        if (!crc_ok) {
            if (critical) return OAVIF_PNG_ERR_DECODE;
            continue;  // ancillary chunk with a bad CRC: discarded
        }
        if (memcmp(type, "IDAT", 4) == 0) {
            if (idat_closed) return OAVIF_PNG_ERR_DECODE;  // IDAT chunks must be consecutive
            seen_idat = true;
            png.idat.push_back(Span{data, clen});
            continue;
        }
        if (seen_idat) idat_closed = true;
        if (memcmp(type, "IEND", 4) == 0) {
            end = true;
        } else if (memcmp(type, "PLTE", 4) == 0) {
            if (seen_idat || clen == 0 || clen % 3 != 0 || clen > 768) return OAVIF_PNG_ERR_DECODE;
            png.plte = Span{data, clen};
        } else if (memcmp(type, "iCCP", 4) == 0) {
            if (!seen_idat && !png.plte.p) png.iccp = Span{data, clen};
        } else if (critical) {
            return OAVIF_PNG_ERR_DECODE;  // unknown critical chunk
        }
    }
    if (png.idat.empty()) return OAVIF_PNG_ERR_DECODE;
    if (png.ctype == 3 && !png.plte.p) return OAVIF_PNG_ERR_DECODE;
    return OAVIF_PNG_OK;  // tRNS is an ancillary chunk this loader does not apply (decode flags 0, see the header)
}

// iCCP: keyword (1-79 bytes) NUL, compression method 0, zlib stream
int inflate_icc(const Png& png, std::vector<uint8_t>& out) {
    out.clear();
    if (!png.iccp.p) return OAVIF_PNG_OK;
    const uint8_t* p = png.iccp.p;
    size_t k = 0;
    while (k < png.iccp.n && k < 80 && p[k]) ++k;
    if (k == 0 || k >= 80 || k + 2 > png.iccp.n || p[k + 1] != 0) return OAVIF_PNG_OK;  // malformed: no profile
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit(&zs) != Z_OK) return OAVIF_PNG_ERR_OOM;
    zs.next_in = const_cast<uint8_t*>(p + k + 2);
    zs.avail_in = (uInt)(png.iccp.n - k - 2);
    uint8_t chunk[16384];
    int rc = Z_OK;
    while (rc == Z_OK) {
        zs.next_out = chunk;
        zs.avail_out = sizeof chunk;
        rc = inflate(&zs, Z_NO_FLUSH);
        if (rc == Z_OK || rc == Z_STREAM_END) {
            try {
                out.insert(out.end(), chunk, chunk + (sizeof chunk - zs.avail_out));
            } catch (...) {
                inflateEnd(&zs);
                return OAVIF_PNG_ERR_OOM;
            }
        }
        if (rc == Z_OK && zs.avail_in == 0 && zs.avail_out != 0) break;  // truncated stream
    }
    inflateEnd(&zs);
    if (rc != Z_STREAM_END) out.clear();  // a broken profile is dropped, not fatal (ancillary)
    return OAVIF_PNG_OK;
}

inline uint8_t paeth(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (uint8_t)(pa <= pb && pa <= pc ? a : (pb <= pc ? b : c));
}

#if defined(__SSE2__)
// The Sub / Average / Paeth filters of 3- and 4-byte pixels with the channels of ONE pixel side by side in an SSE2
// register (16-bit lanes): the dependence runs from pixel to pixel along the row, so a row cannot be vectorised
// across pixels, but its 3-4 channels and the three-way Paeth choice can -- no branches, ~10 instructions per
// pixel against ~35 for the byte-wise loop (a 1920x1080 RGB8 file with Paeth rows: 35 -> 7 ms on this image's CPU).
// Same arithmetic as the byte-wise form below (tests/test_png.py: every colour type, depth and filter against
// the expectation and against Pillow; tests/c/png_sanitize.cpp).
template <size_t BPP>
inline __m128i load_px(const uint8_t* p) {
    uint32_t v = 0;
    memcpy(&v, p, BPP);
    return _mm_unpacklo_epi8(_mm_cvtsi32_si128((int)v), _mm_setzero_si128());
}
template <size_t BPP>
inline void store_px(uint8_t* p, __m128i v16) {
    const uint32_t v = (uint32_t)_mm_cvtsi128_si32(_mm_packus_epi16(v16, v16));
    memcpy(p, &v, BPP);
}
inline __m128i abs16(__m128i x) { return _mm_max_epi16(x, _mm_sub_epi16(_mm_setzero_si128(), x)); }
template <size_t BPP>
void unfilter_sse2(int ftype, uint8_t* row, const uint8_t* prev, size_t n) {
    const __m128i lo8 = _mm_set1_epi16(0xff);
    __m128i a = _mm_setzero_si128(), c = _mm_setzero_si128();
    const size_t px = n / BPP;  // n is a whole number of pixels for 8-bit RGB / RGBA rows
    if (ftype == 1) {
        for (size_t i = 0; i < px; ++i, row += BPP) {
            a = _mm_and_si128(_mm_add_epi16(load_px<BPP>(row), a), lo8);
            store_px<BPP>(row, a);
        }
    } else if (ftype == 3) {
        for (size_t i = 0; i < px; ++i, row += BPP, prev += BPP) {
            const __m128i avg = _mm_srli_epi16(_mm_add_epi16(a, load_px<BPP>(prev)), 1);
            a = _mm_and_si128(_mm_add_epi16(load_px<BPP>(row), avg), lo8);
            store_px<BPP>(row, a);
        }
    } else {  // 4: Paeth; p = a + b - c, so |p - a| = |b - c|, |p - b| = |a - c|, |p - c| = |(b - c) + (a - c)|
        for (size_t i = 0; i < px; ++i, row += BPP, prev += BPP) {
            const __m128i b = load_px<BPP>(prev);
            const __m128i dbc = _mm_sub_epi16(b, c), dac = _mm_sub_epi16(a, c);
            const __m128i pa = abs16(dbc), pb = abs16(dac), pc = abs16(_mm_add_epi16(dbc, dac));
            const __m128i smallest = _mm_min_epi16(pc, _mm_min_epi16(pa, pb));
            // ties go to a, then b, then c (pa <= pb && pa <= pc ? a : pb <= pc ? b : c)
            const __m128i is_a = _mm_cmpeq_epi16(smallest, pa);
            const __m128i is_b = _mm_andnot_si128(is_a, _mm_cmpeq_epi16(smallest, pb));
            const __m128i pick = _mm_or_si128(_mm_and_si128(is_a, a),
                                              _mm_or_si128(_mm_and_si128(is_b, b),
                                                           _mm_andnot_si128(_mm_or_si128(is_a, is_b), c)));
            c = b;
            a = _mm_and_si128(_mm_add_epi16(load_px<BPP>(row), pick), lo8);
            store_px<BPP>(row, a);
        }
    }
}
#endif

// undo the filter of one row in place; `prev` = the unfiltered row above (zeros for a pass's first
// row).  BPP as a compile-time constant for the common pixel sizes: the left / upper-left
// neighbours then live in registers (the generic loop re-reads them through memory).
template <size_t BPP>
int unfilter_bpp(int ftype, uint8_t* row, const uint8_t* prev, size_t n, size_t bpp_dyn) {
    const size_t bpp = BPP ? BPP : bpp_dyn;
#if defined(__SSE2__)
    if ((BPP == 3 || BPP == 4) && (ftype == 1 || ftype == 3 || ftype == 4) && n % (BPP ? BPP : 1) == 0) {
        unfilter_sse2<BPP == 3 ? 3 : 4>(ftype, row, prev, n);
        return 0;
    }
#endif
    switch (ftype) {
        case 0: return 0;
        case 1:
            for (size_t i = bpp; i < n; ++i) row[i] = (uint8_t)(row[i] + row[i - bpp]);
            return 0;
        case 2:
            for (size_t i = 0; i < n; ++i) row[i] = (uint8_t)(row[i] + prev[i]);
            return 0;
        case 3: {
            const size_t head = bpp < n ? bpp : n;
            for (size_t i = 0; i < head; ++i) row[i] = (uint8_t)(row[i] + (prev[i] >> 1));
            for (size_t i = head; i < n; ++i) row[i] = (uint8_t)(row[i] + ((row[i - bpp] + prev[i]) >> 1));
            return 0;
        }
        case 4: {
            const size_t head = bpp < n ? bpp : n;
            for (size_t i = 0; i < head; ++i) row[i] = (uint8_t)(row[i] + prev[i]);  // paeth(0, b, 0) = b
            for (size_t i = head; i < n; ++i) row[i] = (uint8_t)(row[i] + paeth(row[i - bpp], prev[i], prev[i - bpp]));
            return 0;
        }
        default: return -1;
    }
}
int unfilter(int ftype, uint8_t* row, const uint8_t* prev, size_t n, size_t bpp) {
    switch (bpp) {
        case 1: return unfilter_bpp<1>(ftype, row, prev, n, bpp);
        case 3: return unfilter_bpp<3>(ftype, row, prev, n, bpp);
        case 4: return unfilter_bpp<4>(ftype, row, prev, n, bpp);
        default: return unfilter_bpp<0>(ftype, row, prev, n, bpp);
    }
}

// sample s (0-based, across the row) of an unfiltered row, for depth <= 8
inline unsigned sample8(const uint8_t* row, size_t s, int depth) {
    if (depth == 8) return row[s];
    const size_t bit = s * (size_t)depth;
    return (row[bit >> 3] >> (8 - depth - (int)(bit & 7))) & ((1u << depth) - 1u);
}

struct Expander {
    const Png& png;
    uint8_t* out;  // RGB8 / RGBA8 / RGBA16 (host-endian)
    // one pixel of an unfiltered row -> the output pixel (x, y); returns false on a bad palette index
    bool put(const uint8_t* row, size_t i, uint32_t x, uint32_t y) const {
        const size_t at = (size_t)y * png.w + x;
        const int d = png.depth;
        if (png.hbd()) {
            uint16_t* o = reinterpret_cast<uint16_t*>(out) + at * 4;
            const uint8_t* p = row + i * (size_t)png.samples() * 2;
            uint16_t r, g, b, a = 0xFFFF;
            switch (png.ctype) {
                case 0:
                    r = g = b = be16(p);
                    break;
                case 4:
                    r = g = b = be16(p);
                    a = be16(p + 2);
                    break;
                case 2:
                    r = be16(p), g = be16(p + 2), b = be16(p + 4);
                    break;
                default:
                    r = be16(p), g = be16(p + 2), b = be16(p + 4), a = be16(p + 6);
            }
            o[0] = r, o[1] = g, o[2] = b, o[3] = a;
            return true;
        }
        if (png.ctype == 2) {  // RGB8: copied as it is
            memcpy(out + at * 3, row + i * 3, 3);
            return true;
        }
        uint8_t* o = out + at * 4;
        switch (png.ctype) {
            case 0: {
                const unsigned v = sample8(row, i, d);
                const uint8_t g = (uint8_t)(d == 8 ? v : v * (255u / ((1u << d) - 1u)));
                o[0] = o[1] = o[2] = g;
                o[3] = 255;
                return true;
            }
            case 3: {
                const unsigned idx = sample8(row, i, d);
                if ((size_t)idx * 3 + 2 >= png.plte.n) return false;
                memcpy(o, png.plte.p + idx * 3, 3);
                o[3] = 255;
                return true;
            }
            case 4:
                o[0] = o[1] = o[2] = row[i * 2];
                o[3] = row[i * 2 + 1];
                return true;
            default:
                memcpy(o, row + i * 4, 4);
                return true;
        }
    }
};

size_t row_bytes(const Png& png, uint32_t w) { return ((size_t)w * (size_t)png.bits_per_pixel() + 7) / 8; }

struct Pass {
    uint32_t x0, y0, dx, dy;
};
const Pass kAdam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};

}  // namespace

namespace {
// sub-image sizes of the passes and the bytes of all filtered scanlines (1 filter byte + the row each)
struct Geometry {
    int npass;
    uint32_t pw[7], ph[7];
    uint64_t filtered_bytes;
};
Geometry geometry(const Png& png) {
    Geometry g;
    g.npass = png.interlace ? 7 : 1;
    g.filtered_bytes = 0;
    for (int k = 0; k < g.npass; ++k) {
        if (png.interlace) {
            const Pass& a = kAdam7[k];
            g.pw[k] = png.w > a.x0 ? (png.w - a.x0 + a.dx - 1) / a.dx : 0;
            g.ph[k] = png.h > a.y0 ? (png.h - a.y0 + a.dy - 1) / a.dy : 0;
        } else {
            g.pw[k] = png.w;
            g.ph[k] = png.h;
        }
        if (g.pw[k] && g.ph[k]) g.filtered_bytes += (uint64_t)g.ph[k] * (1 + row_bytes(png, g.pw[k]));
    }
    return g;
}

// the output geometry of a parsed file (+ its decompressed profile)
int describe(const Png& png, std::vector<uint8_t>& icc, oavif_png_info* out) {
    const uint64_t px = (uint64_t)png.w * png.h;
    const uint64_t bytes = px * png.out_channels() * (png.hbd() ? 2u : 1u);
    if (px > (1ull << 40) || bytes / png.out_channels() / (png.hbd() ? 2u : 1u) != px || bytes > (uint64_t)SIZE_MAX / 2)
        return OAVIF_PNG_ERR_SIZE;
    // A header that promises more scanline bytes than its IDAT data can inflate to is a lie (deflate
    // expands by at most 1032 : 1): fail here, before a caller sizes a buffer from the header -- a
    // 70-byte file claiming 60000 x 60000 pixels must not cost 10 GB.
    uint64_t idat_bytes = 0;
    for (const Span& c : png.idat) idat_bytes += c.n;
    if (geometry(png).filtered_bytes > idat_bytes * 1032u + 1024u) return OAVIF_PNG_ERR_DECODE;
    const int rc = inflate_icc(png, icc);
    if (rc) return rc;
    out->width = png.w;
    out->height = png.h;
    out->channels = png.out_channels();
    out->hbd = png.hbd() ? 1 : 0;
    out->data_bytes = (size_t)bytes;
    out->icc_bytes = icc.size();
    out->bit_depth = (uint32_t)png.depth;
    out->color_type = (uint32_t)png.ctype;
    out->interlaced = (uint32_t)png.interlace;
    return OAVIF_PNG_OK;
}
}  // namespace

extern "C" {

int oavif_png_info_from_memory(const uint8_t* png_bytes, size_t len, oavif_png_info* out) {
    if (!png_bytes || !out) return OAVIF_PNG_ERR_ARG;
    memset(out, 0, sizeof *out);
    try {
        Png png;
        const int rc = parse(png_bytes, len, png);
        if (rc) return rc;
        std::vector<uint8_t> icc;
        return describe(png, icc, out);
    } catch (...) {
        return OAVIF_PNG_ERR_OOM;
    }
}

int oavif_png_decode(const uint8_t* png_bytes, size_t len, uint8_t* out_pixels, size_t out_cap, uint8_t* out_icc,
                     size_t icc_cap) {
    if (!png_bytes || !out_pixels) return OAVIF_PNG_ERR_ARG;
    try {
        Png png;
        int rc = parse(png_bytes, len, png);  // one walk over the chunks (and their CRCs)
        if (rc) return rc;
        oavif_png_info info;
        memset(&info, 0, sizeof info);
        std::vector<uint8_t> icc;
        if ((rc = describe(png, icc, &info))) return rc;
        if (out_cap < info.data_bytes || (info.icc_bytes && out_icc && icc_cap < info.icc_bytes)) return OAVIF_PNG_ERR_SIZE;
        if (info.hbd && (reinterpret_cast<uintptr_t>(out_pixels) & 1u)) return OAVIF_PNG_ERR_ARG;  // u16 output
        if (out_icc && info.icc_bytes) memcpy(out_icc, icc.data(), icc.size());
        // The IDAT stream is inflated by inflate_fast.h (1.5-1.75 x zlib on PNG streams) into a sliding buffer:
        // 32 KB of history + a strip of at most kStripBytes (at least one row).  Each filtered scanline is copied
        // out of it -- for 8-bit RGB / RGBA straight into the caller's pixels, else into a row buffer --,
        // unfiltered against the row above and expanded: memory is the compressed stream + one strip + two rows
        // whatever the header claims, and a stream that ends early fails where it ends.  (Matches copy from the
        // FILTERED bytes, so rows are never unfiltered inside the window.)
        const Geometry geo = geometry(png);
        size_t max_rb = 0;
        for (int k = 0; k < geo.npass; ++k)
            if (geo.pw[k] && geo.ph[k]) max_rb = row_bytes(png, geo.pw[k]) > max_rb ? row_bytes(png, geo.pw[k]) : max_rb;
        size_t zbytes = 0;
        for (const auto& c : png.idat) zbytes += c.n;
        if (zbytes < 2) return OAVIF_PNG_ERR_DECODE;
        std::vector<uint8_t> z(zbytes + finf::kPad, 0);
        {
            size_t at = 0;
            for (const auto& c : png.idat) {
                memcpy(z.data() + at, c.p, c.n);
                at += c.n;
            }
        }
        // zlib wrapper (RFC 1950): deflate, window <= 32 KB, header check, no preset dictionary.  The Adler-32
        // behind the stream is not read: decoding stops at the last scanline, as it always did here.
        if ((z[0] & 0x0f) != 8 || (z[0] >> 4) > 7 || ((z[0] << 8) | z[1]) % 31 != 0 || (z[1] & 0x20)) return OAVIF_PNG_ERR_DECODE;
        constexpr size_t kStripBytes = 256 * 1024;
        const size_t strip_cap = (max_rb + 1) > kStripBytes ? (max_rb + 1) : kStripBytes;
        std::vector<uint8_t> win(finf::kWindow + strip_cap + (max_rb + 1) + finf::kOutMargin);
        std::vector<uint8_t> rowbuf(2 * (max_rb + 1));
        static thread_local finf::Stream zs;  // 13 KB of tables: not on the stack of a worker thread
        zs.init(z.data() + 2, zbytes - 2);
        size_t have = 0;     // valid bytes at the start of `win`
        size_t row_pos = 0;  // offset in `win` of the next filtered scanline that has not been taken
        bool ended = false;
        // make at least n bytes available at win[row_pos..]; false = the stream ended or broke first
        auto need = [&](size_t n) -> bool {
            while (have - row_pos < n) {
                if (ended) return false;
                // slide: keep the history a match may reach (kWindow) and everything not yet taken
                const size_t keep_from = row_pos < (have > finf::kWindow ? have - finf::kWindow : 0)
                                             ? row_pos : (have > finf::kWindow ? have - finf::kWindow : 0);
                if (keep_from) {
                    memmove(win.data(), win.data() + keep_from, have - keep_from);
                    have -= keep_from;
                    row_pos -= keep_from;
                }
                uint8_t* o = win.data() + have;
                const finf::Result r = finf::run(zs, win.data(), o, win.data() + win.size());
                const size_t got = (size_t)(o - (win.data() + have));
                have += got;
                if (r == finf::kError) return false;
                if (r == finf::kDone) ended = true;
                else if (got == 0) return false;  // no progress with room to spare: cannot happen; do not spin
            }
            return true;
        };
        const size_t bpp = (size_t)(png.bits_per_pixel() + 7) / 8;
        const Expander ex{png, out_pixels};
        int result = OAVIF_PNG_OK;
        const bool direct = !png.interlace && png.depth == 8 && (png.ctype == 2 || png.ctype == 6);  // rows go out as they are
        for (int k = 0; k < geo.npass && result == OAVIF_PNG_OK; ++k) {
            if (!geo.pw[k] || !geo.ph[k]) continue;
            const size_t rb = row_bytes(png, geo.pw[k]);
            uint8_t* cur = rowbuf.data();
            uint8_t* above = rowbuf.data() + max_rb + 1;
            memset(above, 0, rb);  // a pass's first row has zeros above it
            const uint8_t* prev = above;
            const Pass a = png.interlace ? kAdam7[k] : Pass{0, 0, 1, 1};
            for (uint32_t j = 0; j < geo.ph[k]; ++j) {
                if (!need(rb + 1)) {
                    result = OAVIF_PNG_ERR_DECODE;
                    break;
                }
                const uint8_t* frow = win.data() + row_pos;
                row_pos += rb + 1;
                const uint32_t y = a.y0 + j * a.dy;
                uint8_t* dst = direct ? out_pixels + (size_t)y * rb : cur;
                memcpy(dst, frow + 1, rb);
                if (unfilter(frow[0], dst, prev, rb, bpp)) {
                    result = OAVIF_PNG_ERR_DECODE;
                    break;
                }
                prev = dst;
                if (!direct) {
                    for (uint32_t i = 0; i < geo.pw[k]; ++i)
                        if (!ex.put(dst, i, a.x0 + i * a.dx, y)) {
                            result = OAVIF_PNG_ERR_DECODE;
                            break;
                        }
                    if (result != OAVIF_PNG_OK) break;
                    uint8_t* t = cur;  // the row just made is the row above the next one
                    cur = above;
                    above = t;
                }
            }
        }
        // data behind the last scanline is ignored; a stream that only "decoded" because the padding behind the
        // input reads as zeros is a truncated one
        if (result == OAVIF_PNG_OK && zs.overrun()) result = OAVIF_PNG_ERR_DECODE;
        if (result != OAVIF_PNG_OK) return result;
    } catch (const std::bad_alloc&) {
        return OAVIF_PNG_ERR_OOM;
    } catch (...) {
        return OAVIF_PNG_ERR_DECODE;
    }
    return OAVIF_PNG_OK;
}

}  // extern "C"
