// PIZ decompression for the OpenEXR scanline reader (images.cpp): the `exr` crate the reference reads its environment maps
// with (get_exr_image, src/core/loader.rs:374-390) accepts PIZ, the most common compression of HDR environment files.
//
// Restated from the published description of the format (OpenEXR "Technical Introduction" and the layout of
// ImfPizCompressor / ImfHuf / ImfWav as documented with the library), not from its source, which this image does not hold:
//   block = u16 minNonZero, u16 maxNonZero, bitmap bytes [minNonZero .. maxNonZero] of the 65536-bit "value occurs" map,
//           i32 length, Huffman stream (20-byte header: im, iM, table length, nBits, reserved; 6-bit code lengths with
//           zero-run escapes 59 .. 62 (2 .. 5 zeros) and 63 + 8 bits (6 .. 261 zeros); canonical codes; the symbol iM
//           is a run-length escape followed by an 8-bit repeat count)
//   data  = per channel (file order) `lines x width x (1 | 2)` 16-bit words, 2-D Haar-like wavelet (14-bit form when
//           the largest LUT index is < 16384, else the 16-bit modulo form), values mapped through the LUT of occurring
//           values, then re-interleaved into scanlines.
// PARITY UNPINNED: neither the reference nor this image contains a PIZ file or an OpenEXR library to write one; the tests
// (tests/test_exr.py) pair this decoder with an encoder written from the same description.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "host_scene.hpp"

namespace spt_host {

namespace {

constexpr int kHufEncBits = 16, kHufDecBits = 14;
constexpr int kHufEncSize = (1 << kHufEncBits) + 1, kHufDecSize = 1 << kHufDecBits, kHufDecMask = kHufDecSize - 1;
constexpr int kShortZeroRun = 59, kLongZeroRun = 63, kShortestLongRun = 2 + kLongZeroRun - kShortZeroRun;

[[noreturn]] void bad(const char* what) { throw HostError(SPT_HOST_ERR_PARSE, std::string("exr: bad PIZ data (") + what + ")"); }

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t c = 0;
    int lc = 0;
    uint32_t bits(int n) {
        while (lc < n) {
            if (p >= end) bad("code table runs past the block");
            c = (c << 8) | *p++;
            lc += 8;
        }
        lc -= n;
        return (uint32_t)((c >> lc) & ((1ull << n) - 1));
    }
};

struct DecEntry {
    uint32_t len = 0;      // short code: its length; 0 = long codes (or nothing) behind this prefix
    uint32_t lit = 0;      // short code: the symbol
    std::vector<uint32_t> longs;   // symbols whose code is longer than kHufDecBits and starts with this prefix
};

void huf_uncompress(const uint8_t* src, size_t n_src, uint16_t* out, size_t n_out) {
    if (n_src < 20) { if (n_out) bad("Huffman header"); return; }
    auto u32 = [&](size_t o) { uint32_t v; std::memcpy(&v, src + o, 4); return v; };
    const uint32_t im = u32(0), iM = u32(4), n_bits = u32(12);
    if (im >= (uint32_t)kHufEncSize || iM >= (uint32_t)kHufEncSize || im > iM) bad("symbol range");
    // code lengths
    std::vector<uint64_t> hcode((size_t)kHufEncSize, 0);
    BitReader br{src + 20, src + n_src};
    for (uint32_t s = im; s <= iM; ++s) {
        const uint32_t l = br.bits(6);
        hcode[s] = l;
        if (l == (uint32_t)kLongZeroRun) {
            uint32_t zerun = br.bits(8) + (uint32_t)kShortestLongRun;
            if (s + zerun > iM + 1) bad("zero run past the table");
            while (zerun--) hcode[s++] = 0;
            --s;
        } else if (l >= (uint32_t)kShortZeroRun) {
            uint32_t zerun = l - (uint32_t)kShortZeroRun + 2;
            if (s + zerun > iM + 1) bad("zero run past the table");
            while (zerun--) hcode[s++] = 0;
            --s;
        }
    }
    const uint8_t* data = br.p;
    // canonical codes: lengths -> (code << 6) | length
    {
        uint64_t n[59] = {0};
        for (int i = 0; i < kHufEncSize; ++i) n[hcode[(size_t)i]] += 1;
        uint64_t c = 0;
        for (int i = 58; i > 0; --i) {
            const uint64_t nc = (c + n[i]) >> 1;
            n[i] = c;
            c = nc;
        }
        for (int i = 0; i < kHufEncSize; ++i) {
            const uint64_t l = hcode[(size_t)i];
            if (l > 0) hcode[(size_t)i] = l | (n[l]++ << 6);
        }
    }
    // decoding table
    std::vector<DecEntry> dec((size_t)kHufDecSize);
    for (uint32_t s = im; s <= iM; ++s) {
        const uint64_t c = hcode[s] >> 6;
        const int l = (int)(hcode[s] & 63);
        if (c >> l) bad("code longer than its length");
        if (l > kHufDecBits) {
            DecEntry& e = dec[(size_t)(c >> (l - kHufDecBits))];
            if (e.len) bad("long code behind a short one");
            e.longs.push_back(s);
        } else if (l) {
            size_t first = (size_t)(c << (kHufDecBits - l));
            for (size_t i = (size_t)1 << (kHufDecBits - l); i > 0; --i, ++first) {
                DecEntry& e = dec[first];
                if (e.len || !e.longs.empty()) bad("ambiguous code");
                e.len = (uint32_t)l;
                e.lit = s;
            }
        }
    }
    // the stream
    if ((size_t)(src + n_src - data) < ((size_t)n_bits + 7) / 8) bad("bit stream shorter than its header says");
    const uint8_t* in = data;
    const uint8_t* ie = data + ((size_t)n_bits + 7) / 8;
    uint64_t c = 0;
    int lc = 0;
    size_t o = 0;
    const uint32_t rlc = iM;
    auto emit = [&](uint32_t sym) {
        if (sym == rlc) {
            if (lc < 8) {
                if (in >= ie) bad("run length past the stream");
                c = (c << 8) | *in++;
                lc += 8;
            }
            lc -= 8;
            uint32_t cs = (uint32_t)((c >> lc) & 0xff);
            if (o == 0 || o + cs > n_out) bad("run past the output");
            const uint16_t v = out[o - 1];
            while (cs--) out[o++] = v;
        } else {
            if (o >= n_out) bad("more symbols than samples");
            out[o++] = (uint16_t)sym;
        }
    };
    while (in < ie) {
        c = (c << 8) | *in++;
        lc += 8;
        while (lc >= kHufDecBits) {
            const DecEntry& e = dec[(size_t)((c >> (lc - kHufDecBits)) & (uint64_t)kHufDecMask)];
            if (e.len) {
                lc -= (int)e.len;
                emit(e.lit);
            } else {
                if (e.longs.empty()) bad("unknown code");
                bool found = false;
                for (uint32_t s : e.longs) {
                    const int l = (int)(hcode[s] & 63);
                    while (lc < l && in < ie) {
                        c = (c << 8) | *in++;
                        lc += 8;
                    }
                    if (lc >= l && (hcode[s] >> 6) == ((c >> (lc - l)) & ((1ull << l) - 1))) {
                        lc -= l;
                        emit(s);
                        found = true;
                        break;
                    }
                }
                if (!found) bad("unknown long code");
            }
        }
    }
    // the bits left over behind the last whole byte
    const int pad = (8 - (int)(n_bits & 7u)) & 7;
    c >>= pad;
    lc -= pad;
    while (lc > 0) {
        const DecEntry& e = dec[(size_t)((c << (kHufDecBits - lc)) & (uint64_t)kHufDecMask)];
        if (!e.len || (int)e.len > lc) bad("truncated code");
        lc -= (int)e.len;
        emit(e.lit);
    }
    if (o != n_out) bad("fewer symbols than samples");
}

// inverse wavelet steps
inline void wdec14(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
    const int ls = (int16_t)l, hs = (int16_t)h;
    const int ai = ls + (hs & 1) + (hs >> 1);
    a = (uint16_t)(int16_t)ai;
    b = (uint16_t)(int16_t)(ai - hs);
}
inline void wdec16(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
    const int m = l, d = h;
    const int bb = (m - (d >> 1)) & 0xffff;
    const int aa = (d + bb - 0x8000) & 0xffff;
    b = (uint16_t)bb;
    a = (uint16_t)aa;
}
void wav2_decode(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t mx) {
    const bool w14 = mx < (1 << 14);
    const int n = nx > ny ? ny : nx;
    int p = 1;
    while (p <= n) p <<= 1;
    p >>= 1;
    int p2 = p;
    p >>= 1;
    auto dec = [&](uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) { if (w14) wdec14(l, h, a, b); else wdec16(l, h, a, b); };
    // offsets instead of the customary pointers: `end` can lie before the start of the plane (ny < p2), which is not a
    // pointer one may form
    while (p >= 1) {
        const int64_t oy1 = (int64_t)oy * p, oy2 = (int64_t)oy * p2, ox1 = (int64_t)ox * p, ox2 = (int64_t)ox * p2;
        const int64_t ey = (int64_t)oy * (ny - p2);
        int64_t py = 0;
        for (; py <= ey; py += oy2) {
            int64_t px = py;
            const int64_t ex = py + (int64_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t& v00 = in[px];
                uint16_t& v01 = in[px + ox1];
                uint16_t& v10 = in[px + oy1];
                uint16_t& v11 = in[px + oy1 + ox1];
                uint16_t i00, i01, i10, i11;
                dec(v00, v10, i00, i10);
                dec(v01, v11, i01, i11);
                dec(i00, i01, v00, v01);
                dec(i10, i11, v10, v11);
            }
            if (nx & p) {
                uint16_t i00, i10;
                dec(in[px], in[px + oy1], i00, i10);
                in[px + oy1] = i10;
                in[px] = i00;
            }
        }
        if (ny & p) {
            int64_t px = py;
            const int64_t ex = py + (int64_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t i00, i01;
                dec(in[px], in[px + ox1], i00, i01);
                in[px + ox1] = i01;
                in[px] = i00;
            }
        }
        p2 = p;
        p >>= 1;
    }
}

}  // namespace

// One PIZ block of `lines` scanlines: `words_per_pixel[c]` is 1 for a HALF channel, 2 for FLOAT / UINT (file order);
// `raw` receives the scanline-interleaved little-endian samples, exactly what an uncompressed block holds.
void exr_piz_decode(const uint8_t* src, size_t size, const std::vector<int>& words_per_pixel, int64_t width, int64_t lines, uint8_t* raw, size_t raw_bytes) {
    size_t total = 0;
    for (int wpp : words_per_pixel) total += (size_t)wpp * (size_t)width * (size_t)lines;
    if (total * 2 != raw_bytes) bad("block size");
    if (size < 4) bad("header");
    uint16_t min_nz, max_nz;
    std::memcpy(&min_nz, src, 2);
    std::memcpy(&max_nz, src + 2, 2);
    size_t pos = 4;
    std::vector<uint8_t> bitmap(8192, 0);
    if (min_nz <= max_nz) {
        if (max_nz >= 8192) bad("bitmap range");
        const size_t n = (size_t)max_nz - min_nz + 1;
        if (pos + n > size) bad("bitmap");
        std::memcpy(&bitmap[min_nz], src + pos, n);
        pos += n;
    }
    std::vector<uint16_t> lut(65536, 0);
    uint32_t k = 0;
    for (uint32_t i = 0; i < 65536; ++i)
        if (i == 0 || (bitmap[i >> 3] & (1u << (i & 7)))) lut[k++] = (uint16_t)i;
    const uint16_t max_value = (uint16_t)(k - 1);
    if (pos + 4 > size) bad("length");
    int32_t length;
    std::memcpy(&length, src + pos, 4);
    pos += 4;
    if (length < 0 || pos + (size_t)length > size) bad("Huffman length");
    std::vector<uint16_t> buf(total);
    huf_uncompress(src + pos, (size_t)length, buf.data(), total);
    // per channel: inverse wavelet on each 16-bit plane of the channel
    size_t start = 0;
    std::vector<size_t> chan_start(words_per_pixel.size());
    for (size_t c = 0; c < words_per_pixel.size(); ++c) {
        const int wpp = words_per_pixel[c];
        chan_start[c] = start;
        for (int j = 0; j < wpp; ++j) wav2_decode(buf.data() + start + j, (int)width, wpp, (int)lines, (int)width * wpp, max_value);
        start += (size_t)wpp * (size_t)width * (size_t)lines;
    }
    for (uint16_t& v : buf) v = lut[v];
    // back to scanlines: line by line, channel by channel
    uint8_t* dst = raw;
    for (int64_t l = 0; l < lines; ++l)
        for (size_t c = 0; c < words_per_pixel.size(); ++c) {
            const size_t n = (size_t)words_per_pixel[c] * (size_t)width;
            std::memcpy(dst, buf.data() + chan_start[c] + (size_t)l * n, n * 2);
            dst += n * 2;
        }
}

}  // namespace spt_host
