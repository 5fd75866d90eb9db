// JPEG (ITU T.81) decoder and baseline encoder for the host side: image textures named by scene files / glTF
// (reference: `image::open` in src/core/loader.rs:366-371, src/loader/gltf.rs) and `-o x.jpg` output
// (`RgbImage::save`, src/renderer/pt.rs:292-294).
//
// Decoder: 8-bit baseline / extended-sequential / progressive Huffman files (SOF0, SOF1, SOF2), 1 or 3 components,
// any sampling factors, restart intervals.  The arithmetic is the IJG one - the "islow" integer IDCT, triangle
// ("fancy") chroma upsampling for 2:1 ratios, 16-bit fixed-point YCbCr -> RGB - so the texels equal what libjpeg /
// libjpeg-turbo produce (tests/test_jpeg.py compares with Pillow).  The reference decodes with the Rust `jpeg-decoder`
// crate, whose IDCT and colour conversion are ports of the same code but are not guaranteed to round identically:
// JPEG texels are "parity unpinned" by +-1 code value against the reference (PNG texels are exact).
//
// Encoder: baseline, 4:4:4, the Annex K tables scaled to quality 75 (the `image` crate's JpegEncoder default).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "host_scene.hpp"

namespace {

using spt_host::HostError;

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    bool present = false;
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    int32_t maxcode[18];
    int32_t valptr[17];
    int32_t mincode[17];
    uint8_t look_len[256];   // 8-bit lookahead: code length (0 = longer than 8 bits)
    uint8_t look_val[256];
    void build() {
        int32_t code = 0, k = 0;
        std::memset(look_len, 0, sizeof look_len);
        for (int len = 1; len <= 16; ++len) {
            valptr[len] = k;
            mincode[len] = code;
            for (int i = 0; i < bits[len]; ++i, ++k, ++code) {
                if (len <= 8) {
                    const int first = code << (8 - len), n = 1 << (8 - len);
                    for (int j = 0; j < n && first + j < 256; ++j) { look_len[first + j] = (uint8_t)len; look_val[first + j] = vals[k]; }
                }
            }
            maxcode[len] = bits[len] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
    }
};

struct BitReader {
    const uint8_t* d;
    size_t n, p;
    uint32_t acc = 0;
    int cnt = 0;
    int marker = 0;   // a marker met in the entropy-coded data (zeros are fed from there on)
    void fill() {
        while (cnt <= 24) {
            uint32_t byte = 0;
            if (!marker && p < n) {
                byte = d[p];
                if (byte == 0xff) {
                    size_t q = p + 1;
                    while (q < n && d[q] == 0xff) ++q;   // fill bytes
                    if (q < n && d[q] == 0) { p = q + 1; }
                    else { marker = q < n ? d[q] : 0xd9; byte = 0; p = q < n ? q + 1 : n; }
                } else {
                    ++p;
                }
            }
            acc |= byte << (24 - cnt);
            cnt += 8;
        }
    }
    uint32_t peek(int k) { if (cnt < k) fill(); return acc >> (32 - k); }
    void skip(int k) { acc <<= k; cnt -= k; }
    int32_t get(int k) { if (k == 0) return 0; uint32_t v = peek(k); skip(k); return (int32_t)v; }
    int32_t bit() { return get(1); }
    void reset() { acc = 0; cnt = 0; marker = 0; }
};

inline int32_t extend(int32_t v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

int decode_sym(BitReader& br, const Huff& h, const std::string& label) {
    uint32_t look = br.peek(16);
    int len = h.look_len[look >> 8];
    if (len) { br.skip(len); return h.look_val[look >> 8]; }
    for (len = 9; len <= 16; ++len) {
        const int32_t code = (int32_t)(look >> (16 - len));   // the first `len` bits
        if (code <= h.maxcode[len]) {
            br.skip(len);
            return h.vals[(h.valptr[len] + code - h.mincode[len]) & 255];
        }
    }
    throw HostError(SPT_HOST_ERR_PARSE, "jpeg '" + label + "': bad Huffman code");
}

struct Comp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int bw = 0, bh = 0;         // blocks per row / column, padded to whole MCUs
    int cw = 0, ch = 0;         // samples that belong to the image: ceil(W h / hmax), ceil(H v / vmax)
    int32_t pred = 0;
    std::vector<int16_t> coef;  // bw * bh * 64, natural order
    std::vector<uint8_t> plane; // (bw * 8) x (bh * 8)
};

// jidctint.c (IJG "islow"): 13-bit constants, 2 extra bits between the passes
void idct_islow(const int16_t* in, const uint16_t* q, uint8_t* out, int stride) {
    constexpr int CB = 13, P1 = 2;
    constexpr int64_t F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299,
                      F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
    auto descale = [](int64_t x, int n) { return (x + ((int64_t)1 << (n - 1))) >> n; };   // 64-bit: a corrupt file cannot overflow it, a valid one gives the 32-bit values
    int64_t ws[64];
    for (int c = 0; c < 8; ++c) {
        const int16_t* ip = in + c;
        const uint16_t* qp = q + c;
        int64_t* wp = ws + c;
        if (!ip[8] && !ip[16] && !ip[24] && !ip[32] && !ip[40] && !ip[48] && !ip[56]) {
            int64_t dc = (int64_t)ip[0] * qp[0] * (1 << P1);
            for (int r = 0; r < 8; ++r) wp[8 * r] = dc;
            continue;
        }
        int64_t z2 = (int64_t)ip[16] * qp[16], z3 = (int64_t)ip[48] * qp[48];
        int64_t z1 = (z2 + z3) * F0_541;
        int64_t tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
        z2 = (int64_t)ip[0] * qp[0]; z3 = (int64_t)ip[32] * qp[32];
        int64_t tmp0 = (z2 + z3) * (1 << CB), tmp1 = (z2 - z3) * (1 << CB);
        int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = (int64_t)ip[56] * qp[56]; tmp1 = (int64_t)ip[40] * qp[40]; tmp2 = (int64_t)ip[24] * qp[24]; tmp3 = (int64_t)ip[8] * qp[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3, z5 = (z3 + z4) * F1_175;
        tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
        z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        wp[0] = descale(tmp10 + tmp3, CB - P1);  wp[56] = descale(tmp10 - tmp3, CB - P1);
        wp[8] = descale(tmp11 + tmp2, CB - P1);  wp[48] = descale(tmp11 - tmp2, CB - P1);
        wp[16] = descale(tmp12 + tmp1, CB - P1); wp[40] = descale(tmp12 - tmp1, CB - P1);
        wp[24] = descale(tmp13 + tmp0, CB - P1); wp[32] = descale(tmp13 - tmp0, CB - P1);
    }
    auto clamp8 = [](int64_t x) { x += 128; return (uint8_t)(x < 0 ? 0 : x > 255 ? 255 : x); };
    for (int r = 0; r < 8; ++r) {
        const int64_t* wp = ws + 8 * r;
        uint8_t* op = out + (size_t)r * stride;
        int64_t z2 = wp[2], z3 = wp[6];
        int64_t z1 = (z2 + z3) * F0_541;
        int64_t tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
        int64_t tmp0 = (wp[0] + wp[4]) * (1 << CB), tmp1 = (wp[0] - wp[4]) * (1 << CB);
        int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = wp[7]; tmp1 = wp[5]; tmp2 = wp[3]; tmp3 = wp[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3, z5 = (z3 + z4) * F1_175;
        tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
        z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        constexpr int S = CB + P1 + 3;
        op[0] = clamp8(descale(tmp10 + tmp3, S)); op[7] = clamp8(descale(tmp10 - tmp3, S));
        op[1] = clamp8(descale(tmp11 + tmp2, S)); op[6] = clamp8(descale(tmp11 - tmp2, S));
        op[2] = clamp8(descale(tmp12 + tmp1, S)); op[5] = clamp8(descale(tmp12 - tmp1, S));
        op[3] = clamp8(descale(tmp13 + tmp0, S)); op[4] = clamp8(descale(tmp13 - tmp0, S));
    }
}

struct Decoder {
    const std::vector<uint8_t>& d;
    const std::string& label;
    size_t p = 2;
    int W = 0, H = 0, ncomp = 0, hmax = 1, vmax = 1, restart = 0;
    bool progressive = false, have_frame = false;
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    Comp comp[3];
    int mcux = 0, mcuy = 0;

    Decoder(const std::vector<uint8_t>& bytes, const std::string& l) : d(bytes), label(l) {}
    [[noreturn]] void bad(const std::string& what) const { throw HostError(SPT_HOST_ERR_PARSE, "jpeg '" + label + "': " + what); }
    [[noreturn]] void unsupported(const std::string& what) const { throw HostError(SPT_HOST_ERR_UNSUPPORTED, "jpeg '" + label + "': " + what); }
    uint32_t u8(size_t at) const { if (at >= d.size()) bad("truncated file"); return d[at]; }
    uint32_t u16(size_t at) const { return (u8(at) << 8) | u8(at + 1); }

    void read_dqt(size_t at, size_t end) {
        while (at < end) {
            const uint32_t pq = u8(at) >> 4, tq = u8(at) & 15;
            ++at;
            if (tq > 3 || pq > 1) bad("bad DQT");
            for (int i = 0; i < 64; ++i) {
                qt[tq][kZigzag[i]] = (uint16_t)(pq ? u16(at) : u8(at));
                at += pq ? 2 : 1;
            }
            qt_present[tq] = true;
        }
        if (at != end) bad("bad DQT length");
    }
    void read_dht(size_t at, size_t end) {
        while (at < end) {
            const uint32_t tc = u8(at) >> 4, th = u8(at) & 15;
            ++at;
            if (tc > 1 || th > 3) bad("bad DHT");
            Huff& h = tc ? ac[th] : dc[th];
            int total = 0;
            h.bits[0] = 0;
            for (int i = 1; i <= 16; ++i) { h.bits[i] = (uint8_t)u8(at++); total += h.bits[i]; }
            if (total > 256) bad("bad DHT");
            for (int i = 0; i < total; ++i) h.vals[i] = (uint8_t)u8(at++);
            h.build();
        }
        if (at != end) bad("bad DHT length");
    }
    void read_sof(size_t at, size_t end) {
        if (have_frame) bad("more than one frame");
        if (u8(at) != 8) unsupported("only 8-bit samples are decoded");
        H = (int)u16(at + 1); W = (int)u16(at + 3); ncomp = (int)u8(at + 5);
        if (W <= 0 || H <= 0) unsupported("zero-sized frame (DNL)");
        if ((uint64_t)W * H > 0x3fffffffull) unsupported("image too large");
        if (ncomp != 1 && ncomp != 3) unsupported("only grayscale and 3-component (YCbCr) files are decoded");
        if (at + 6 + 3 * (size_t)ncomp != end) bad("bad SOF length");
        for (int c = 0; c < ncomp; ++c) {
            Comp& k = comp[c];
            k.id = (int)u8(at + 6 + 3 * c);
            k.h = (int)(u8(at + 7 + 3 * c) >> 4); k.v = (int)(u8(at + 7 + 3 * c) & 15);
            k.tq = (int)u8(at + 8 + 3 * c);
            if (k.h < 1 || k.h > 4 || k.v < 1 || k.v > 4 || k.tq > 3) bad("bad component");
            hmax = std::max(hmax, k.h); vmax = std::max(vmax, k.v);
        }
        if (ncomp == 1) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; }   // a single component is never subsampled
        mcux = (W + 8 * hmax - 1) / (8 * hmax); mcuy = (H + 8 * vmax - 1) / (8 * vmax);
        for (int c = 0; c < ncomp; ++c) {
            Comp& k = comp[c];
            k.bw = mcux * k.h; k.bh = mcuy * k.v;
            k.cw = (W * k.h + hmax - 1) / hmax; k.ch = (H * k.v + vmax - 1) / vmax;
            k.coef.assign((size_t)k.bw * k.bh * 64, 0);
        }
        have_frame = true;
    }

    // one scan: header at `at` (after the length), entropy-coded data follows at `end`; returns the next marker position
    size_t read_scan(size_t at, size_t end) {
        if (!have_frame) bad("SOS before SOF");
        const int ns = (int)u8(at);
        if (ns < 1 || ns > ncomp || at + 1 + 2 * (size_t)ns + 3 != end) bad("bad SOS");
        Comp* sc[3];
        for (int i = 0; i < ns; ++i) {
            const int id = (int)u8(at + 1 + 2 * i);
            sc[i] = nullptr;
            for (int c = 0; c < ncomp; ++c) if (comp[c].id == id) sc[i] = &comp[c];
            if (!sc[i]) bad("SOS names an unknown component");
            sc[i]->td = (int)(u8(at + 2 + 2 * i) >> 4); sc[i]->ta = (int)(u8(at + 2 + 2 * i) & 15);
            if (sc[i]->td > 3 || sc[i]->ta > 3) bad("bad table index");
        }
        const int Ss = (int)u8(end - 3), Se = (int)u8(end - 2), Ah = (int)(u8(end - 1) >> 4), Al = (int)(u8(end - 1) & 15);
        if (progressive) {
            if (Ss > Se || Se > 63 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || Al > 13) bad("bad progressive scan parameters");
        } else if (Ss != 0 || Se != 63 || Ah != 0 || Al != 0) {
            bad("bad sequential scan parameters");
        }
        for (int i = 0; i < ns; ++i) {
            if ((!progressive || Ss == 0) && Ah == 0 && !dc[sc[i]->td].present) bad("missing DC Huffman table");
            if ((!progressive || Ss > 0) && !ac[sc[i]->ta].present) bad("missing AC Huffman table");
        }
        BitReader br{d.data(), d.size(), end};
        int32_t eobrun = 0;
        auto decode_block = [&](Comp& k, int16_t* blk) {
            if (!progressive) {
                const int t = decode_sym(br, dc[k.td], label);
                if (t > 15) bad("bad DC category");
                k.pred = (int32_t)((uint32_t)k.pred + (uint32_t)(t ? extend(br.get(t), t) : 0));   // wraps on corrupt data instead of overflowing
                blk[0] = (int16_t)k.pred;
                const Huff& h = ac[k.ta];
                for (int i = 1; i < 64;) {
                    const int rs = decode_sym(br, h, label), r = rs >> 4, s = rs & 15;
                    if (s == 0) { if (r == 15) { i += 16; continue; } break; }
                    i += r;
                    if (i > 63) bad("AC run past the block");
                    blk[kZigzag[i]] = (int16_t)extend(br.get(s), s);
                    ++i;
                }
                return;
            }
            if (Ss == 0) {
                if (Ah == 0) {
                    const int t = decode_sym(br, dc[k.td], label);
                    if (t > 15) bad("bad DC category");
                    k.pred = (int32_t)((uint32_t)k.pred + (uint32_t)(t ? extend(br.get(t), t) : 0));
                    blk[0] = (int16_t)((uint32_t)k.pred << Al);
                } else if (br.bit()) {
                    blk[0] = (int16_t)(blk[0] | (1 << Al));
                }
                return;
            }
            const Huff& h = ac[k.ta];
            if (Ah == 0) {
                if (eobrun > 0) { --eobrun; return; }
                for (int i = Ss; i <= Se;) {
                    const int rs = decode_sym(br, h, label), r = rs >> 4, s = rs & 15;
                    if (s) {
                        i += r;
                        if (i > 63) bad("AC run past the block");
                        blk[kZigzag[i]] = (int16_t)((uint32_t)extend(br.get(s), s) << Al);
                        ++i;
                    } else if (r == 15) {
                        i += 16;
                    } else {
                        eobrun = (1 << r) - 1;
                        if (r) eobrun += br.get(r);
                        break;
                    }
                }
                return;
            }
            // successive-approximation refinement of AC coefficients (jdphuff.c decode_mcu_AC_refine)
            const int32_t p1 = 1 << Al, m1 = -(1 << Al);
            int i = Ss;
            auto refine = [&](int16_t& c) {
                if (br.bit() && (c & p1) == 0) c = (int16_t)(c + (c >= 0 ? p1 : m1));
            };
            if (eobrun == 0) {
                for (; i <= Se; ++i) {
                    const int rs = decode_sym(br, h, label);
                    int r = rs >> 4, s = rs & 15;
                    int32_t val = 0;
                    if (s) {
                        if (s != 1) bad("bad refinement code");
                        val = br.bit() ? p1 : m1;
                    } else if (r != 15) {
                        eobrun = 1 << r;
                        if (r) eobrun += br.get(r);
                        break;
                    }
                    for (; i <= Se; ++i) {
                        int16_t& c = blk[kZigzag[i]];
                        if (c != 0) refine(c);
                        else if (--r < 0) break;
                    }
                    if (val && i <= Se) blk[kZigzag[i]] = (int16_t)val;
                }
            }
            if (eobrun > 0) {
                for (; i <= Se; ++i) {
                    int16_t& c = blk[kZigzag[i]];
                    if (c != 0) refine(c);
                }
                --eobrun;
            }
        };
        auto restart_here = [&](int expected) {
            // the marker sits at the next byte boundary; fill() stopped in front of it
            br.fill();
            if (!br.marker) {   // skip to the marker (tolerates trailing bits)
                while (br.p + 1 < d.size() && !(d[br.p] == 0xff && d[br.p + 1] >= 0xd0 && d[br.p + 1] <= 0xd7)) ++br.p;
                if (br.p + 1 >= d.size()) bad("missing restart marker");
                br.marker = d[br.p + 1];
                br.p += 2;
            }
            if (br.marker != 0xd0 + (expected & 7)) bad("restart markers out of order");
            br.reset();
            for (int c = 0; c < ncomp; ++c) comp[c].pred = 0;
            eobrun = 0;
        };
        for (int c = 0; c < ncomp; ++c) comp[c].pred = 0;
        int unit = 0, rst = 0;
        auto tick = [&](bool last) {
            ++unit;
            if (restart && unit % restart == 0 && !last) restart_here(rst++);
        };
        if (ns == 1) {
            // non-interleaved: the blocks that cover the component's own samples, row by row
            Comp& k = *sc[0];
            const int nbx = (k.cw + 7) / 8, nby = (k.ch + 7) / 8;
            for (int by = 0; by < nby; ++by)
                for (int bx = 0; bx < nbx; ++bx) {
                    decode_block(k, &k.coef[((size_t)by * k.bw + bx) * 64]);
                    tick(by == nby - 1 && bx == nbx - 1);
                }
        } else {
            for (int my = 0; my < mcuy; ++my)
                for (int mx = 0; mx < mcux; ++mx) {
                    for (int i = 0; i < ns; ++i) {
                        Comp& k = *sc[i];
                        for (int v = 0; v < k.v; ++v)
                            for (int hh = 0; hh < k.h; ++hh)
                                decode_block(k, &k.coef[((size_t)(my * k.v + v) * k.bw + (mx * k.h + hh)) * 64]);
                    }
                    tick(my == mcuy - 1 && mx == mcux - 1);
                }
        }
        // position of the next marker
        br.fill();
        if (br.marker) {
            size_t q = br.p;   // one past the marker byte
            return q - 2;
        }
        size_t q = br.p;
        while (q + 1 < d.size() && !(d[q] == 0xff && d[q + 1] != 0 && d[q + 1] != 0xff && !(d[q + 1] >= 0xd0 && d[q + 1] <= 0xd7))) ++q;
        return q;
    }

    void reconstruct() {
        for (int c = 0; c < ncomp; ++c) {
            Comp& k = comp[c];
            if (!qt_present[k.tq]) bad("missing quantisation table");
            k.plane.assign((size_t)k.bw * 8 * k.bh * 8, 0);
            const int stride = k.bw * 8;
            for (int by = 0; by < k.bh; ++by)
                for (int bx = 0; bx < k.bw; ++bx)
                    idct_islow(&k.coef[((size_t)by * k.bw + bx) * 64], qt[k.tq], &k.plane[(size_t)by * 8 * stride + bx * 8], stride);
        }
    }

    // full-resolution plane of a component: jdsample.c (fancy upsampling for 2:1, replication otherwise)
    std::vector<uint8_t> upsample(const Comp& k) const {
        const int stride = k.bw * 8;
        std::vector<uint8_t> out((size_t)W * H);
        const int hr = hmax / k.h, vr = vmax / k.v;
        if (hmax % k.h || vmax % k.v) unsupported("fractional sampling ratios");
        auto in_row = [&](int r) { return &k.plane[(size_t)std::min(std::max(r, 0), k.ch - 1) * stride]; };
        if (hr == 1 && vr == 1) {
            for (int y = 0; y < H; ++y) std::memcpy(&out[(size_t)y * W], in_row(y), (size_t)W);
        } else if (hr == 2 && vr == 1) {
            std::vector<uint8_t> row((size_t)k.cw * 2);
            for (int y = 0; y < H; ++y) {
                const uint8_t* in = in_row(y);
                const int n = k.cw;
                if (n == 1) { row[0] = row[1] = in[0]; }
                else {
                    row[0] = in[0];
                    row[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
                    for (int i = 1; i < n - 1; ++i) {
                        const int v = in[i] * 3;
                        row[2 * i] = (uint8_t)((v + in[i - 1] + 1) >> 2);
                        row[2 * i + 1] = (uint8_t)((v + in[i + 1] + 2) >> 2);
                    }
                    row[2 * n - 2] = (uint8_t)((in[n - 1] * 3 + in[n - 2] + 1) >> 2);
                    row[2 * n - 1] = in[n - 1];
                }
                std::memcpy(&out[(size_t)y * W], row.data(), (size_t)W);
            }
        } else if (hr == 2 && vr == 2) {
            std::vector<uint8_t> row((size_t)k.cw * 2);
            const int n = k.cw;
            for (int y = 0; y < H; ++y) {
                const int r = y >> 1;
                const uint8_t* in0 = in_row(r);
                const uint8_t* in1 = (y & 1) ? in_row(r + 1) : in_row(r - 1);   // the nearer neighbour row (edge rows repeat)
                if (n == 1) {
                    const int s = in0[0] * 3 + in1[0];
                    row[0] = (uint8_t)((s * 4 + 8) >> 4);
                    row[1] = (uint8_t)((s * 4 + 7) >> 4);
                } else {
                    int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
                    row[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
                    row[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                    lastcol = thiscol; thiscol = nextcol;
                    for (int i = 1; i < n - 1; ++i) {
                        nextcol = in0[i + 1] * 3 + in1[i + 1];
                        row[2 * i] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
                        row[2 * i + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                        lastcol = thiscol; thiscol = nextcol;
                    }
                    row[2 * n - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
                    row[2 * n - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
                }
                std::memcpy(&out[(size_t)y * W], row.data(), (size_t)W);
            }
        } else if (hr == 1 && vr == 2) {
            for (int y = 0; y < H; ++y) {
                const int r = y >> 1;
                const uint8_t* in0 = in_row(r);
                const uint8_t* in1 = (y & 1) ? in_row(r + 1) : in_row(r - 1);
                const int bias = (y & 1) ? 2 : 1;
                for (int x = 0; x < W; ++x) out[(size_t)y * W + x] = (uint8_t)((in0[x] * 3 + in1[x] + bias) >> 2);
            }
        } else {
            for (int y = 0; y < H; ++y) {
                const uint8_t* in = in_row(y / vr);
                for (int x = 0; x < W; ++x) out[(size_t)y * W + x] = in[std::min(x / hr, k.cw - 1)];
            }
        }
        return out;
    }

    void run(uint32_t* width, uint32_t* height, std::vector<uint32_t>* texels) {
        if (d.size() < 4 || d[0] != 0xff || d[1] != 0xd8) bad("not a JPEG file");
        bool done = false;
        int adobe_transform = -1;
        while (!done) {
            while (p < d.size() && d[p] != 0xff) ++p;   // tolerate garbage between segments
            while (p < d.size() && d[p] == 0xff) ++p;
            if (p >= d.size()) break;
            const uint32_t m = d[p++];
            if (m == 0xd9) break;
            if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;
            const size_t len = u16(p);
            if (len < 2 || p + len > d.size()) bad("bad segment length");
            const size_t at = p + 2, end = p + len;
            switch (m) {
            case 0xdb: read_dqt(at, end); break;
            case 0xc4: read_dht(at, end); break;
            case 0xc0: case 0xc1: read_sof(at, end); break;
            case 0xc2: progressive = true; read_sof(at, end); break;
            case 0xdd: if (len != 4) bad("bad DRI"); restart = (int)u16(at); break;
            case 0xee: if (len >= 14 && std::memcmp(&d[at], "Adobe", 5) == 0) adobe_transform = (int)d[at + 11]; break;
            case 0xda: p = read_scan(at, end); continue;
            case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
                unsupported("lossless / hierarchical / arithmetic-coded files are not decoded");
            default: break;   // APPn, COM, ...
            }
            p = end;
        }
        if (!have_frame) bad("no frame");
        reconstruct();
        *width = (uint32_t)W; *height = (uint32_t)H;
        texels->assign((size_t)W * H, 0);
        if (ncomp == 1) {
            const std::vector<uint8_t> y = upsample(comp[0]);
            for (size_t i = 0; i < y.size(); ++i) (*texels)[i] = 0xff000000u | ((uint32_t)y[i] << 16) | ((uint32_t)y[i] << 8) | y[i];
            return;
        }
        const std::vector<uint8_t> a = upsample(comp[0]), b = upsample(comp[1]), c = upsample(comp[2]);
        const bool rgb = adobe_transform == 0 || (adobe_transform < 0 && comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B');
        auto clamp8 = [](int32_t x) { return (uint32_t)(x < 0 ? 0 : x > 255 ? 255 : x); };
        for (size_t i = 0; i < a.size(); ++i) {
            uint32_t r, g, bl;
            if (rgb) { r = a[i]; g = b[i]; bl = c[i]; }
            else {   // jdcolor.c: 16-bit fixed point
                const int32_t y = a[i], cb = (int32_t)b[i] - 128, cr = (int32_t)c[i] - 128;
                r = clamp8(y + ((91881 * cr + 32768) >> 16));
                g = clamp8(y + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
                bl = clamp8(y + ((116130 * cb + 32768) >> 16));
            }
            (*texels)[i] = 0xff000000u | (bl << 16) | (g << 8) | r;
        }
    }
};

// ------------------------------------------------------------------------------------------------ encoder
const uint8_t kLumaQ[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                            18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kChromaQ[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                              99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
const uint8_t kDcLumaBits[17] = {0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChromaBits[17] = {0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumaBits[17] = {0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumaVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1,
    0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a,
    0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3,
    0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcChromaBits[17] = {0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1,
    0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
    0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct EncTable { uint16_t code[256]; uint8_t len[256]; };
EncTable make_enc(const uint8_t* bits, const uint8_t* vals) {
    EncTable t;
    std::memset(&t, 0, sizeof t);
    int code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
        for (int i = 0; i < bits[len]; ++i, ++k, ++code) { t.code[vals[k]] = (uint16_t)code; t.len[vals[k]] = (uint8_t)len; }
        code <<= 1;
    }
    return t;
}

struct BitWriter {
    std::vector<uint8_t>& o;
    uint32_t acc = 0;
    int cnt = 0;
    void put(uint32_t v, int n) {
        acc = (acc << n) | (v & ((1u << n) - 1u));
        cnt += n;
        while (cnt >= 8) {
            const uint8_t b = (uint8_t)(acc >> (cnt - 8));
            o.push_back(b);
            if (b == 0xff) o.push_back(0);
            cnt -= 8;
        }
    }
    void flush() { if (cnt) put(0x7f, 8 - cnt); }
};

// forward DCT, double precision, then quantisation with rounding to nearest (deterministic; not the `image`
// crate's integer FDCT: the output file is "parity unpinned" against the reference)
void fdct_quant(const float* px, const uint16_t* q, int16_t* out) {
    static double c[8][8];
    static bool init = false;
    if (!init) {
        for (int u = 0; u < 8; ++u)
            for (int x = 0; x < 8; ++x) c[u][x] = (u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
        init = true;
    }
    double tmp[64];
    for (int y = 0; y < 8; ++y)
        for (int u = 0; u < 8; ++u) {
            double s = 0;
            for (int x = 0; x < 8; ++x) s += c[u][x] * px[8 * y + x];
            tmp[8 * y + u] = s;
        }
    for (int v = 0; v < 8; ++v)
        for (int u = 0; u < 8; ++u) {
            double s = 0;
            for (int y = 0; y < 8; ++y) s += c[v][y] * tmp[8 * y + u];
            out[8 * v + u] = (int16_t)std::lround(s / q[8 * v + u]);
        }
}

}  // namespace

namespace spt_host {

void decode_jpeg_rgba8(const std::vector<uint8_t>& bytes, const std::string& label, uint32_t* width, uint32_t* height, std::vector<uint32_t>* texels) {
    Decoder dec(bytes, label);
    dec.run(width, height, texels);
}

// RGB8 (row 0 = top) -> baseline JPEG, 4:4:4, quality 1..100 (IJG scaling of the Annex K tables)
void encode_jpeg_rgb8(const uint8_t* rgb, uint32_t w, uint32_t h, int quality, std::vector<uint8_t>* out) {
    quality = quality < 1 ? 1 : quality > 100 ? 100 : quality;
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    uint16_t q[2][64];
    for (int i = 0; i < 64; ++i) {
        for (int t = 0; t < 2; ++t) {
            int v = ((t ? kChromaQ[i] : kLumaQ[i]) * scale + 50) / 100;
            q[t][i] = (uint16_t)(v < 1 ? 1 : v > 255 ? 255 : v);
        }
    }
    std::vector<uint8_t>& o = *out;
    o.clear();
    auto seg = [&](uint8_t m, const std::vector<uint8_t>& body) {
        o.push_back(0xff); o.push_back(m);
        const size_t len = body.size() + 2;
        o.push_back((uint8_t)(len >> 8)); o.push_back((uint8_t)len);
        o.insert(o.end(), body.begin(), body.end());
    };
    o.push_back(0xff); o.push_back(0xd8);
    seg(0xe0, {'J', 'F', 'I', 'F', 0, 1, 2, 0, 0, 1, 0, 1, 0, 0});
    for (int t = 0; t < 2; ++t) {
        std::vector<uint8_t> b{(uint8_t)t};
        for (int i = 0; i < 64; ++i) b.push_back((uint8_t)q[t][kZigzag[i]]);
        seg(0xdb, b);
    }
    seg(0xc0, {8, (uint8_t)(h >> 8), (uint8_t)h, (uint8_t)(w >> 8), (uint8_t)w, 3, 1, 0x11, 0, 2, 0x11, 1, 3, 0x11, 1});
    auto dht = [&](uint8_t id, const uint8_t* bits, const uint8_t* vals, int n) {
        std::vector<uint8_t> b{id};
        for (int i = 1; i <= 16; ++i) b.push_back(bits[i]);
        b.insert(b.end(), vals, vals + n);
        seg(0xc4, b);
    };
    dht(0x00, kDcLumaBits, kDcVals, 12);
    dht(0x10, kAcLumaBits, kAcLumaVals, 162);
    dht(0x01, kDcChromaBits, kDcVals, 12);
    dht(0x11, kAcChromaBits, kAcChromaVals, 162);
    seg(0xda, {3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0});
    const EncTable dct[2] = {make_enc(kDcLumaBits, kDcVals), make_enc(kDcChromaBits, kDcVals)};
    const EncTable act[2] = {make_enc(kAcLumaBits, kAcLumaVals), make_enc(kAcChromaBits, kAcChromaVals)};
    BitWriter bw{o};
    int pred[3] = {0, 0, 0};
    auto put_value = [&](int v, int nbits) { bw.put((uint32_t)(v < 0 ? v + (1 << nbits) - 1 : v), nbits); };
    auto nbits_of = [](int v) { int a = v < 0 ? -v : v, n = 0; while (a) { ++n; a >>= 1; } return n; };
    for (uint32_t by = 0; by < (h + 7) / 8; ++by)
        for (uint32_t bx = 0; bx < (w + 7) / 8; ++bx) {
            float px[3][64];
            for (int y = 0; y < 8; ++y)
                for (int x = 0; x < 8; ++x) {
                    const uint32_t sx = std::min(bx * 8 + (uint32_t)x, w - 1), sy = std::min(by * 8 + (uint32_t)y, h - 1);
                    const uint8_t* s = rgb + ((size_t)sy * w + sx) * 3;
                    const float r = s[0], g = s[1], b = s[2];
                    px[0][8 * y + x] = 0.299f * r + 0.587f * g + 0.114f * b - 128.0f;
                    px[1][8 * y + x] = -0.168735892f * r - 0.331264108f * g + 0.5f * b;
                    px[2][8 * y + x] = 0.5f * r - 0.418687589f * g - 0.081312411f * b;
                }
            for (int c = 0; c < 3; ++c) {
                const int t = c ? 1 : 0;
                int16_t blk[64];
                fdct_quant(px[c], q[t], blk);
                const int diff = blk[0] - pred[c];
                pred[c] = blk[0];
                int nb = nbits_of(diff);
                bw.put(dct[t].code[nb], dct[t].len[nb]);
                if (nb) put_value(diff, nb);
                int run = 0;
                for (int i = 1; i < 64; ++i) {
                    const int v = blk[kZigzag[i]];
                    if (v == 0) { ++run; continue; }
                    while (run > 15) { bw.put(act[t].code[0xf0], act[t].len[0xf0]); run -= 16; }
                    nb = nbits_of(v);
                    bw.put(act[t].code[(run << 4) | nb], act[t].len[(run << 4) | nb]);
                    put_value(v, nb);
                    run = 0;
                }
                if (run) bw.put(act[t].code[0], act[t].len[0]);
            }
        }
    bw.flush();
    o.push_back(0xff); o.push_back(0xd9);
}

}  // namespace spt_host
