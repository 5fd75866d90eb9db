// Image I/O for the host side: PNG write-out of the u8 film (image.save,
// reference src/renderer/pt.rs:290-294) and a minimal OpenEXR scanline
// reader/writer for `environment {type: "exr"}` (get_exr_image, reference
// src/core/loader.rs:374-390 reads the first RGBA layer as f32).
// Supported EXR subset: single-part scanline files, channels R,G,B (A ignored) of
// type HALF or FLOAT, compression NONE / RLE / ZIPS / ZIP / PIZ (exr_piz.cpp) / PXR24, any line order.
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cctype>
#include <cstring>
#include <string>
#include <vector>

#include "host_scene.hpp"

namespace spt_host {
void set_error(const std::string& m);
}
using spt_host::HostError;

namespace {

float half_to_float(uint16_t h) {
    uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31u, m = h & 1023u, u;
    if (e == 0) {
        if (m == 0) {
            u = s;
        } else {
            int sh = 0;
            while (!(m & 1024u)) { m <<= 1; ++sh; }
            m &= 1023u;
            u = s | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
        }
    } else if (e == 31) {
        u = s | 0x7f800000u | (m << 13);
    } else {
        u = s | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

struct Reader {
    const std::vector<uint8_t>& d;
    size_t p = 0;
    explicit Reader(const std::vector<uint8_t>& data) : d(data) {}
    void need(size_t n) { if (p + n > d.size()) throw HostError(SPT_HOST_ERR_PARSE, "exr: truncated file"); }
    uint8_t u8() { need(1); return d[p++]; }
    int32_t i32() { need(4); int32_t v; std::memcpy(&v, &d[p], 4); p += 4; return v; }
    uint64_t u64() { need(8); uint64_t v; std::memcpy(&v, &d[p], 8); p += 8; return v; }
    std::string cstr() {
        std::string s;
        while (true) { uint8_t c = u8(); if (!c) break; s += (char)c; }
        return s;
    }
};

void put_u32(std::vector<uint8_t>& o, uint32_t v) { for (int i = 0; i < 4; ++i) o.push_back((uint8_t)(v >> (8 * i))); }
void put_u64(std::vector<uint8_t>& o, uint64_t v) { for (int i = 0; i < 8; ++i) o.push_back((uint8_t)(v >> (8 * i))); }
void put_str(std::vector<uint8_t>& o, const char* s) { while (*s) o.push_back((uint8_t)*s++); o.push_back(0); }
void put_f32(std::vector<uint8_t>& o, float f) { uint32_t u; std::memcpy(&u, &f, 4); put_u32(o, u); }
void put_attr(std::vector<uint8_t>& o, const char* name, const char* type, const std::vector<uint8_t>& data) {
    put_str(o, name); put_str(o, type); put_u32(o, (uint32_t)data.size());
    o.insert(o.end(), data.begin(), data.end());
}

}  // namespace

namespace spt_host {

// PNG -> RGBA8 as `image::open(path)` + `DynamicImage::get_pixel` present it (src/core/loader.rs:366-371,
// src/texture/image_tex.rs:153-160): gray -> (l,l,l,255), RGB -> a = 255, palette / tRNS expanded,
// 16-bit samples rounded to 8 bits as (c + 128) / 257.  Non-interlaced files only.
void read_png_rgba8(const std::string& path, uint32_t* width, uint32_t* height, std::vector<uint32_t>* texels) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw HostError(SPT_HOST_ERR_IO, "can't read image '" + path + "'");
    std::vector<uint8_t> d;
    uint8_t buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
    std::fclose(f);
    decode_png_rgba8(d, path, width, height, texels);
}

void decode_png_rgba8(const std::vector<uint8_t>& d, const std::string& path, uint32_t* width, uint32_t* height, std::vector<uint32_t>* texels) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (d.size() >= 3 && d[0] == 0xff && d[1] == 0xd8 && d[2] == 0xff) {   // the content decides, as image::open's format guess does
        decode_jpeg_rgba8(d, path, width, height, texels);
        return;
    }
    if (d.size() < 8 || std::memcmp(d.data(), sig, 8) != 0)
        throw HostError(SPT_HOST_ERR_UNSUPPORTED, "image '" + path + "': only PNG and JPEG files are decoded");
    auto be32 = [&](size_t p) { return ((uint32_t)d[p] << 24) | ((uint32_t)d[p + 1] << 16) | ((uint32_t)d[p + 2] << 8) | (uint32_t)d[p + 3]; };
    uint32_t w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    size_t p = 8;
    bool seen_end = false;
    while (p + 12 <= d.size()) {
        uint32_t len = be32(p);
        if (p + 12 + (size_t)len > d.size()) throw HostError(SPT_HOST_ERR_PARSE, "png '" + path + "': truncated chunk");
        std::string type((const char*)&d[p + 4], 4);
        const uint8_t* body = &d[p + 8];
        if (be32(p + 8 + len) != (uint32_t)crc32(0L, &d[p + 4], (uInt)(len + 4))) throw HostError(SPT_HOST_ERR_PARSE, "png '" + path + "': CRC mismatch in " + type);
        if (type == "IHDR") {
            if (len != 13) throw HostError(SPT_HOST_ERR_PARSE, "png '" + path + "': bad IHDR");
            w = be32(p + 8); h = be32(p + 12);
            depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (type == "PLTE") {
            plte.assign(body, body + len);
        } else if (type == "tRNS") {
            trns.assign(body, body + len);
        } else if (type == "IDAT") {
            idat.insert(idat.end(), body, body + len);
        } else if (type == "IEND") {
            seen_end = true;
            break;
        }
        p += 12 + (size_t)len;
    }
    if (!seen_end || w == 0 || h == 0) throw HostError(SPT_HOST_ERR_PARSE, "png '" + path + "': missing IHDR / IEND");
    if (interlace) throw HostError(SPT_HOST_ERR_UNSUPPORTED, "png '" + path + "': interlaced files are not supported");
    uint32_t channels;
    switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: throw HostError(SPT_HOST_ERR_PARSE, "png '" + path + "': bad colour type");
    }
    const bool depth_ok = (ctype == 0) ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                        : (ctype == 3) ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                                       : (depth == 8 || depth == 16);
    if (!depth_ok) throw HostError(SPT_HOST_ERR_PARSE, "png '" + path + "': bad bit depth");
    if ((uint64_t)w * h > 0x3fffffffull) throw HostError(SPT_HOST_ERR_UNSUPPORTED, "png '" + path + "': image too large");
    const size_t bpp_bits = (size_t)channels * depth;
    const size_t stride = ((size_t)w * bpp_bits + 7) / 8;
    const size_t fbpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;   // filter distance in bytes
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf dl = (uLongf)raw.size();
    if (uncompress(raw.data(), &dl, idat.data(), (uLong)idat.size()) != Z_OK || dl != raw.size())
        throw HostError(SPT_HOST_ERR_PARSE, "png '" + path + "': zlib inflate failed");
    // undo the scanline filters in place
    std::vector<uint8_t> zero(stride, 0);
    for (uint32_t y = 0; y < h; ++y) {
        uint8_t* row = &raw[(stride + 1) * (size_t)y + 1];
        const uint8_t* up = y ? &raw[(stride + 1) * (size_t)(y - 1) + 1] : zero.data();
        const uint8_t ft = row[-1];
        for (size_t x = 0; x < stride; ++x) {
            int a = x >= fbpp ? row[x - fbpp] : 0, b = up[x], c = x >= fbpp ? up[x - fbpp] : 0, pred;
            switch (ft) {
            case 0: pred = 0; break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4: {
                int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                break;
            }
            default: throw HostError(SPT_HOST_ERR_PARSE, "png '" + path + "': bad filter type");
            }
            row[x] = (uint8_t)(row[x] + pred);
        }
    }
    auto sample = [&](const uint8_t* row, size_t i) -> uint32_t {   // i-th sample of the row, raw value
        if (depth == 8) return row[i];
        if (depth == 16) return ((uint32_t)row[2 * i] << 8) | row[2 * i + 1];
        size_t bit = i * depth;
        return (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
    };
    auto to8 = [&](uint32_t v) -> uint32_t {
        if (depth == 8) return v;
        if (depth == 16) return (v + 128u) / 257u;
        return v * 255u / ((1u << depth) - 1u);
    };
    texels->assign((size_t)w * h, 0u);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* row = &raw[(stride + 1) * (size_t)y + 1];
        for (uint32_t x = 0; x < w; ++x) {
            uint32_t r, g, b, a = 255u;
            if (ctype == 3) {
                uint32_t i = sample(row, x);
                if ((size_t)i * 3 + 2 >= plte.size()) throw HostError(SPT_HOST_ERR_PARSE, "png '" + path + "': palette index out of range");
                r = plte[3 * i]; g = plte[3 * i + 1]; b = plte[3 * i + 2];
                if (i < trns.size()) a = trns[i];
            } else if (ctype == 0 || ctype == 4) {
                uint32_t v = sample(row, (size_t)x * channels);
                r = g = b = to8(v);
                if (ctype == 4) a = to8(sample(row, (size_t)x * 2 + 1));
                else if (trns.size() >= 2 && v == (((uint32_t)trns[0] << 8) | trns[1])) a = 0u;
            } else {
                uint32_t v0 = sample(row, (size_t)x * channels), v1 = sample(row, (size_t)x * channels + 1), v2 = sample(row, (size_t)x * channels + 2);
                r = to8(v0); g = to8(v1); b = to8(v2);
                if (ctype == 6) a = to8(sample(row, (size_t)x * 4 + 3));
                else if (trns.size() >= 6 && v0 == (((uint32_t)trns[0] << 8) | trns[1]) && v1 == (((uint32_t)trns[2] << 8) | trns[3]) &&
                         v2 == (((uint32_t)trns[4] << 8) | trns[5])) a = 0u;
            }
            (*texels)[(size_t)y * w + x] = r | (g << 8) | (b << 16) | (a << 24);
        }
    }
    *width = w;
    *height = h;
}

}  // namespace spt_host

extern "C" {

void spt_host_free(void* p) { std::free(p); }

spt_status spt_host_read_exr(const char* path, uint32_t* width, uint32_t* height, float** rgb_out) {
    try {
        if (!path || !width || !height || !rgb_out) throw HostError(SPT_ERR_INVALID_ARG, "read_exr: null argument");
        FILE* f = std::fopen(path, "rb");
        if (!f) throw HostError(SPT_HOST_ERR_IO, std::string("cannot open EXR '") + path + "'");
        std::vector<uint8_t> data;
        uint8_t buf[65536];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + n);
        std::fclose(f);
        Reader r(data);
        if (r.i32() != 20000630) throw HostError(SPT_HOST_ERR_PARSE, "exr: bad magic");
        int32_t version = r.i32();
        if (version & 0x1A00) throw HostError(SPT_HOST_ERR_UNSUPPORTED, "exr: tiled / deep / multipart files are not supported");
        struct Chan { std::string name; int32_t type; };
        std::vector<Chan> chans;
        int compression = -1;
        int32_t dw[4] = {0, 0, -1, -1};
        while (true) {
            std::string name = r.cstr();
            if (name.empty()) break;
            std::string type = r.cstr();
            int32_t size = r.i32();
            size_t end = r.p + (size_t)size;
            r.need((size_t)size);
            if (name == "channels") {
                while (true) {
                    std::string cn = r.cstr();
                    if (cn.empty()) break;
                    Chan c; c.name = cn; c.type = r.i32();
                    r.p += 4;  // pLinear + reserved
                    int32_t xs = r.i32(), ys = r.i32();
                    if (xs != 1 || ys != 1) throw HostError(SPT_HOST_ERR_UNSUPPORTED, "exr: subsampled channels are not supported");
                    chans.push_back(c);
                }
            } else if (name == "compression") {
                compression = r.u8();
            } else if (name == "dataWindow") {
                for (int i = 0; i < 4; ++i) dw[i] = r.i32();
            }
            r.p = end;
        }
        if (compression < 0 || compression > 5)
            throw HostError(SPT_HOST_ERR_UNSUPPORTED, "exr: only NONE / RLE / ZIPS / ZIP / PIZ / PXR24 compression is supported (not B44, DWA)");
        int64_t w = (int64_t)dw[2] - dw[0] + 1, h = (int64_t)dw[3] - dw[1] + 1;
        if (w <= 0 || h <= 0 || w > 65536 || h > 65536) throw HostError(SPT_HOST_ERR_PARSE, "exr: bad dataWindow");
        int lines_per_block = (compression == 3 || compression == 5) ? 16 : (compression == 4 ? 32 : 1);
        int64_t n_blocks = (h + lines_per_block - 1) / lines_per_block;
        size_t row_bytes = 0;
        std::vector<size_t> chan_off(chans.size());
        int idx[3] = {-1, -1, -1};
        for (size_t c = 0; c < chans.size(); ++c) {
            chan_off[c] = row_bytes;
            if (chans[c].type != 1 && chans[c].type != 2 && chans[c].type != 0) throw HostError(SPT_HOST_ERR_PARSE, "exr: bad channel type");
            row_bytes += (size_t)w * (chans[c].type == 1 ? 2 : 4);
            if (chans[c].name == "R") idx[0] = (int)c;
            if (chans[c].name == "G") idx[1] = (int)c;
            if (chans[c].name == "B") idx[2] = (int)c;
        }
        if (idx[0] < 0 || idx[1] < 0 || idx[2] < 0) throw HostError(SPT_HOST_ERR_UNSUPPORTED, "exr: needs R, G and B channels");
        for (int k = 0; k < 3; ++k)
            if (chans[idx[k]].type == 0) throw HostError(SPT_HOST_ERR_UNSUPPORTED, "exr: UINT colour channels are not supported");
        std::vector<uint64_t> offsets((size_t)n_blocks);
        for (auto& o : offsets) o = r.u64();
        float* out = (float*)std::malloc((size_t)w * h * 3 * sizeof(float));
        if (!out) throw HostError(SPT_ERR_OUT_OF_MEMORY, "exr: out of memory");
        try {
            std::vector<uint8_t> raw, tmp;
            for (int64_t b = 0; b < n_blocks; ++b) {
                r.p = (size_t)offsets[(size_t)b];
                int32_t y0 = r.i32() - dw[1];
                int32_t size = r.i32();
                if (size < 0 || y0 < 0 || y0 >= h) throw HostError(SPT_HOST_ERR_PARSE, "exr: bad chunk");
                r.need((size_t)size);
                int64_t lines = std::min<int64_t>(lines_per_block, h - y0);
                size_t want = row_bytes * (size_t)lines;
                raw.resize(want);
                if (compression == 0 || (size_t)size == want) {
                    if ((size_t)size != want) throw HostError(SPT_HOST_ERR_PARSE, "exr: chunk size mismatch");
                    std::memcpy(raw.data(), &data[r.p], want);
                } else if (compression == 4) {
                    std::vector<int> wpp;
                    for (const Chan& c : chans) wpp.push_back(c.type == 1 ? 1 : 2);
                    spt_host::exr_piz_decode(&data[r.p], (size_t)size, wpp, w, lines, raw.data(), want);
                } else if (compression == 5) {
                    // PXR24 (ImfPxr24Compressor): zlib over byte planes of per-row, per-channel running differences;
                    // FLOAT samples keep their top 24 bits
                    size_t packed = 0;
                    for (const Chan& c : chans) packed += (size_t)w * (c.type == 1 ? 2 : c.type == 2 ? 3 : 4);
                    packed *= (size_t)lines;
                    tmp.resize(packed);
                    uLongf dl = (uLongf)packed;
                    if (uncompress(tmp.data(), &dl, &data[r.p], (uLong)size) != Z_OK || dl != packed)
                        throw HostError(SPT_HOST_ERR_PARSE, "exr: zlib inflate failed (PXR24)");
                    const uint8_t* in = tmp.data();
                    uint8_t* dst = raw.data();
                    for (int64_t l = 0; l < lines; ++l)
                        for (const Chan& c : chans) {
                            const size_t n = (size_t)w;
                            if (c.type == 1) {
                                uint16_t px = 0;
                                for (size_t x = 0; x < n; ++x) {
                                    px = (uint16_t)(px + (((uint32_t)in[x] << 8) | in[n + x]));
                                    std::memcpy(dst + 2 * x, &px, 2);
                                }
                                in += 2 * n; dst += 2 * n;
                            } else {
                                uint32_t px = 0;
                                for (size_t x = 0; x < n; ++x) {
                                    uint32_t diff = ((uint32_t)in[x] << 24) | ((uint32_t)in[n + x] << 16) | ((uint32_t)in[2 * n + x] << 8);
                                    if (c.type == 0) diff |= in[3 * n + x];
                                    px += diff;
                                    std::memcpy(dst + 4 * x, &px, 4);
                                }
                                in += (c.type == 0 ? 4 : 3) * n; dst += 4 * n;
                            }
                        }
                } else {
                    tmp.resize(want);
                    if (compression == 1) {
                        // RLE (ImfRle): a signed count byte, negative = that many literal bytes, else count + 1 copies
                        size_t ip = r.p, end = r.p + (size_t)size, op = 0;
                        while (ip < end) {
                            const int8_t c = (int8_t)data[ip++];
                            if (c < 0) {
                                const size_t cnt = (size_t)(-(int)c);
                                if (ip + cnt > end || op + cnt > want) throw HostError(SPT_HOST_ERR_PARSE, "exr: bad RLE data");
                                std::memcpy(&tmp[op], &data[ip], cnt);
                                ip += cnt; op += cnt;
                            } else {
                                const size_t cnt = (size_t)c + 1;
                                if (ip >= end || op + cnt > want) throw HostError(SPT_HOST_ERR_PARSE, "exr: bad RLE data");
                                std::memset(&tmp[op], data[ip++], cnt);
                                op += cnt;
                            }
                        }
                        if (op != want) throw HostError(SPT_HOST_ERR_PARSE, "exr: RLE data of the wrong length");
                    } else {
                        uLongf dl = (uLongf)want;
                        if (uncompress(tmp.data(), &dl, &data[r.p], (uLong)size) != Z_OK || dl != want)
                            throw HostError(SPT_HOST_ERR_PARSE, "exr: zlib inflate failed");
                    }
                    for (size_t i = 1; i < want; ++i) tmp[i] = (uint8_t)(tmp[i - 1] + tmp[i] - 128);
                    size_t half = (want + 1) / 2;
                    for (size_t i = 0; i < want; ++i) raw[i] = (i & 1) ? tmp[half + i / 2] : tmp[i / 2];
                }
                for (int64_t l = 0; l < lines; ++l) {
                    const uint8_t* row = raw.data() + row_bytes * (size_t)l;
                    for (int k = 0; k < 3; ++k) {
                        const Chan& c = chans[idx[k]];
                        const uint8_t* src = row + chan_off[idx[k]];
                        for (int64_t x = 0; x < w; ++x) {
                            float v;
                            if (c.type == 1) { uint16_t hv; std::memcpy(&hv, src + 2 * x, 2); v = half_to_float(hv); }
                            else std::memcpy(&v, src + 4 * x, 4);
                            out[((size_t)(y0 + l) * w + x) * 3 + k] = v;
                        }
                    }
                }
            }
        } catch (...) {
            std::free(out);
            throw;
        }
        *width = (uint32_t)w; *height = (uint32_t)h; *rgb_out = out;
        return SPT_OK;
    } catch (const HostError& e) {
        spt_host::set_error(e.msg);
        return e.code;
    }
}

spt_status spt_host_write_exr(const char* path, const float* rgb, uint32_t width, uint32_t height) {
    if (!path || !rgb || !width || !height) { spt_host::set_error("write_exr: bad argument"); return SPT_ERR_INVALID_ARG; }
    std::vector<uint8_t> o;
    put_u32(o, 20000630u);
    put_u32(o, 2u);
    {
        std::vector<uint8_t> ch;
        for (const char* n : {"B", "G", "R"}) {
            put_str(ch, n); put_u32(ch, 2u); put_u32(ch, 0u); put_u32(ch, 1u); put_u32(ch, 1u);
        }
        ch.push_back(0);
        put_attr(o, "channels", "chlist", ch);
    }
    put_attr(o, "compression", "compression", {0});
    std::vector<uint8_t> box;
    put_u32(box, 0); put_u32(box, 0); put_u32(box, width - 1); put_u32(box, height - 1);
    put_attr(o, "dataWindow", "box2i", box);
    put_attr(o, "displayWindow", "box2i", box);
    put_attr(o, "lineOrder", "lineOrder", {0});
    { std::vector<uint8_t> v; put_f32(v, 1.0f); put_attr(o, "pixelAspectRatio", "float", v); }
    { std::vector<uint8_t> v; put_f32(v, 0.0f); put_f32(v, 0.0f); put_attr(o, "screenWindowCenter", "v2f", v); }
    { std::vector<uint8_t> v; put_f32(v, 1.0f); put_attr(o, "screenWindowWidth", "float", v); }
    o.push_back(0);
    size_t row = (size_t)width * 12;
    uint64_t first = o.size() + (uint64_t)height * 8;
    for (uint32_t y = 0; y < height; ++y) put_u64(o, first + (uint64_t)y * (8 + row));
    for (uint32_t y = 0; y < height; ++y) {
        put_u32(o, y);
        put_u32(o, (uint32_t)row);
        for (int c = 2; c >= 0; --c)  // B, G, R planes
            for (uint32_t x = 0; x < width; ++x) put_f32(o, rgb[((size_t)y * width + x) * 3 + c]);
    }
    FILE* f = std::fopen(path, "wb");
    if (!f) { spt_host::set_error(std::string("cannot write '") + path + "'"); return SPT_HOST_ERR_IO; }
    size_t wr = std::fwrite(o.data(), 1, o.size(), f);
    std::fclose(f);
    if (wr != o.size()) { spt_host::set_error("short write"); return SPT_HOST_ERR_IO; }
    return SPT_OK;
}

void spt_host_film_to_rgb8(const float* rgb_mean, uint64_t n_pixels, uint8_t* rgb8_out) {
    for (uint64_t i = 0; i < n_pixels * 3; ++i) {
        float c = rgb_mean[i] * 255.0f;
        // Rust clamp keeps NaN, and `NaN as u8` is 0
        float cl = c < 0.0f ? 0.0f : (c > 255.0f ? 255.0f : c);
        rgb8_out[i] = (cl != cl) ? 0 : (uint8_t)cl;
    }
}

spt_status spt_host_write_png(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height) {
    if (!path || !rgb8 || !width || !height) { spt_host::set_error("write_png: bad argument"); return SPT_ERR_INVALID_ARG; }
    std::vector<uint8_t> raw;
    raw.reserve((size_t)height * (1 + (size_t)width * 3));
    for (uint32_t y = 0; y < height; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb8 + (size_t)y * width * 3, rgb8 + (size_t)(y + 1) * width * 3);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { spt_host::set_error("png: deflate failed"); return SPT_HOST_ERR_IO; }
    comp.resize(clen);
    std::vector<uint8_t> o = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    auto be32 = [&](std::vector<uint8_t>& v, uint32_t x) { for (int i = 3; i >= 0; --i) v.push_back((uint8_t)(x >> (8 * i))); };
    auto chunk = [&](const char* type, const std::vector<uint8_t>& d) {
        be32(o, (uint32_t)d.size());
        std::vector<uint8_t> td(type, type + 4);
        td.insert(td.end(), d.begin(), d.end());
        o.insert(o.end(), td.begin(), td.end());
        be32(o, (uint32_t)crc32(0L, td.data(), (uInt)td.size()));
    };
    std::vector<uint8_t> ihdr;
    be32(ihdr, width); be32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk("IHDR", ihdr);
    chunk("IDAT", comp);
    chunk("IEND", {});
    FILE* f = std::fopen(path, "wb");
    if (!f) { spt_host::set_error(std::string("cannot write '") + path + "'"); return SPT_HOST_ERR_IO; }
    size_t wr = std::fwrite(o.data(), 1, o.size(), f);
    std::fclose(f);
    if (wr != o.size()) { spt_host::set_error("short write"); return SPT_HOST_ERR_IO; }
    return SPT_OK;
}

// image.save (src/renderer/pt.rs:292-294): the extension picks the format
spt_status spt_host_write_jpeg(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height, int32_t quality) {
    if (!path || !rgb8 || !width || !height || width > 65535 || height > 65535) { spt_host::set_error("write_jpeg: bad argument"); return SPT_ERR_INVALID_ARG; }
    std::vector<uint8_t> o;
    spt_host::encode_jpeg_rgb8(rgb8, width, height, quality, &o);
    FILE* f = std::fopen(path, "wb");
    if (!f) { spt_host::set_error(std::string("cannot write '") + path + "'"); return SPT_HOST_ERR_IO; }
    size_t wr = std::fwrite(o.data(), 1, o.size(), f);
    std::fclose(f);
    if (wr != o.size()) { spt_host::set_error("short write"); return SPT_HOST_ERR_IO; }
    return SPT_OK;
}

spt_status spt_host_write_image(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height) {
    if (!path) { spt_host::set_error("write_image: bad argument"); return SPT_ERR_INVALID_ARG; }
    std::string p(path), ext;
    const size_t dot = p.find_last_of('.');
    if (dot != std::string::npos && p.find('/', dot) == std::string::npos)
        for (size_t i = dot + 1; i < p.size(); ++i) ext.push_back((char)std::tolower((unsigned char)p[i]));
    if (ext == "png") return spt_host_write_png(path, rgb8, width, height);
    if (ext == "jpg" || ext == "jpeg") return spt_host_write_jpeg(path, rgb8, width, height, 75);   // the image crate's JpegEncoder default
    spt_host::set_error("Failed to save image, err: the image format could not be determined from the extension of '" + p + "' (png, jpg and jpeg are written)");
    return SPT_HOST_ERR_UNSUPPORTED;
}

spt_status spt_host_read_image(const char* path, uint32_t* width, uint32_t* height, uint32_t** rgba8_out) {
    return spt_host_read_png(path, width, height, rgba8_out);
}

spt_status spt_host_read_png(const char* path, uint32_t* width, uint32_t* height, uint32_t** rgba8_out) {
    try {
        if (!path || !width || !height || !rgba8_out) throw HostError(SPT_ERR_INVALID_ARG, "read_png: null argument");
        std::vector<uint32_t> t;
        spt_host::read_png_rgba8(path, width, height, &t);
        uint32_t* out = (uint32_t*)std::malloc(t.size() * 4);
        if (!out) throw HostError(SPT_HOST_ERR_IO, "read_png: out of memory");
        std::memcpy(out, t.data(), t.size() * 4);
        *rgba8_out = out;
        return SPT_OK;
    } catch (const HostError& e) {
        spt_host::set_error(e.msg);
        return e.code;
    }
}

}  // extern "C"
