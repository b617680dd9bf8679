// ImageLoader.cpp — the harness counterpart of `Image.Load<Rgb24>` in Textures/Image.fs:21-24: decode a local image
// file to tightly packed RGB bytes, row 0 = top.  The reference delegates to SixLabors.ImageSharp (any format, local
// file or HTTP); this loader reads what can be decoded without it: PNG (8 / 16 bit, grey, grey+alpha, RGB, RGBA,
// palette; non-interlaced; inflate by the system zlib) and binary / ASCII PPM / PGM.  Alpha is dropped and 16-bit
// samples keep their high byte, which is what a conversion to Rgb24 does; baseline JPEG (below).  URLs are refused with a message.
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "Scene.hpp"

namespace FuncTracer {
namespace {

uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

bool decodePng(const std::vector<uint8_t>& f, int& w, int& h, std::vector<uint8_t>& rgb, std::string& err) {
    size_t at = 8;
    std::vector<uint8_t> idat, palette;
    int depth = 0, colour = 0, interlace = 0;
    bool header = false, end = false;
    while (!end && at + 12 <= f.size()) {
        const uint32_t len = be32(&f[at]);
        if (at + 12 + (size_t)len > f.size()) { err = "PNG: truncated chunk"; return false; }
        const uint8_t* type = &f[at + 4];
        const uint8_t* body = &f[at + 8];
        if (crc32(0L, type, len + 4) != be32(body + len)) { err = "PNG: chunk CRC mismatch"; return false; }
        if (!std::memcmp(type, "IHDR", 4) && len == 13) {
            w = (int)be32(body); h = (int)be32(body + 4); depth = body[8]; colour = body[9]; interlace = body[12]; header = true;
        } else if (!std::memcmp(type, "PLTE", 4)) palette.assign(body, body + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(type, "IEND", 4)) end = true;
        at += 12 + (size_t)len;
    }
    if (!header || w <= 0 || h <= 0 || (int64_t)w * h > (1ll << 28)) { err = "PNG: bad header"; return false; }
    if (interlace) { err = "PNG: interlaced images are not supported"; return false; }
    int channels;
    switch (colour) { case 0: channels = 1; break; case 2: channels = 3; break; case 3: channels = 1; break; case 4: channels = 2; break; case 6: channels = 4; break;
                      default: err = "PNG: bad colour type"; return false; }
    const bool depth_ok = colour == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                        : colour == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16) : (depth == 8 || depth == 16);
    if (!depth_ok) { err = "PNG: bad bit depth"; return false; }
    const size_t bits_per_pixel = (size_t)channels * depth, stride = ((size_t)w * bits_per_pixel + 7) / 8, bpp = (bits_per_pixel + 7) / 8;
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf got = (uLongf)raw.size();
    if (uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw.size()) { err = "PNG: inflate failed"; return false; }
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    rgb.assign((size_t)w * h * 3, 0);
    for (int y = 0; y < h; ++y) {
        const uint8_t* line = &raw[(stride + 1) * (size_t)y];
        const int filter = line[0];
        for (size_t i = 0; i < stride; ++i) {                        // PNG filters (spec section 9): None, Sub, Up, Average, Paeth
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int pred = 0;
            switch (filter) {
                case 0: break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                          pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: err = "PNG: bad filter"; return false;
            }
            cur[i] = (uint8_t)(line[1 + i] + pred);
        }
        for (int x = 0; x < w; ++x) {
            uint8_t* out = &rgb[((size_t)y * w + x) * 3];
            auto sample = [&](int ch) -> int {                       // sample `ch` of pixel x; 16-bit samples keep the high byte
                if (depth >= 8) return cur[((size_t)x * channels + ch) * (depth / 8)];
                const size_t bit = (size_t)x * depth;
                return (cur[bit / 8] >> (8 - depth - (bit % 8))) & ((1 << depth) - 1);
            };
            if (colour == 3) {
                const size_t idx = (size_t)sample(0) * 3;
                if (idx + 3 > palette.size()) { err = "PNG: palette index out of range"; return false; }
                out[0] = palette[idx]; out[1] = palette[idx + 1]; out[2] = palette[idx + 2];
            } else if (channels <= 2) {
                int g = sample(0);
                if (depth < 8) g = g * 255 / ((1 << depth) - 1);
                out[0] = out[1] = out[2] = (uint8_t)g;
            } else { out[0] = (uint8_t)sample(0); out[1] = (uint8_t)sample(1); out[2] = (uint8_t)sample(2); }
        }
        prev.swap(cur);
    }
    return true;
}

bool decodePnm(const std::vector<uint8_t>& f, int& w, int& h, std::vector<uint8_t>& rgb, std::string& err) {
    size_t at = 2;
    auto token = [&]() -> long {                                     // whitespace / '#' comments, then a decimal number
        for (;;) {
            while (at < f.size() && std::isspace(f[at])) ++at;
            if (at < f.size() && f[at] == '#') { while (at < f.size() && f[at] != '\n') ++at; continue; }
            break;
        }
        long v = -1;
        while (at < f.size() && std::isdigit(f[at])) { v = (v < 0 ? 0 : v) * 10 + (f[at] - '0'); ++at; }
        return v;
    };
    const char kind = (char)f[1];
    const int channels = (kind == '3' || kind == '6') ? 3 : 1;
    const bool binary = kind == '5' || kind == '6';
    w = (int)token(); h = (int)token();
    const long maxval = token();
    if (w <= 0 || h <= 0 || maxval <= 0 || maxval > 65535 || (int64_t)w * h > (1ll << 28)) { err = "PNM: bad header"; return false; }
    const size_t n = (size_t)w * h * channels;
    std::vector<long> v(n);
    if (binary) {
        ++at;                                                        // the single whitespace byte after maxval
        const size_t bytes = maxval > 255 ? 2 : 1;
        if (at + n * bytes > f.size()) { err = "PNM: truncated raster"; return false; }
        for (size_t i = 0; i < n; ++i) v[i] = bytes == 2 ? (f[at + 2 * i] << 8 | f[at + 2 * i + 1]) : f[at + i];
    } else {
        for (size_t i = 0; i < n; ++i) { v[i] = token(); if (v[i] < 0) { err = "PNM: truncated raster"; return false; } }
    }
    rgb.resize((size_t)w * h * 3);
    for (size_t p = 0; p < (size_t)w * h; ++p)
        for (int c = 0; c < 3; ++c) { const long s = v[p * channels + (channels == 3 ? c : 0)]; rgb[3 * p + c] = (uint8_t)(maxval == 255 ? s : s * 255 / maxval); }
    return true;
}

// ---- JPEG: baseline / extended sequential Huffman, 8 bit, one (grey) or three (YCbCr) components ---------------------------------------
// What `Image.Load<Rgb24>` of Scenes/sample.scene:6 ("c:\Temp\env4.jpg") needs.  The arithmetic is the Independent JPEG Group's reference
// one, restated from its published description: the "slow integer" inverse DCT (Loeffler-Ligtenberg-Moschytz, 13-bit constants), triangle
// ("fancy") chroma upsampling for 2:1 horizontal and 2:1 x 2:1 subsampling, the 16-bit fixed-point YCbCr -> RGB conversion - the output
// equals libjpeg's byte for byte (tests/test_host_logic.py compares with Pillow).  ImageSharp, which the reference calls, is its own
// decoder and may differ from either by a level in places; progressive and arithmetic-coded files are refused.
struct JpegDecoder {
    const std::vector<uint8_t>& f;
    std::string& err;
    size_t at = 2;
    uint16_t quant[4][64] = {};
    struct Huff { uint8_t bits[17] = {}; uint8_t vals[256] = {}; int mincode[17] = {}, maxcode[18] = {}, valptr[17] = {}; bool set = false; };
    Huff dc[4], ac[4];
    struct Comp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0, bw = 0, bh = 0; std::vector<uint8_t> plane; int pw = 0, ph = 0; };
    std::vector<Comp> comps;
    int width = 0, height = 0, restart = 0;
    uint32_t bitbuf = 0; int bitcnt = 0; bool hit_marker = false;
    JpegDecoder(const std::vector<uint8_t>& file, std::string& e) : f(file), err(e) {}
    static constexpr int kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                        35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    bool fail(const char* m) { err = std::string("JPEG: ") + m; return false; }
    int u16(size_t p) const { return f[p] << 8 | f[p + 1]; }
    void buildHuff(Huff& h) {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            h.valptr[l] = k; h.mincode[l] = code;
            code += h.bits[l]; k += h.bits[l];
            h.maxcode[l] = h.bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        h.maxcode[17] = 0x7FFFFFFF; h.set = true;
    }
    int nextBit() {
        if (bitcnt == 0) {
            int b = 0;
            if (!hit_marker && at < f.size()) {
                b = f[at++];
                if (b == 0xFF) {
                    const int b2 = at < f.size() ? f[at] : 0;
                    if (b2 == 0) ++at;                               // a stuffed zero byte
                    else { hit_marker = true; --at; b = 0; }         // a marker: the entropy-coded segment ends, zeros from here on
                }
            }
            bitbuf = (uint32_t)b; bitcnt = 8;
        }
        --bitcnt;
        return (bitbuf >> bitcnt) & 1;
    }
    int receive(int n) { int v = 0; while (n--) v = v << 1 | nextBit(); return v; }
    static int extend(int v, int n) { return n && v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }
    int decodeSymbol(const Huff& h) {
        int code = 0;
        for (int l = 1; l <= 16; ++l) {
            code = code << 1 | nextBit();
            if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
        }
        return -1;
    }
    static inline int descale(long x, int n) { return (int)((x + (1L << (n - 1))) >> n); }
    static inline uint8_t clamp255(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
    // jidctint: two passes of a scaled 1-D LL&M transform; coefficients arrive dequantised.
    static void idct(const int in[64], uint8_t* out, int stride) {
        constexpr int CB = 13, P1 = 2;
        constexpr long F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299, F1_847 = 15137, F1_961 = 16069,
                       F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
        long ws[64];
        for (int c = 0; c < 8; ++c) {
            const int* p = in + c;
            if (!p[8] && !p[16] && !p[24] && !p[32] && !p[40] && !p[48] && !p[56]) { const long d = (long)p[0] << P1; for (int r = 0; r < 8; ++r) ws[8 * r + c] = d; continue; }
            long z2 = p[16], z3 = p[48];
            long z1 = (z2 + z3) * F0_541;
            long tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
            z2 = p[0]; z3 = p[32];
            long tmp0 = (z2 + z3) << CB, tmp1 = (z2 - z3) << CB;
            const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = p[56]; tmp1 = p[40]; tmp2 = p[24]; tmp3 = p[8];
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3;
            const long z5 = (z3 + z4) * F1_175;
            tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
            z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            ws[c] = descale(tmp10 + tmp3, CB - P1); ws[56 + c] = descale(tmp10 - tmp3, CB - P1);
            ws[8 + c] = descale(tmp11 + tmp2, CB - P1); ws[48 + c] = descale(tmp11 - tmp2, CB - P1);
            ws[16 + c] = descale(tmp12 + tmp1, CB - P1); ws[40 + c] = descale(tmp12 - tmp1, CB - P1);
            ws[24 + c] = descale(tmp13 + tmp0, CB - P1); ws[32 + c] = descale(tmp13 - tmp0, CB - P1);
        }
        for (int r = 0; r < 8; ++r) {
            const long* p = ws + 8 * r;
            uint8_t* o = out + (size_t)r * stride;
            // (the reference's range-limit table wraps the index to 10 bits before clamping: the same for every value a real image produces)
            if (!p[1] && !p[2] && !p[3] && !p[4] && !p[5] && !p[6] && !p[7]) { const uint8_t d = clamp255(descale(p[0], P1 + 3) + 128); for (int c = 0; c < 8; ++c) o[c] = d; continue; }
            long z2 = p[2], z3 = p[6];
            long z1 = (z2 + z3) * F0_541;
            long tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
            long tmp0 = (p[0] + p[4]) << CB, tmp1 = (p[0] - p[4]) << CB;
            const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = p[7]; tmp1 = p[5]; tmp2 = p[3]; tmp3 = p[1];
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3;
            const long z5 = (z3 + z4) * F1_175;
            tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
            z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            constexpr int S = CB + P1 + 3;
            o[0] = clamp255(descale(tmp10 + tmp3, S) + 128); o[7] = clamp255(descale(tmp10 - tmp3, S) + 128);
            o[1] = clamp255(descale(tmp11 + tmp2, S) + 128); o[6] = clamp255(descale(tmp11 - tmp2, S) + 128);
            o[2] = clamp255(descale(tmp12 + tmp1, S) + 128); o[5] = clamp255(descale(tmp12 - tmp1, S) + 128);
            o[3] = clamp255(descale(tmp13 + tmp0, S) + 128); o[4] = clamp255(descale(tmp13 - tmp0, S) + 128);
        }
    }
    bool decodeBlock(Comp& c, uint8_t* out, int stride) {
        int coef[64] = {};
        int t = decodeSymbol(dc[c.td]);
        if (t < 0 || t > 15) return fail("bad DC code");
        c.pred += extend(receive(t), t);
        const uint16_t* q = quant[c.tq];
        coef[0] = c.pred * q[0];
        for (int k = 1; k < 64;) {
            const int rs = decodeSymbol(ac[c.ta]);
            if (rs < 0) return fail("bad AC code");
            const int r = rs >> 4, sz = rs & 15;
            if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
            k += r;
            if (k > 63) return fail("AC run past the block");
            coef[kZigzag[k]] = extend(receive(sz), sz) * q[k];
            ++k;
        }
        idct(coef, out, stride);
        return true;
    }
    bool parse(int& w, int& h, std::vector<uint8_t>& rgb) {
        bool sof = false;
        for (;;) {
            while (at < f.size() && f[at] != 0xFF) ++at;
            while (at < f.size() && f[at] == 0xFF) ++at;
            if (at >= f.size()) return fail("no scan found");
            const int m = f[at++];
            if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
            if (m == 0xD9) return fail("end of image before a scan");
            if (at + 2 > f.size()) return fail("truncated marker");
            const int len = u16(at);
            if (len < 2 || at + (size_t)len > f.size()) return fail("truncated segment");
            const size_t seg = at + 2, end = at + (size_t)len;
            if (m == 0xDB) {                                         // DQT
                for (size_t p = seg; p < end;) {
                    const int pq = f[p] >> 4, tq = f[p] & 15; ++p;
                    if (tq > 3 || p + (size_t)(pq ? 128 : 64) > end) return fail("bad quantisation table");
                    for (int k = 0; k < 64; ++k) { quant[tq][k] = (uint16_t)(pq ? u16(p + 2 * k) : f[p + k]); }
                    p += pq ? 128 : 64;
                }
            } else if (m == 0xC4) {                                  // DHT
                for (size_t p = seg; p < end;) {
                    const int tc = f[p] >> 4, th = f[p] & 15; ++p;
                    if (tc > 1 || th > 3 || p + 16 > end) return fail("bad Huffman table");
                    Huff& hf = tc ? ac[th] : dc[th];
                    int total = 0;
                    for (int l = 1; l <= 16; ++l) { hf.bits[l] = f[p + l - 1]; total += hf.bits[l]; }
                    p += 16;
                    if (total > 256 || p + (size_t)total > end) return fail("bad Huffman table");
                    for (int k = 0; k < total; ++k) hf.vals[k] = f[p + k];
                    p += total;
                    buildHuff(hf);
                }
            } else if (m == 0xC0 || m == 0xC1) {                     // SOF0 / SOF1: baseline, extended sequential (Huffman)
                if (len < 8 || f[seg] != 8) return fail("only 8-bit samples are read");
                height = u16(seg + 1); width = u16(seg + 3);
                const int n = f[seg + 5];
                if (width <= 0 || height <= 0 || (int64_t)width * height > (1ll << 28) || (n != 1 && n != 3) || seg + 6 + 3 * (size_t)n > end) return fail("bad frame header");
                comps.resize((size_t)n);
                for (int k = 0; k < n; ++k) { Comp& c = comps[(size_t)k]; c.id = f[seg + 6 + 3 * k]; c.h = f[seg + 7 + 3 * k] >> 4; c.v = f[seg + 7 + 3 * k] & 15; c.tq = f[seg + 8 + 3 * k] & 3; }
                sof = true;
            } else if (m == 0xC2 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) return fail("progressive, lossless and arithmetic-coded files are not read (baseline only)");
            else if (m == 0xDD) { if (len >= 4) restart = u16(seg); }
            else if (m == 0xDA) {                                    // SOS
                if (!sof) return fail("scan before the frame header");
                const int ns = f[seg];
                if (ns != (int)comps.size() || seg + 1 + 2 * (size_t)ns + 3 > end) return fail("only single-scan (non-progressive) files are read");
                for (int k = 0; k < ns; ++k) {
                    const int id = f[seg + 1 + 2 * k];
                    Comp* c = nullptr;
                    for (auto& q : comps) if (q.id == id) c = &q;
                    if (!c) return fail("scan names an unknown component");
                    c->td = f[seg + 2 + 2 * k] >> 4; c->ta = f[seg + 2 + 2 * k] & 15;
                    if (c->td > 3 || c->ta > 3 || !dc[c->td].set || !ac[c->ta].set) return fail("scan uses a missing Huffman table");
                }
                at = end;
                return decodeScan(w, h, rgb);
            }
            at = end;
        }
    }
    bool decodeScan(int& w, int& h, std::vector<uint8_t>& rgb) {
        int hmax = 1, vmax = 1;
        for (auto& c : comps) { if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2) return fail("sampling factors other than 1 and 2 are not read"); hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v); }
        if (comps.size() == 3 && (comps[1].h != 1 || comps[1].v != 1 || comps[2].h != 1 || comps[2].v != 1)) return fail("subsampled luma / oversampled chroma is not read");
        if (comps.size() == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; }             // a single component is never interleaved: 8x8 MCUs
        const int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
        for (auto& c : comps) { c.pw = mcux * c.h * 8; c.ph = mcuy * c.v * 8; c.plane.assign((size_t)c.pw * c.ph, 0); c.pred = 0; }
        int togo = restart;
        for (int my = 0; my < mcuy; ++my)
            for (int mx = 0; mx < mcux; ++mx) {
                if (restart && togo == 0) {                          // RSTn: byte align, skip the marker, reset the predictors
                    bitcnt = 0; hit_marker = false;
                    while (at + 1 < f.size() && !(f[at] == 0xFF && f[at + 1] >= 0xD0 && f[at + 1] <= 0xD7)) ++at;
                    if (at + 1 >= f.size()) return fail("missing restart marker");
                    at += 2;
                    for (auto& c : comps) c.pred = 0;
                    togo = restart;
                }
                for (auto& c : comps)
                    for (int by = 0; by < c.v; ++by)
                        for (int bx = 0; bx < c.h; ++bx)
                            if (!decodeBlock(c, &c.plane[(size_t)(my * c.v + by) * 8 * c.pw + (size_t)(mx * c.h + bx) * 8], c.pw)) return false;
                if (restart) --togo;
            }
        w = width; h = height;
        rgb.assign((size_t)width * height * 3, 0);
        if (comps.size() == 1) {
            for (int y = 0; y < height; ++y) for (int x = 0; x < width; ++x) { const uint8_t g = comps[0].plane[(size_t)y * comps[0].pw + x]; uint8_t* o = &rgb[((size_t)y * width + x) * 3]; o[0] = o[1] = o[2] = g; }
            return true;
        }
        // chroma at full resolution: the true (unpadded) downsampled size is what the upsampler sees
        std::vector<uint8_t> up[2];
        for (int k = 0; k < 2; ++k) {
            const Comp& c = comps[(size_t)k + 1];
            const int dw = (width * c.h + hmax - 1) / hmax, dh = (height * c.v + vmax - 1) / vmax;     // ceil: the component's own dimensions
            up[k].assign((size_t)width * height, 0);
            upsample(c, dw, dh, hmax / c.h, vmax / c.v, up[k]);
        }
        const Comp& Y = comps[0];
        for (int y = 0; y < height; ++y)
            for (int x = 0; x < width; ++x) {
                const int yy = Y.plane[(size_t)y * Y.pw + x], cb = up[0][(size_t)y * width + x] - 128, cr = up[1][(size_t)y * width + x] - 128;
                // 16-bit fixed point: FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802, FIX(0.34414) = 22554; ONE_HALF = 32768
                const int r = yy + (int)((91881L * cr + 32768) >> 16), b = yy + (int)((116130L * cb + 32768) >> 16);
                const int g = yy + (int)(((-22554L * cb + 32768) + (-46802L * cr)) >> 16);
                uint8_t* o = &rgb[((size_t)y * width + x) * 3];
                o[0] = clamp255(r); o[1] = clamp255(g); o[2] = clamp255(b);
            }
        return true;
    }
    // sx, sy in {1, 2}: triangle filter for 2:1 horizontally (3/4 nearer + 1/4 further) and 2:1 x 2:1 (the same both ways, 16ths), pixel
    // replication for 1 x 2; in: the component's plane (stride c.pw) of which dw x dh samples are real.
    void upsample(const Comp& c, int dw, int dh, int sx, int sy, std::vector<uint8_t>& out) const {
        auto in = [&](int x, int y) -> int { return c.plane[(size_t)y * c.pw + x]; };
        auto put = [&](int x, int y, int v) { if (x < width && y < height) out[(size_t)y * width + x] = (uint8_t)v; };
        if (sx == 1 && sy == 1) { for (int y = 0; y < height; ++y) for (int x = 0; x < width; ++x) put(x, y, in(x, y)); return; }
        if (sx == 2 && sy == 1) {
            for (int y = 0; y < dh; ++y) {
                if (dw == 1) { put(0, y, in(0, y)); put(1, y, in(0, y)); continue; }
                put(0, y, in(0, y)); put(1, y, (in(0, y) * 3 + in(1, y) + 2) >> 2);
                for (int x = 1; x < dw - 1; ++x) { const int v = in(x, y) * 3; put(2 * x, y, (v + in(x - 1, y) + 1) >> 2); put(2 * x + 1, y, (v + in(x + 1, y) + 2) >> 2); }
                put(2 * (dw - 1), y, (in(dw - 1, y) * 3 + in(dw - 2, y) + 1) >> 2); put(2 * (dw - 1) + 1, y, in(dw - 1, y));
            }
            return;
        }
        if (sx == 2 && sy == 2) {
            for (int y = 0; y < dh; ++y)
                for (int half = 0; half < 2; ++half) {               // output row 2y: nearer row y, further row y - 1; row 2y + 1: further row y + 1 (edges repeat)
                    const int y1 = half == 0 ? (y > 0 ? y - 1 : 0) : (y < dh - 1 ? y + 1 : dh - 1), oy = 2 * y + half;
                    auto col = [&](int x) { return in(x, y) * 3 + in(x, y1); };
                    if (dw == 1) { const int t = col(0); put(0, oy, (t * 4 + 8) >> 4); put(1, oy, (t * 4 + 7) >> 4); continue; }
                    int last = col(0), cur = col(0), next = col(1);
                    put(0, oy, (cur * 4 + 8) >> 4); put(1, oy, (cur * 3 + next + 7) >> 4);
                    for (int x = 1; x < dw - 1; ++x) {
                        last = cur; cur = next; next = col(x + 1);
                        put(2 * x, oy, (cur * 3 + last + 8) >> 4); put(2 * x + 1, oy, (cur * 3 + next + 7) >> 4);
                    }
                    last = cur; cur = next;
                    put(2 * (dw - 1), oy, (cur * 3 + last + 8) >> 4); put(2 * (dw - 1) + 1, oy, (cur * 4 + 7) >> 4);
                }
            return;
        }
        for (int y = 0; y < height; ++y) for (int x = 0; x < width; ++x) put(x, y, in(x / sx, y / sy));   // 1 x 2: replication
    }
};
constexpr int JpegDecoder::kZigzag[64];

}  // namespace

bool loadImageRgb24(const std::string& path, int& width, int& height, std::vector<uint8_t>& rgb, std::string& err) {
    if (path.rfind("http://", 0) == 0 || path.rfind("https://", 0) == 0) { err = "image texture: URLs cannot be fetched (Textures/Image.fs:11-13 uses HTTP; no network here): " + path; return false; }
    std::string bytes;
    if (!readWholeFile(path, bytes)) { err = "cannot open image file: " + path; return false; }
    std::vector<uint8_t> f(bytes.begin(), bytes.end());
    static const uint8_t png_magic[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    if (f.size() > 8 && !std::memcmp(f.data(), png_magic, 8)) return decodePng(f, width, height, rgb, err);
    if (f.size() > 2 && f[0] == 'P' && (f[1] == '2' || f[1] == '3' || f[1] == '5' || f[1] == '6')) return decodePnm(f, width, height, rgb, err);
    if (f.size() > 4 && f[0] == 0xFF && f[1] == 0xD8) {
        JpegDecoder jd(f, err);
        if (jd.parse(width, height, rgb)) return true;
        err += ": " + path;
        return false;
    }
    err = "image texture: unknown file format (PNG, JPEG and PPM/PGM are read): " + path;
    return false;
}

}  // namespace FuncTracer
