// ImageLoader.cpp — the harness counterpart of `Image.Load<Rgb24>` in Textures/Image.fs:21-24: decode a local image
// file to tightly packed RGB bytes, row 0 = top.  The reference delegates to SixLabors.ImageSharp (any format, local
// file or HTTP); this loader reads what can be decoded without it: PNG (8 / 16 bit, grey, grey+alpha, RGB, RGBA,
// palette; non-interlaced; inflate by the system zlib) and binary / ASCII PPM / PGM.  Alpha is dropped and 16-bit
// samples keep their high byte, which is what a conversion to Rgb24 does.  JPEG and URLs are refused with a message.
#include <zlib.h>

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

}  // namespace

bool loadImageRgb24(const std::string& path, int& width, int& height, std::vector<uint8_t>& rgb, std::string& err) {
    if (path.rfind("http://", 0) == 0 || path.rfind("https://", 0) == 0) { err = "image texture: URLs cannot be fetched (Textures/Image.fs:11-13 uses HTTP; no network here): " + path; return false; }
    std::string bytes;
    if (!readWholeFile(path, bytes)) { err = "cannot open image file: " + path; return false; }
    std::vector<uint8_t> f(bytes.begin(), bytes.end());
    static const uint8_t png_magic[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    if (f.size() > 8 && !std::memcmp(f.data(), png_magic, 8)) return decodePng(f, width, height, rgb, err);
    if (f.size() > 2 && f[0] == 'P' && (f[1] == '2' || f[1] == '3' || f[1] == '5' || f[1] == '6')) return decodePnm(f, width, height, rgb, err);
    if (f.size() > 2 && f[0] == 0xFF && f[1] == 0xD8) { err = "image texture: JPEG needs a decoder this build does not carry (convert to PNG or PPM): " + path; return false; }
    err = "image texture: unknown file format (PNG and PPM/PGM are read): " + path;
    return false;
}

}  // namespace FuncTracer
