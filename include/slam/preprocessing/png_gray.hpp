// png_gray.hpp -- minimal PNG reader for the frame source: 8-bit,
// non-interlaced gray / RGB / RGBA PNGs via zlib.  Replaces cv::imread(..., IMREAD_GRAYSCALE)
// at the reference's call sites (test_feature_detector.cpp:16, preprocessor.cpp:115).
// Gray PNGs are returned as stored.  Colour PNGs are converted with OpenCV's fixed-point
// BGR2GRAY weights (R 4899, G 9617, B 1868, >> 14) -- third-party arithmetic, parity unpinned
// (SURVEY.md 8c item 5); the parity fixtures are all gray.
#pragma once

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include <slam/common/common.hpp>

namespace slamx_tools {

inline uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

inline slam::EigenGrayMatrix read_png_gray(const std::string& path) {
    std::ifstream in(path, std::ios::binary);
    if (!in) {
        throw std::runtime_error("Failed to read image from file: " + path);
    }
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 33 || !std::equal(sig, sig + 8, file.begin())) {
        throw std::runtime_error("not a PNG file: " + path);
    }
    uint32_t width = 0, height = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat;
    size_t pos = 8;
    while (pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        const std::string type(reinterpret_cast<const char*>(&file[pos + 4]), 4);
        if (pos + 12 + len > file.size()) {
            break;
        }
        const uint8_t* data = &file[pos + 8];
        if (type == "IHDR") {
            width = be32(data);
            height = be32(data + 4);
            depth = data[8];
            ctype = data[9];
            interlace = data[12];
        } else if (type == "IDAT") {
            idat.insert(idat.end(), data, data + len);
        } else if (type == "IEND") {
            break;
        }
        pos += 12 + len;
    }
    int channels = 0;
    switch (ctype) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: throw std::runtime_error("unsupported PNG colour type (palette): " + path);
    }
    if (depth != 8 || interlace != 0 || width == 0 || height == 0) {
        throw std::runtime_error("unsupported PNG (need 8-bit, non-interlaced): " + path);
    }
    const size_t stride = static_cast<size_t>(width) * channels;
    std::vector<uint8_t> raw((stride + 1) * height);
    uLongf raw_len = static_cast<uLongf>(raw.size());
    if (uncompress(raw.data(), &raw_len, idat.data(), static_cast<uLong>(idat.size())) != Z_OK || raw_len != raw.size()) {
        throw std::runtime_error("PNG inflate failed: " + path);
    }
    std::vector<uint8_t> pix(stride * height);
    const int bpp = channels;
    for (uint32_t y = 0; y < height; ++y) {
        const uint8_t ft = raw[y * (stride + 1)];
        const uint8_t* src = &raw[y * (stride + 1) + 1];
        uint8_t* dst = &pix[y * stride];
        const uint8_t* up = y > 0 ? &pix[(y - 1) * stride] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= static_cast<size_t>(bpp) ? dst[i - bpp] : 0;
            const int b = up ? up[i] : 0;
            const int c = (up && i >= static_cast<size_t>(bpp)) ? up[i - bpp] : 0;
            int pred = 0;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) / 2; break;
                case 4: {
                    const int p = a + b - c;
                    const int pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
                default: throw std::runtime_error("bad PNG filter: " + path);
            }
            dst[i] = static_cast<uint8_t>(src[i] + pred);
        }
    }
    slam::EigenGrayMatrix img(static_cast<std::ptrdiff_t>(height), static_cast<std::ptrdiff_t>(width));
    for (uint32_t y = 0; y < height; ++y) {
        for (uint32_t x = 0; x < width; ++x) {
            const uint8_t* p = &pix[y * stride + static_cast<size_t>(x) * channels];
            uint8_t g;
            if (channels <= 2) {
                g = p[0];
            } else {
                g = static_cast<uint8_t>((p[0] * 4899 + p[1] * 9617 + p[2] * 1868 + 8192) >> 14);
            }
            img(y, x) = g;
        }
    }
    return img;
}

}  // namespace slamx_tools
