// png.hpp — the step after the render path: the reference's examples end with
// `canvas.to_image().save("image.png")` (examples/*.rs, last line; canvas.rs:31-45 + the `image`
// crate).  Canvas::to_image is the device quantiser (rt_quantize_u8); this header writes the
// resulting 8-bit RGB buffer as a PNG with nothing but zlib (libpng is not in the image).
//
// 8-bit truecolour, no interlace, filter type 0 on every row: any decoder returns exactly the bytes
// given, which is all that "<= 1 LSB in the PNG" (north star) needs.  Link with -lz.
#pragma once

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

namespace rt {

inline void png_put_u32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back(uint8_t(x >> 24)), v.push_back(uint8_t(x >> 16)), v.push_back(uint8_t(x >> 8)), v.push_back(uint8_t(x));
}
inline void png_chunk(std::vector<uint8_t>& out, const char type[4], const uint8_t* data, size_t n) {
    png_put_u32(out, static_cast<uint32_t>(n));
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    if (n) out.insert(out.end(), data, data + n);
    png_put_u32(out, static_cast<uint32_t>(crc32(0L, out.data() + at, static_cast<uInt>(n + 4))));
}

// rgb: height rows of width pixels, 3 bytes each, row 0 = top (the Canvas order).
inline std::vector<uint8_t> encode_png(const uint8_t* rgb, size_t width, size_t height, int level = 6) {
    if (!rgb || width == 0 || height == 0 || width > 0x7FFFFFFFu || height > 0x7FFFFFFFu)
        throw std::invalid_argument("encode_png: empty or oversized image");
    std::vector<uint8_t> raw;
    raw.reserve((width * 3 + 1) * height);
    for (size_t y = 0; y < height; ++y) {
        raw.push_back(0);  // filter type 0 (None)
        raw.insert(raw.end(), rgb + y * width * 3, rgb + (y + 1) * width * 3);
    }
    uLongf clen = compressBound(static_cast<uLong>(raw.size()));
    std::vector<uint8_t> z(clen);
    if (compress2(z.data(), &clen, raw.data(), static_cast<uLong>(raw.size()), level) != Z_OK)
        throw std::runtime_error("encode_png: zlib compress2 failed");
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    std::vector<uint8_t> ihdr;
    png_put_u32(ihdr, static_cast<uint32_t>(width));
    png_put_u32(ihdr, static_cast<uint32_t>(height));
    const uint8_t tail[5] = {8, 2, 0, 0, 0};  // bit depth 8, colour type 2 (RGB), deflate, adaptive filtering, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    png_chunk(out, "IHDR", ihdr.data(), ihdr.size());
    png_chunk(out, "IDAT", z.data(), clen);
    png_chunk(out, "IEND", nullptr, 0);
    return out;
}

inline void save_png(const std::string& path, const uint8_t* rgb, size_t width, size_t height) {
    const std::vector<uint8_t> png = encode_png(rgb, width, height);
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("save_png: cannot open " + path);
    const size_t n = std::fwrite(png.data(), 1, png.size(), f);
    const int rc = std::fclose(f);
    if (n != png.size() || rc != 0) throw std::runtime_error("save_png: short write to " + path);
}

}  // namespace rt
