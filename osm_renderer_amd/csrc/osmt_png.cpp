/*
 * osmt_png.cpp — host-side PNG encoding of a rendered tile: the counterpart of
 * draw::png_writer::rgb_triples_to_png (src/draw/png_writer.rs:4-21), the last step of
 * Drawer::draw_tile (src/draw/drawer.rs:40-58).  SURVEY.md 8(f) row N3 ("next" after the raster
 * path): only the DECODED pixels are pinned by the reference's tests (tests/test_rendering.rs:
 * 15-23,46-51), so the encoder is free to choose filters / compression level.
 *
 * RGB8 (colour type 2), filter 0 on every row, one zlib stream in a single IDAT chunk.
 */
#include <zlib.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/osmtile.h"

extern int osmt_fail_public(int code, const char* msg);

namespace {
void put32(uint8_t* p, uint32_t v) {
    p[0] = (uint8_t)(v >> 24);
    p[1] = (uint8_t)(v >> 16);
    p[2] = (uint8_t)(v >> 8);
    p[3] = (uint8_t)v;
}
/* writes one chunk at `dst`, returns its total size (12 + len) */
size_t chunk(uint8_t* dst, const char type[4], const uint8_t* data, uint32_t len) {
    put32(dst, len);
    memcpy(dst + 4, type, 4);
    if (len) memcpy(dst + 8, data, len);
    uLong c = crc32(0L, Z_NULL, 0);
    c = crc32(c, dst + 4, 4 + len);
    put32(dst + 8 + len, (uint32_t)c);
    return 12u + len;
}
}  // namespace

extern "C" {

size_t osmt_png_bound(uint32_t width, uint32_t height) {
    const size_t raw = ((size_t)width * 3 + 1) * height;
    return 8 + (12 + 13) + (12 + compressBound((uLong)raw)) + 12;
}

int osmt_encode_png(const uint8_t* rgba, uint32_t width, uint32_t height, size_t row_stride_bytes, int level,
                    uint8_t* out_png, size_t out_capacity, size_t* out_len) {
    if (!rgba || !out_png || !out_len) return osmt_fail_public(OSMT_INVALID_ARG, "osmt_encode_png: NULL argument");
    if (width == 0 || height == 0 || row_stride_bytes < (size_t)width * 4)
        return osmt_fail_public(OSMT_INVALID_ARG, "osmt_encode_png: bad dimensions / stride");
    if (level < 0 || level > 9) level = Z_DEFAULT_COMPRESSION;
    if (out_capacity < osmt_png_bound(width, height))
        return osmt_fail_public(OSMT_INVALID_ARG, "osmt_encode_png: output buffer smaller than osmt_png_bound()");
    /* scanlines: filter byte 0 + RGB (alpha dropped: the framebuffer's A is the constant 255) */
    const size_t line = (size_t)width * 3 + 1;
    std::vector<uint8_t> raw;
    try { /* no C++ exception may cross the C boundary */
        raw.resize(line * height);
    } catch (...) {
        return osmt_fail_public(OSMT_OOM, "osmt_encode_png: out of host memory");
    }
    for (uint32_t y = 0; y < height; ++y) {
        uint8_t* d = raw.data() + y * line;
        const uint8_t* s = rgba + (size_t)y * row_stride_bytes;
        *d++ = 0;
        for (uint32_t x = 0; x < width; ++x) {
            d[0] = s[0];
            d[1] = s[1];
            d[2] = s[2];
            d += 3;
            s += 4;
        }
    }
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    size_t off = 0;
    memcpy(out_png, sig, 8);
    off += 8;
    uint8_t ihdr[13];
    put32(ihdr, width);
    put32(ihdr + 4, height);
    ihdr[8] = 8;  /* bit depth */
    ihdr[9] = 2;  /* colour type RGB (png_writer.rs:8 ColorType::Rgb) */
    ihdr[10] = 0; /* deflate */
    ihdr[11] = 0; /* adaptive filtering (type 0 used on every row) */
    ihdr[12] = 0; /* no interlace */
    off += chunk(out_png + off, "IHDR", ihdr, 13);
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z;
    try {
        z.resize(zlen);
    } catch (...) {
        return osmt_fail_public(OSMT_OOM, "osmt_encode_png: out of host memory");
    }
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), level) != Z_OK)
        return osmt_fail_public(OSMT_HIP_ERROR, "osmt_encode_png: zlib compress2 failed");
    off += chunk(out_png + off, "IDAT", z.data(), (uint32_t)zlen);
    off += chunk(out_png + off, "IEND", nullptr, 0);
    *out_len = off;
    return OSMT_OK;
}

} /* extern "C" */
