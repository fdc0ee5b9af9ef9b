// png.cpp -- minimal PNG encoder for the headless output stage (the reference hands its
// pixels to stbi_write_png, Raytracer.cpp:104).  Only what a viewer needs: IHDR, one IDAT
// holding a zlib stream of the filter-0 scanlines, IEND.
#include <zlib.h>

#include <cstdio>
#include <vector>

#include "scene.hpp"

namespace rth {
namespace {

void put32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

void chunk(std::vector<uint8_t>& file, const char type[4], const uint8_t* data, size_t n) {
    put32(file, (uint32_t)n);
    size_t at = file.size();
    file.insert(file.end(), type, type + 4);
    if (n) file.insert(file.end(), data, data + n);
    put32(file, (uint32_t)crc32(0L, file.data() + at, (uInt)(n + 4)));
}

}  // namespace

bool writePng(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height) {
    if (!path || !rgb8 || width == 0 || height == 0) return false;
    const size_t stride = (size_t)width * 3;
    std::vector<uint8_t> raw((stride + 1) * height);
    for (uint32_t y = 0; y < height; ++y) {
        raw[(stride + 1) * y] = 0;                                  // filter type: none
        std::copy(rgb8 + stride * y, rgb8 + stride * (y + 1), raw.begin() + (stride + 1) * y + 1);
    }
    uLongf zn = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zn);
    if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;

    std::vector<uint8_t> file = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<uint8_t> ihdr;
    put32(ihdr, width); put32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit RGB
    chunk(file, "IHDR", ihdr.data(), ihdr.size());
    chunk(file, "IDAT", z.data(), zn);
    chunk(file, "IEND", nullptr, 0);
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    bool ok = std::fwrite(file.data(), 1, file.size(), f) == file.size();
    ok = std::fclose(f) == 0 && ok;
    return ok;
}

}  // namespace rth
