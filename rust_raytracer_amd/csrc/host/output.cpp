// Output stage — behavioural restatement of reference src/output.rs:23-49 with
// `writer.tonemap = tonemap_aces` (src/main.rs:81, src/tonemapping/aces.rs:5-33):
// linear radiance -> ACES filmic fit -> sRGB OETF (gamma 1/2.4, 0.0031308 knee)
// -> `(x * 255.999) as u8` -> 8-bit RGB PNG.  PNG encoding uses zlib (the reference uses
// the `image` crate); pixel values are what is pinned, not the compressed byte stream.
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>

#include "host_internal.h"

namespace rth {

// aces.rs:5-18 (the stray 0.1 at [3][3] only touches w, which is never read)
static const double kAcesIn[9] = {0.59719, 0.35458, 0.04823, 0.07600, 0.90834, 0.01566, 0.02840, 0.13383, 0.83777};
static const double kAcesOut[9] = {1.60475, -0.53108, -0.07367, -0.10208, 1.10813, -0.00605, -0.00327, -0.07276, 1.07602};

static inline double clamp01(double x) {
    // f64::clamp: NaN stays NaN
    if (x < 0.0) return 0.0;
    if (x > 1.0) return 1.0;
    return x;
}

static inline uint8_t to_u8(double x) {
    double s = x * 255.999;
    if (!(s > 0.0)) return 0;  // saturating `as u8`, NaN -> 0
    if (s >= 255.0) return 255;
    return uint8_t(s);
}

void tonemap_rgb8(const double* rgba, uint32_t w, uint32_t h, uint8_t* rgb) {
    const double gamma = 1.0 / 2.4;  // output.rs:7
    for (size_t i = 0; i < size_t(w) * h; i++) {
        const double* p = rgba + 4 * i;
        double c[3], f[3], o[3];
        for (int r = 0; r < 3; r++)  // Mat4 * Vec4, mat4.rs:345-351 (w term is 0 * w)
            c[r] = kAcesIn[3 * r] * p[0] + kAcesIn[3 * r + 1] * p[1] + kAcesIn[3 * r + 2] * p[2] + 0.0 * p[3];
        for (int k = 0; k < 3; k++) {  // rrt_and_odt_fit, aces.rs:20-24
            double a = c[k] * (c[k] + 0.0245786) - 0.000090537;
            double b = c[k] * (c[k] * 0.983729 + 0.4329510) + 0.238081;
            f[k] = a / b;
        }
        for (int r = 0; r < 3; r++)
            o[r] = kAcesOut[3 * r] * f[0] + kAcesOut[3 * r + 1] * f[1] + kAcesOut[3 * r + 2] * f[2] + 0.0 * 0.0;
        for (int k = 0; k < 3; k++) {
            double x = clamp01(o[k]);
            double s = x < 0.0031308 ? x * 12.92 : std::pow(x, gamma) * 1.055 - 0.055;  // output.rs:42-49
            rgb[3 * i + k] = to_u8(s);
        }
    }
}

static void put_u32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back(uint8_t(x >> 24)); v.push_back(uint8_t(x >> 16)); v.push_back(uint8_t(x >> 8)); v.push_back(uint8_t(x));
}

static void put_chunk(std::vector<uint8_t>& out, const char type[4], const uint8_t* data, size_t n) {
    put_u32(out, uint32_t(n));
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    if (n) out.insert(out.end(), data, data + n);
    uint32_t crc = uint32_t(crc32(0L, out.data() + start, uInt(n + 4)));
    put_u32(out, crc);
}

bool write_png_rgb8(const std::string& path, const uint8_t* rgb, uint32_t w, uint32_t h, std::string* err) {
    std::vector<uint8_t> raw;
    raw.reserve(size_t(h) * (size_t(w) * 3 + 1));
    for (uint32_t y = 0; y < h; y++) {
        raw.push_back(0);  // filter type None
        raw.insert(raw.end(), rgb + size_t(y) * w * 3, rgb + size_t(y + 1) * w * 3);
    }
    uLongf clen = compressBound(uLong(raw.size()));
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), uLong(raw.size()), 6) != Z_OK) {
        *err = "zlib compress failed";
        return false;
    }
    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_u32(ihdr, w);
    put_u32(ihdr, h);
    ihdr.push_back(8);  // bit depth
    ihdr.push_back(2);  // colour type RGB
    ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    put_chunk(png, "IHDR", ihdr.data(), ihdr.size());
    put_chunk(png, "IDAT", comp.data(), clen);
    put_chunk(png, "IEND", nullptr, 0);
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) {
        *err = "cannot open " + path + " for writing";
        return false;
    }
    bool ok = std::fwrite(png.data(), 1, png.size(), f) == png.size();
    std::fclose(f);
    if (!ok) *err = "short write on " + path;
    return ok;
}

}  // namespace rth
