// image_loader.cpp — PNG and baseline-JPEG decoding to the RGB f32 buffer the reference keeps
// for image textures (src/buffer.rs:30-48: `ImageReader::open(..).decode()?.into_rgb32f()`).
//
// The reference delegates decoding to the `image` crate (not vendored).  Restated here from the
// file-format specifications: PNG (RFC 2083: zlib stream, five scan-line filters, colour types
// 0/2/3/4/6, 8 and 16 bit, non-interlaced) and JPEG (ITU T.81 baseline sequential DCT, Huffman,
// 8-bit, 1 or 3 components, any sampling factors, restart intervals).  Conversion to f32 follows
// `into_rgb32f`: 8-bit x/255, 16-bit x/65535, grey replicated, alpha dropped.
// PNG decoding is exact by construction.  JPEG is "parity unpinned": IDCT rounding and chroma
// upsampling (sample replication here) may differ from the crate's decoder by an 8-bit step.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <zlib.h>

#include "host_internal.h"

namespace rth {
namespace {

bool read_file(const std::string& path, std::vector<uint8_t>* out, std::string* err) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { *err = "No such file or directory (os error 2): " + path; return false; }
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out->resize(n > 0 ? size_t(n) : 0);
    size_t got = out->empty() ? 0 : std::fread(out->data(), 1, out->size(), f);
    std::fclose(f);
    if (got != out->size()) { *err = "short read: " + path; return false; }
    return true;
}

uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

// ------------------------------------------------------------------ PNG
int paeth(int a, int b, int c) {
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

bool decode_png(const std::vector<uint8_t>& d, std::vector<float>* rgb, uint32_t* w, uint32_t* h, std::string* err) {
    size_t pos = 8;
    uint32_t width = 0, height = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool seen_ihdr = false;
    while (pos + 12 <= d.size()) {
        uint32_t len = be32(&d[pos]);
        const uint8_t* type = &d[pos + 4];
        if (pos + 12 + size_t(len) > d.size()) { *err = "PNG: truncated chunk"; return false; }
        const uint8_t* body = &d[pos + 8];
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
            width = be32(body); height = be32(body + 4);
            depth = body[8]; ctype = body[9]; interlace = body[12];
            seen_ihdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            plte.assign(body, body + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + size_t(len);
    }
    if (!seen_ihdr || width == 0 || height == 0) { *err = "PNG: missing IHDR"; return false; }
    if (interlace != 0) { *err = "PNG: interlaced images are not supported"; return false; }
    int channels;
    switch (ctype) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: *err = "PNG: bad colour type"; return false;
    }
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) {
        *err = "PNG: unsupported bit depth"; return false;
    }
    if (ctype == 3 && depth == 16) { *err = "PNG: bad palette depth"; return false; }
    size_t bpp_bits = size_t(channels) * size_t(depth);
    size_t stride = (size_t(width) * bpp_bits + 7) / 8;
    size_t bpp = (bpp_bits + 7) / 8;  // filter unit, at least 1
    std::vector<uint8_t> raw((stride + 1) * size_t(height));
    uLongf raw_len = uLongf(raw.size());
    int zr = uncompress(raw.data(), &raw_len, idat.data(), uLong(idat.size()));
    if (zr != Z_OK || raw_len != raw.size()) { *err = "PNG: zlib stream error"; return false; }
    // unfilter in place (RFC 2083 section 6)
    std::vector<uint8_t> img(stride * size_t(height));
    for (uint32_t y = 0; y < height; y++) {
        const uint8_t* in = &raw[(stride + 1) * y];
        uint8_t* cur = &img[stride * y];
        const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
        int ft = in[0];
        for (size_t x = 0; x < stride; x++) {
            int a = x >= bpp ? cur[x - bpp] : 0;
            int b = up ? up[x] : 0;
            int c = (up && x >= bpp) ? up[x - bpp] : 0;
            int v = in[1 + x];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) / 2; break;
                case 4: v += paeth(a, b, c); break;
                default: *err = "PNG: bad filter type"; return false;
            }
            cur[x] = uint8_t(v);
        }
    }
    rgb->assign(size_t(width) * height * 3, 0.f);
    auto sample = [&](const uint8_t* row, size_t idx) -> uint32_t {  // idx-th sample of the row
        if (depth == 8) return row[idx];
        if (depth == 16) return (uint32_t(row[2 * idx]) << 8) | row[2 * idx + 1];
        size_t bit = idx * size_t(depth);
        return (row[bit / 8] >> (8 - depth - int(bit % 8))) & ((1u << depth) - 1u);
    };
    const float maxv = depth == 16 ? 65535.0f : 255.0f;
    for (uint32_t y = 0; y < height; y++) {
        const uint8_t* row = &img[stride * y];
        for (uint32_t x = 0; x < width; x++) {
            float* o = &(*rgb)[(size_t(y) * width + x) * 3];
            if (ctype == 3) {
                uint32_t i = sample(row, x);
                if (size_t(i) * 3 + 2 >= plte.size()) { *err = "PNG: palette index out of range"; return false; }
                for (int k = 0; k < 3; k++) o[k] = float(plte[i * 3 + k]) / 255.0f;
            } else if (ctype == 0 || ctype == 4) {
                uint32_t g = sample(row, size_t(x) * channels);
                float f;
                if (depth < 8) f = float(g * (255u / ((1u << depth) - 1u))) / 255.0f;  // scaled to 8 bit first
                else f = float(g) / maxv;
                o[0] = o[1] = o[2] = f;
            } else {
                for (int k = 0; k < 3; k++) o[k] = float(sample(row, size_t(x) * channels + k)) / maxv;
            }
        }
    }
    *w = width; *h = height;
    return true;
}

// ------------------------------------------------------------------ JPEG (baseline sequential, Huffman)
struct Huff {
    // canonical decoding tables (T.81 F.2.2.3)
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    bool present = false;
};
struct Comp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int blocks_w = 0, blocks_h = 0;  // allocated size in blocks
    std::vector<uint8_t> plane;      // blocks_w*8 x blocks_h*8
    int pred = 0;
};
struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int nbits = 0;
    bool hit_marker = false;
    int bit() {
        if (nbits == 0) {
            uint8_t b = 0;
            if (p < end && !hit_marker) {
                b = *p++;
                if (b == 0xFF) {
                    if (p < end && *p == 0x00) p++;
                    else { hit_marker = true; p--; b = 0; }
                }
            }
            acc = b;
            nbits = 8;
        }
        nbits--;
        return (acc >> nbits) & 1;
    }
    int bits(int n) { int v = 0; for (int i = 0; i < n; i++) v = (v << 1) | bit(); return v; }
    void reset() { nbits = 0; hit_marker = false; }
};
int huff_decode(BitReader& br, const Huff& h) {
    int code = 0;
    for (int len = 1; len <= 16; len++) {
        code = (code << 1) | br.bit();
        if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.vals[h.valptr[len] + code - h.mincode[len]];
    }
    return -1;
}
int extend(int v, int t) { return (t && v < (1 << (t - 1))) ? v - (1 << t) + 1 : v; }

const int kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                         41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                         15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

void idct8x8(const float* in, uint8_t* out, int stride) {
    static float c[8][8];
    static bool init = false;
    if (!init) {
        for (int x = 0; x < 8; x++)
            for (int u = 0; u < 8; u++) c[x][u] = float((u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0));
        init = true;
    }
    float tmp[64];
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) {
            float s = 0.f;
            for (int u = 0; u < 8; u++) s += c[x][u] * in[y * 8 + u];
            tmp[y * 8 + x] = s;
        }
    for (int x = 0; x < 8; x++)
        for (int y = 0; y < 8; y++) {
            float s = 0.f;
            for (int v = 0; v < 8; v++) s += c[y][v] * tmp[v * 8 + x];
            int q = int(std::lround(s + 128.0f));
            out[y * stride + x] = uint8_t(q < 0 ? 0 : (q > 255 ? 255 : q));
        }
}

bool decode_jpeg(const std::vector<uint8_t>& d, std::vector<float>* rgb, uint32_t* w, uint32_t* h, std::string* err) {
    uint16_t qt[4][64] = {};
    Huff hdc[4], hac[4];
    std::vector<Comp> comps;
    int width = 0, height = 0, restart = 0;
    bool adobe = false;
    int adobe_transform = -1;
    size_t pos = 2;
    bool got_sof = false;
    while (pos + 4 <= d.size()) {
        if (d[pos] != 0xFF) { pos++; continue; }
        int m = d[pos + 1];
        if (m == 0xFF) { pos++; continue; }
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) { pos += 2; continue; }
        if (m == 0xD9) break;
        size_t len = (size_t(d[pos + 2]) << 8) | d[pos + 3];
        if (len < 2 || pos + 2 + len > d.size()) { *err = "JPEG: truncated segment"; return false; }
        const uint8_t* s = &d[pos + 4];
        size_t n = len - 2;
        if (m == 0xDB) {  // DQT
            size_t i = 0;
            while (i < n) {
                int pq = s[i] >> 4, tq = s[i] & 15;
                i++;
                if (tq > 3 || i + (pq ? 128u : 64u) > n) { *err = "JPEG: bad DQT"; return false; }
                for (int k = 0; k < 64; k++) {
                    if (pq) { qt[tq][kZigzag[k]] = uint16_t((s[i] << 8) | s[i + 1]); i += 2; }
                    else qt[tq][kZigzag[k]] = s[i++];
                }
            }
        } else if (m == 0xC4) {  // DHT
            size_t i = 0;
            while (i + 17 <= n) {
                int tc = s[i] >> 4, th = s[i] & 15;
                if (th > 3 || tc > 1) { *err = "JPEG: bad DHT"; return false; }
                Huff& t = tc ? hac[th] : hdc[th];
                int counts[17] = {0};
                int total = 0;
                for (int k = 1; k <= 16; k++) { counts[k] = s[i + k]; total += counts[k]; }
                i += 17;
                if (total > 256 || i + size_t(total) > n) { *err = "JPEG: bad DHT"; return false; }
                std::memcpy(t.vals, s + i, size_t(total));
                i += size_t(total);
                int code = 0, k = 0;
                for (int l = 1; l <= 16; l++) {
                    t.valptr[l] = k;
                    t.mincode[l] = code;
                    code += counts[l];
                    k += counts[l];
                    t.maxcode[l] = counts[l] ? code - 1 : -1;
                    code <<= 1;
                }
                t.present = true;
            }
        } else if (m == 0xC0 || m == 0xC1) {  // SOF0 / SOF1 (Huffman, sequential)
            if (n < 6) { *err = "JPEG: truncated SOF"; return false; }
            if (s[0] != 8) { *err = "JPEG: only 8-bit precision is supported"; return false; }
            height = (s[1] << 8) | s[2];
            width = (s[3] << 8) | s[4];
            int nc = s[5];
            if (!(nc == 1 || nc == 3)) { *err = "JPEG: only 1- or 3-component images are supported"; return false; }
            if (n < 6 + 3 * size_t(nc)) { *err = "JPEG: truncated SOF"; return false; }
            comps.resize(size_t(nc));
            for (int c = 0; c < nc; c++) {
                comps[c].id = s[6 + 3 * c];
                comps[c].h = s[7 + 3 * c] >> 4;
                comps[c].v = s[7 + 3 * c] & 15;
                comps[c].tq = s[8 + 3 * c];
                if (comps[c].h < 1 || comps[c].v < 1 || comps[c].h > 4 || comps[c].v > 4 || comps[c].tq > 3) { *err = "JPEG: bad SOF"; return false; }
            }
            got_sof = true;
        } else if (m == 0xC2 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
            *err = "JPEG: progressive / arithmetic / lossless coding is not supported (baseline only)";
            return false;
        } else if (m == 0xDD) {
            if (n < 2) { *err = "JPEG: truncated DRI"; return false; }
            restart = (s[0] << 8) | s[1];
        } else if (m == 0xEE && n >= 12 && !std::memcmp(s, "Adobe", 5)) {
            adobe = true;
            adobe_transform = s[11];
        } else if (m == 0xDA) {  // SOS: decode the (single, interleaved or not) scan
            if (!got_sof || width <= 0 || height <= 0) { *err = "JPEG: SOS before SOF"; return false; }
            if (n < 1 || n < 1 + 2 * size_t(s[0])) { *err = "JPEG: truncated SOS"; return false; }
            int ns = s[0];
            std::vector<int> order;
            for (int k = 0; k < ns; k++) {
                int cid = s[1 + 2 * k];
                int ci = -1;
                for (size_t c = 0; c < comps.size(); c++) if (comps[c].id == cid) ci = int(c);
                if (ci < 0) { *err = "JPEG: bad SOS"; return false; }
                comps[size_t(ci)].td = s[2 + 2 * k] >> 4;
                comps[size_t(ci)].ta = s[2 + 2 * k] & 15;
                if (comps[size_t(ci)].td > 3 || comps[size_t(ci)].ta > 3) { *err = "JPEG: bad SOS (table selector)"; return false; }
                order.push_back(ci);
            }
            int hmax = 1, vmax = 1;
            for (auto& c : comps) { hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v); }
            int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
            for (auto& c : comps) {
                if (c.plane.empty()) {
                    c.blocks_w = mcux * c.h;
                    c.blocks_h = mcuy * c.v;
                    c.plane.assign(size_t(c.blocks_w) * 8 * size_t(c.blocks_h) * 8, 128);
                }
                c.pred = 0;
            }
            BitReader br{&d[pos + 2 + len], d.data() + d.size()};
            bool single = ns == 1;
            int units_x = mcux, units_y = mcuy;
            if (single) {  // non-interleaved: the MCU is one block, only blocks inside the image are coded
                Comp& c = comps[size_t(order[0])];
                units_x = ((width * c.h + hmax - 1) / hmax + 7) / 8;
                units_y = ((height * c.v + vmax - 1) / vmax + 7) / 8;
            }
            int count = 0;
            for (int my = 0; my < units_y; my++)
                for (int mx = 0; mx < units_x; mx++) {
                    if (restart && count && count % restart == 0) {
                        // skip to the RSTn marker and reset
                        br.reset();
                        const uint8_t* q = br.p;
                        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) q++;
                        br.p = q + 2 <= br.end ? q + 2 : br.end;
                        for (auto& c : comps) c.pred = 0;
                    }
                    count++;
                    for (int oi : order) {
                        Comp& c = comps[size_t(oi)];
                        int bh = single ? 1 : c.h, bv = single ? 1 : c.v;
                        for (int by = 0; by < bv; by++)
                            for (int bx = 0; bx < bh; bx++) {
                                float coef[64] = {0};
                                const Huff& dc = hdc[c.td];
                                const Huff& ac = hac[c.ta];
                                if (!dc.present || !ac.present) { *err = "JPEG: missing Huffman table"; return false; }
                                int t = huff_decode(br, dc);
                                if (t < 0 || t > 11) { *err = "JPEG: bad DC code"; return false; }
                                int diff = t ? extend(br.bits(t), t) : 0;
                                c.pred += diff;
                                coef[0] = float(c.pred * int(qt[c.tq][0]));
                                for (int k = 1; k < 64;) {
                                    int rs = huff_decode(br, ac);
                                    if (rs < 0) { *err = "JPEG: bad AC code"; return false; }
                                    int r = rs >> 4, sz = rs & 15;
                                    if (sz == 0) {
                                        if (r == 15) { k += 16; continue; }
                                        break;  // EOB
                                    }
                                    k += r;
                                    if (k > 63) { *err = "JPEG: coefficient index out of range"; return false; }
                                    int v = extend(br.bits(sz), sz);
                                    coef[kZigzag[k]] = float(v * int(qt[c.tq][kZigzag[k]]));
                                    k++;
                                }
                                int bxa = single ? mx : mx * c.h + bx, bya = single ? my : my * c.v + by;
                                if (bxa < c.blocks_w && bya < c.blocks_h)
                                    idct8x8(coef, &c.plane[(size_t(bya) * 8) * (size_t(c.blocks_w) * 8) + size_t(bxa) * 8], c.blocks_w * 8);
                            }
                    }
                }
            // continue after the entropy-coded data
            const uint8_t* q = br.p;
            while (q + 1 < br.end && !(q[0] == 0xFF && q[1] != 0x00 && !(q[1] >= 0xD0 && q[1] <= 0xD7))) q++;
            pos = size_t(q - d.data());
            continue;
        }
        pos += 2 + len;
    }
    if (!got_sof || comps.empty() || comps[0].plane.empty()) { *err = "JPEG: no image data"; return false; }
    int hmax = 1, vmax = 1;
    for (auto& c : comps) { hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v); }
    rgb->assign(size_t(width) * size_t(height) * 3, 0.f);
    bool ycc = comps.size() == 3 && !(adobe && adobe_transform == 0);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            int v[3] = {0, 0, 0};
            for (size_t c = 0; c < comps.size(); c++) {
                const Comp& cc = comps[c];
                int sx = x * cc.h / hmax, sy = y * cc.v / vmax;  // sample replication
                v[c] = cc.plane[size_t(sy) * (size_t(cc.blocks_w) * 8) + size_t(sx)];
            }
            float* o = &(*rgb)[(size_t(y) * size_t(width) + size_t(x)) * 3];
            if (comps.size() == 1) {
                o[0] = o[1] = o[2] = float(v[0]) / 255.0f;
            } else if (ycc) {  // JFIF: ITU-R BT.601 full range
                float Y = float(v[0]), cb = float(v[1]) - 128.0f, cr = float(v[2]) - 128.0f;
                float r = Y + 1.402f * cr, g = Y - 0.344136f * cb - 0.714136f * cr, b = Y + 1.772f * cb;
                float c3[3] = {r, g, b};
                for (int k = 0; k < 3; k++) {
                    int q = int(std::lround(c3[k]));
                    o[k] = float(q < 0 ? 0 : (q > 255 ? 255 : q)) / 255.0f;
                }
            } else {
                for (int k = 0; k < 3; k++) o[k] = float(v[k]) / 255.0f;
            }
        }
    *w = uint32_t(width); *h = uint32_t(height);
    return true;
}

}  // namespace

bool load_image_rgb32f(const std::string& path, std::vector<float>* rgb, uint32_t* width, uint32_t* height, std::string* err) {
    std::vector<uint8_t> d;
    if (!read_file(path, &d, err)) return false;
    static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (d.size() >= 8 && !std::memcmp(d.data(), png_sig, 8)) return decode_png(d, rgb, width, height, err);
    if (d.size() >= 3 && d[0] == 0xFF && d[1] == 0xD8) return decode_jpeg(d, rgb, width, height, err);
    *err = "The image format could not be determined: " + path;
    return false;
}

}  // namespace rth
