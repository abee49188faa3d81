// Host-side f64 vector / matrix helpers for scene construction (one-shot code).
// Semantics follow reference src/vec4.rs and src/mat4.rs: dot/length/cross ignore w,
// matrices are row-major 4x4, Mat4*Vec4 uses all four components (mat4.rs:342-353).
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <limits>

namespace rth {

struct V4 {
    double x = 0, y = 0, z = 0, w = 0;
    double operator[](int i) const { return i == 0 ? x : i == 1 ? y : i == 2 ? z : w; }
};
inline V4 vec(double x, double y, double z) { return {x, y, z, 0.0}; }
inline V4 point(double x, double y, double z) { return {x, y, z, 1.0}; }
inline V4 operator+(V4 a, V4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline V4 operator-(V4 a, V4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline V4 operator-(V4 a) { return {-a.x, -a.y, -a.z, -a.w}; }
inline V4 operator*(V4 a, double s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline V4 operator*(V4 a, V4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline V4 operator/(V4 a, double s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }
inline double dot(V4 a, V4 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double length_squared(V4 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
inline double length(V4 a) { return std::sqrt(length_squared(a)); }
inline V4 cross(V4 a, V4 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x, 0.0};
}
inline V4 to_unit(V4 a) { return a / length(a); }

struct M4 {
    double m[16];
};
inline M4 m4_identity() { return {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}}; }
inline M4 m4_translation(double x, double y, double z) {
    return {{1, 0, 0, x, 0, 1, 0, y, 0, 0, 1, z, 0, 0, 0, 1}};
}
inline M4 m4_scale(double x, double y, double z) {
    return {{x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1}};
}
inline M4 m4_rotate_x(double t) {
    double s = std::sin(t), c = std::cos(t);
    return {{1, 0, 0, 0, 0, c, -s, 0, 0, s, c, 0, 0, 0, 0, 1}};
}
inline M4 m4_rotate_y(double t) {
    double s = std::sin(t), c = std::cos(t);
    return {{c, 0, s, 0, 0, 1, 0, 0, -s, 0, c, 0, 0, 0, 0, 1}};
}
inline M4 m4_rotate_z(double t) {
    double s = std::sin(t), c = std::cos(t);
    return {{c, -s, 0, 0, s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
}
// row(i) . column(j), summed left to right like mat4.rs:326-336.
inline M4 operator*(const M4& a, const M4& b) {
    M4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.m[i * 4 + j] = a.m[i * 4 + 0] * b.m[0 + j] + a.m[i * 4 + 1] * b.m[4 + j] +
                             a.m[i * 4 + 2] * b.m[8 + j] + a.m[i * 4 + 3] * b.m[12 + j];
    return r;
}
inline V4 operator*(const M4& a, V4 v) {
    return {a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z + a.m[3] * v.w,
            a.m[4] * v.x + a.m[5] * v.y + a.m[6] * v.z + a.m[7] * v.w,
            a.m[8] * v.x + a.m[9] * v.y + a.m[10] * v.z + a.m[11] * v.w,
            a.m[12] * v.x + a.m[13] * v.y + a.m[14] * v.z + a.m[15] * v.w};
}

struct Aabb {
    V4 lo, hi;
};
constexpr double kInf = std::numeric_limits<double>::infinity();
inline Aabb aabb_empty() { return {{kInf, kInf, kInf, 1.0}, {-kInf, -kInf, -kInf, -1.0}}; }

// Host-side scene-construction RNG.  The reference draws these values from an
// entropy-seeded Pcg64Mcg (bvh.rs:33, golden_monkey.rs:83); we use the repo's
// SplitMix64 stream (see DESIGN.md "RNG") keyed by --seed so scenes are reproducible.
struct SceneRng {
    uint64_t s;
    explicit SceneRng(uint64_t seed) : s(mix(seed ^ 0x5CE9E5EEDull)) {}
    static uint64_t mix(uint64_t z) {
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    uint64_t next() {
        s += 0x9E3779B97F4A7C15ull;
        return mix(s);
    }
    double uniform() { return double(next() >> 11) * (1.0 / 9007199254740992.0); }
    double range(double lo, double hi) { return lo + (hi - lo) * uniform(); }
    uint32_t below(uint32_t n) { return uint32_t(((next() >> 32) * uint64_t(n)) >> 32); }
    // StandardNormal stand-in: Box-Muller, cosine branch, two uniforms (as the render RNG, rt_device.h)
    double normal() {
        double u1 = uniform(), u2 = uniform();
        return std::sqrt(-2.0 * std::log(1.0 - u1)) * std::cos(6.283185307179586476925286766559 * u2);
    }
};

}  // namespace rth
