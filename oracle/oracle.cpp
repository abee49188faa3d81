// oracle.cpp — CPU restatement of the reference render path, f64, recursive, virtual
// dispatch: the same algorithmic work per sample as teofum/rust_raytracer.
//
// TEST INFRASTRUCTURE, NOT PRODUCT (see oracle.h).  Every function cites the reference
// file:line it follows (paths relative to the reference checkout).  Build with
// `-ffp-contract=off`: Rust never contracts a*b+c into an FMA.
//
// Deliberate deviations, all at the RNG boundary (third-party crates, entropy seeded,
// "parity unpinned"): draws come from a keyed SplitMix64 stream re-keyed per sample
// (seed, replica, pixel, stratum) instead of one sequential Pcg64Mcg per thread;
// StandardNormal is Box-Muller on two uniforms instead of the ziggurat; U (`Standard`) and
// R (`gen_range(0.0..1.0)`) are the same 53-bit uniform.  sin / cos / ln in the SAMPLING routines
// come from include/rt_detmath.h (pure IEEE arithmetic, ~1 ulp) instead of the platform libm, so
// that the GPU kernels can reproduce the oracle's paths bit for bit (Rust's own f64::sin is
// platform-libm dependent too).  The ORDER and COUNT of draws per
// sample follow the reference.
#include "oracle.h"
#include "../include/rt_detmath.h"

#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr double PI = 3.14159265358979323846264338327950288;  // std::f64::consts::PI
constexpr double F64_EPSILON = 2.220446049250313e-16;         // f64::EPSILON
constexpr double F64_MAX = std::numeric_limits<double>::max();
constexpr double INF = std::numeric_limits<double>::infinity();

// ------------------------------------------------------------------ vec4.rs
struct Vec4 {
    double v[4];
    double operator[](int i) const { return v[i]; }
    double& operator[](int i) { return v[i]; }
    double x() const { return v[0]; }
    double y() const { return v[1]; }
    double z() const { return v[2]; }
};
inline Vec4 vec(double x, double y, double z) { return {{x, y, z, 0.0}}; }     // vec4.rs:19
inline Vec4 point(double x, double y, double z) { return {{x, y, z, 1.0}}; }   // vec4.rs:23
inline Vec4 operator+(Vec4 a, Vec4 b) { return {{a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]}}; }
inline Vec4 operator-(Vec4 a, Vec4 b) { return {{a[0] - b[0], a[1] - b[1], a[2] - b[2], a[3] - b[3]}}; }
inline Vec4 operator*(Vec4 a, Vec4 b) { return {{a[0] * b[0], a[1] * b[1], a[2] * b[2], a[3] * b[3]}}; }
inline Vec4 operator*(Vec4 a, double s) { return {{a[0] * s, a[1] * s, a[2] * s, a[3] * s}}; }
inline Vec4 operator/(Vec4 a, double s) { return {{a[0] / s, a[1] / s, a[2] / s, a[3] / s}}; }
inline Vec4 operator-(Vec4 a) { return {{-a[0], -a[1], -a[2], -a[3]}}; }
inline double length_squared(Vec4 a) { return a[0] * a[0] + a[1] * a[1] + a[2] * a[2]; }   // vec4.rs:105
inline double length(Vec4 a) { return std::sqrt(length_squared(a)); }                      // vec4.rs:101
inline double dot(Vec4 a, Vec4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }      // vec4.rs:109
inline Vec4 cross(Vec4 a, Vec4 b) {                                                        // vec4.rs:113
    return {{a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0], 0.0}};
}
inline Vec4 to_unit(Vec4 a) { return a / length(a); }                                      // vec4.rs:122
inline Vec4 reflect(Vec4 v, Vec4 n) { return v - n * (2.0 * dot(v, n)); }                  // vec4.rs:135
inline Vec4 refract(Vec4 v, Vec4 n, double ior_ratio) {                                    // vec4.rs:140-147
    double cos_theta = std::fmin(1.0, dot(-v, n));
    Vec4 perp = (v + (n * cos_theta)) * ior_ratio;
    Vec4 parallel = n * -std::sqrt(1.0 - length_squared(perp));
    return perp + parallel;
}

// ------------------------------------------------------------------ mat4.rs
struct Mat4 {
    double m[16];
};
inline Vec4 operator*(const Mat4& a, Vec4 r) {  // mat4.rs:342-353
    return {{a.m[0] * r[0] + a.m[1] * r[1] + a.m[2] * r[2] + a.m[3] * r[3],
             a.m[4] * r[0] + a.m[5] * r[1] + a.m[6] * r[2] + a.m[7] * r[3],
             a.m[8] * r[0] + a.m[9] * r[1] + a.m[10] * r[2] + a.m[11] * r[3],
             a.m[12] * r[0] + a.m[13] * r[1] + a.m[14] * r[2] + a.m[15] * r[3]}};
}
inline Mat4 from_columns(Vec4 c0, Vec4 c1, Vec4 c2, Vec4 c3) {  // mat4.rs:37-44
    return {{c0[0], c1[0], c2[0], c3[0], c0[1], c1[1], c2[1], c3[1],
             c0[2], c1[2], c2[2], c3[2], c0[3], c1[3], c2[3], c3[3]}};
}

// ------------------------------------------------------------------ utils.rs
inline Mat4 onb_from_vec(Vec4 w) {  // utils.rs:17-28
    Vec4 a = std::fabs(w.x()) > 0.9 ? vec(0.0, 1.0, 0.0) : vec(1.0, 0.0, 0.0);
    Vec4 v = to_unit(cross(w, a));
    Vec4 u = cross(w, v);
    return from_columns(u, v, w, {{0.0, 0.0, 0.0, 1.0}});
}
inline double powi5(double x) {  // f64::powi(5) = llvm.powi: x^2, x^4, x^4 * x
    double x2 = x * x;
    double x4 = x2 * x2;
    return x4 * x;
}
inline double reflectance(double cos_theta, double ior_ratio) {  // utils.rs:31-36
    double r0 = (1.0 - ior_ratio) / (1.0 + ior_ratio);
    r0 = r0 * r0;
    return r0 + (1.0 - r0) * powi5(1.0 - cos_theta);
}

// ------------------------------------------------------------------ RNG (replaces rand_pcg / rand_distr)
struct Counters {
    uint64_t rays = 0, node_tests = 0, tri_tests = 0, prim_tests = 0;
};
struct Rng {
    uint64_t s = 0;
    Counters cnt;  // the rng object is threaded through every test() like in the reference
    static uint64_t mix(uint64_t z) {
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    void key(uint64_t seed, uint32_t tid, uint64_t pixel, uint32_t stratum) {
        uint64_t k = mix(seed + 0x9E3779B97F4A7C15ull * (uint64_t(tid) + 1));
        k = mix(k ^ (pixel * 0xD1B54A32D192ED03ull + 0x8CB92BA72F3D8DD7ull));
        k = mix(k ^ (uint64_t(stratum) * 0xA0761D6478BD642Full + 0xE7037ED1A0B428DBull));
        s = k;
    }
    uint64_t next() {
        s += 0x9E3779B97F4A7C15ull;
        return mix(s);
    }
    double uniform() { return double(next() >> 11) * (1.0 / 9007199254740992.0); }  // U and R
    double normal() {                                                               // N
        double u1 = uniform();
        double u2 = uniform();
        double r = std::sqrt(-2.0 * det_log(1.0 - u1));
        return r * det_cos(2.0 * PI * u2);
    }
    uint32_t below(uint32_t n) { return uint32_t(((next() >> 32) * uint64_t(n)) >> 32); }  // I(n)
};
inline Vec4 random_in_unit_disk(Rng& rng) {  // vec4.rs:35-40
    double x = rng.normal();
    double y = rng.normal();
    return to_unit({{x, y, 0.0, 0.0}});
}
inline Vec4 random_unit(Rng& rng) {  // vec4.rs:42-48
    double x = rng.normal();
    double y = rng.normal();
    double z = rng.normal();
    return to_unit({{x, y, z, 0.0}});
}
inline Vec4 random_cosine(Rng& rng) {  // vec4.rs:50-61
    double r1 = rng.uniform();
    double r2 = rng.uniform();
    double phi = r1 * 2.0 * PI;
    double sqrt_r2 = std::sqrt(r2);
    double x = det_cos(phi) * sqrt_r2;
    double y = det_sin(phi) * sqrt_r2;
    double z = std::sqrt(1.0 - r2);
    return vec(x, y, z);
}

// ------------------------------------------------------------------ ray.rs, interval.rs, aabb.rs
struct Ray {
    Vec4 origin, dir, inv_dir;
    uint8_t sign[3];
    Ray() = default;
    Ray(Vec4 o, Vec4 d) : origin(o), dir(d) {  // ray.rs:19-33
        inv_dir = vec(1.0 / d[0], 1.0 / d[1], 1.0 / d[2]);
        sign[0] = inv_dir[0] < 0.0 ? 1 : 0;
        sign[1] = inv_dir[1] < 0.0 ? 1 : 0;
        sign[2] = inv_dir[2] < 0.0 ? 1 : 0;
    }
    Vec4 at(double t) const { return origin + (dir * t); }  // ray.rs:35
};
struct Interval {
    double min, max;
};
struct Aabb {
    Vec4 b[2];
};
const Vec4 INFINITY_VEC = {{INF, INF, INF, 1.0}};       // constants.rs:3
const Vec4 EPSILON_VEC = {{0.001, 0.001, 0.001, 0.0}};  // aabb.rs:9

Aabb combine_bounds(const Aabb* boxes, size_t n) {  // aabb.rs:11-27
    Vec4 lo = INFINITY_VEC;
    Vec4 hi = -lo;
    for (size_t k = 0; k < n; k++)
        for (int i = 0; i < 3; i++) {
            if (boxes[k].b[0][i] < lo[i]) lo[i] = boxes[k].b[0][i];
            if (boxes[k].b[1][i] > hi[i]) hi[i] = boxes[k].b[1][i];
        }
    return {{lo - EPSILON_VEC, hi + EPSILON_VEC}};
}
Aabb get_bounding_box(const Vec4* pts, size_t n) {  // aabb.rs:29-45
    Vec4 lo = INFINITY_VEC;
    Vec4 hi = -lo;
    for (size_t k = 0; k < n; k++)
        for (int i = 0; i < 3; i++) {
            if (pts[k][i] < lo[i]) lo[i] = pts[k][i];
            if (pts[k][i] > hi[i]) hi[i] = pts[k][i];
        }
    return {{lo - EPSILON_VEC, hi + EPSILON_VEC}};
}
inline bool test_bounding_box(const Aabb& bounds, const Ray& ray, const Interval& t_int) {  // aabb.rs:50-87
    const Vec4& inv_dir = ray.inv_dir;
    const uint8_t* sign = ray.sign;
    double t_min = (bounds.b[sign[0]][0] - ray.origin[0]) * inv_dir[0];
    double t_max = (bounds.b[1 - sign[0]][0] - ray.origin[0]) * inv_dir[0];
    double ty_min = (bounds.b[sign[1]][1] - ray.origin[1]) * inv_dir[1];
    double ty_max = (bounds.b[1 - sign[1]][1] - ray.origin[1]) * inv_dir[1];
    if ((t_min > ty_max) || (ty_min > t_max)) return false;
    if (ty_min > t_min) t_min = ty_min;
    if (ty_max < t_max) t_max = ty_max;
    double tz_min = (bounds.b[sign[2]][2] - ray.origin[2]) * inv_dir[2];
    double tz_max = (bounds.b[1 - sign[2]][2] - ray.origin[2]) * inv_dir[2];
    if ((t_min > tz_max) || (tz_min > t_max)) return false;
    if (tz_min > t_min) t_min = tz_min;
    if (tz_max < t_max) t_max = tz_max;
    return t_min < t_int.max && t_max > t_int.min;
}

// ------------------------------------------------------------------ texture/*.rs
template <typename T>
struct Sampler {
    virtual ~Sampler() = default;
    virtual T sample(double u, double v, const Vec4& p) const = 0;  // texture.rs:21-25
};
template <typename T>
struct ConstantTexture : Sampler<T> {  // constant.rs:30
    T value;
    explicit ConstantTexture(T v) : value(v) {}
    T sample(double, double, const Vec4&) const override { return value; }
};
inline uint32_t f64_as_u32(double x) {  // Rust `as u32`: saturating, NaN -> 0
    if (!(x > 0.0)) return 0;
    if (x >= 4294967295.0) return 4294967295u;
    return uint32_t(x);
}
inline int32_t f64_as_i32(double x) {
    if (x != x) return 0;
    if (x <= -2147483648.0) return INT32_MIN;
    if (x >= 2147483647.0) return INT32_MAX;
    return int32_t(x);
}
template <typename T>
struct CheckerboardTexture : Sampler<T> {  // checkerboard.rs:34-44
    std::shared_ptr<Sampler<T>> even, odd;
    double scale;
    T sample(double u, double v, const Vec4& p) const override {
        uint32_t iu = f64_as_u32(u * 2.0 / scale);
        uint32_t iv = f64_as_u32(v * 2.0 / scale);
        // `iu + iv` overflows (panic in debug, wraps in release); wrapping keeps parity mod 2
        bool is_even = ((iu + iv) % 2u) == 0u;
        return is_even ? even->sample(u, v, p) : odd->sample(u, v, p);
    }
};
template <typename T>
struct CheckerboardSolidTexture : Sampler<T> {  // checkerboard.rs:74-85
    std::shared_ptr<Sampler<T>> even, odd;
    double scale;
    T sample(double u, double v, const Vec4& p) const override {
        int32_t ix = f64_as_i32(std::floor(p.x() / scale));
        int32_t iy = f64_as_i32(std::floor(p.y() / scale));
        int32_t iz = f64_as_i32(std::floor(p.z() / scale));
        int32_t s = int32_t(uint32_t(ix) + uint32_t(iy) + uint32_t(iz));
        bool is_even = (s % 2) == 0;
        return is_even ? even->sample(u, v, p) : odd->sample(u, v, p);
    }
};
template <typename T>
struct Interpolate : Sampler<T> {  // interpolate.rs:29-39
    std::shared_ptr<Sampler<T>> start, end;
    std::shared_ptr<Sampler<double>> t;
    T sample(double u, double v, const Vec4& p) const override {
        double tt = t->sample(u, v, p);
        if (tt == 0.0) return start->sample(u, v, p);
        if (tt == 1.0) return end->sample(u, v, p);
        return start->sample(u, v, p) * (1.0 - tt) + end->sample(u, v, p) * tt;
    }
};
struct Channel : Sampler<double> {  // channel.rs:22-25
    std::shared_ptr<Sampler<Vec4>> color;
    uint32_t channel;
    double sample(double u, double v, const Vec4& p) const override { return color->sample(u, v, p)[int(channel)]; }
};
inline size_t f64_as_usize(double x) {  // Rust `as usize`: saturating, NaN -> 0
    if (!(x > 0.0)) return 0;
    if (x >= 18446744073709551615.0) return SIZE_MAX;
    return size_t(x);
}
struct ImageTexture : Sampler<Vec4> {  // image.rs:37-53 (TextureRepeat::Repeat, nearest neighbour)
    const float* texels = nullptr;     // Buffer::from_image (buffer.rs:30-48): f32 rgb widened to f64
    uint32_t width = 0, height = 0;
    Vec4 sample(double u, double v, const Vec4&) const override {
        u = u - std::floor(u);
        v = v - std::floor(v);
        double w = double(width) - 0.001, h = double(height) - 0.001;
        size_t x = f64_as_usize(u * w), y = f64_as_usize(v * h);
        const float* t = texels + (y * size_t(width) + x) * 3;  // Buffer::get_pixel (buffer.rs:54-56)
        return vec(double(t[0]), double(t[1]), double(t[2]));
    }
};
struct PerlinNoise3D {  // noise/perlin.rs
    const double* random_vec = nullptr;  // 256 x (x, y, z)
    const uint32_t* perm = nullptr;      // perm_x, perm_y, perm_z
    static double smooth(double x) { return x * x * (3.0 - 2.0 * x); }
    double sample(const Vec4& p) const {  // perlin.rs:81-101 + trilinear_interpolation :57-77
        double u = p[0] - std::floor(p[0]), v = p[1] - std::floor(p[1]), w = p[2] - std::floor(p[2]);
        int32_t i = f64_as_i32(std::floor(p[0])), j = f64_as_i32(std::floor(p[1])), k = f64_as_i32(std::floor(p[2]));
        double uu = smooth(u), vv = smooth(v), ww = smooth(w);
        double acc = 0.0;
        for (int di = 0; di < 2; di++)
            for (int dj = 0; dj < 2; dj++)
                for (int dk = 0; dk < 2; dk++) {
                    uint32_t idx = perm[uint32_t(int32_t(uint32_t(di) + uint32_t(i))) & 255u] ^
                                   perm[256 + (uint32_t(int32_t(uint32_t(dj) + uint32_t(j))) & 255u)] ^
                                   perm[512 + (uint32_t(int32_t(uint32_t(dk) + uint32_t(k))) & 255u)];
                    Vec4 c = vec(random_vec[3 * idx], random_vec[3 * idx + 1], random_vec[3 * idx + 2]);
                    double fi = double(di), fj = double(dj), fk = double(dk);
                    Vec4 v_weight = vec(u - fi, v - fj, w - fk);
                    acc += (fi * uu + (1.0 - fi) * (1.0 - uu)) * (fj * vv + (1.0 - fj) * (1.0 - vv)) *
                           (fk * ww + (1.0 - fk) * (1.0 - ww)) * dot(c, v_weight);
                }
        return acc;
    }
    double sample_turbulence(const Vec4& p0, size_t samples) const {  // perlin.rs:103-113
        double acc = 0.0, weight = 1.0;
        Vec4 p = p0;
        for (size_t s = 0; s < samples; s++) {
            acc += weight * sample(p);
            weight *= 0.5;
            p = p * 2.0;
        }
        return std::fabs(acc);
    }
};
struct NoiseSolidTexture : Sampler<double> {  // noise.rs:10-38
    PerlinNoise3D noise;
    Vec4 scale = vec(1.0, 1.0, 1.0);
    size_t samples = 7;
    double sample(double, double, const Vec4& p) const override {
        Vec4 p_scaled = p * scale;
        double sampled = noise.sample_turbulence(p_scaled, samples);
        return 0.5 * (1.0 + det_sin(p_scaled.z() + 10.0 * sampled));  // noise.rs:28 (f64::sin -> det_sin)
    }
};
struct UvDebugTexture : Sampler<Vec4> {  // uv_debug.rs:11-13
    Vec4 sample(double u, double v, const Vec4&) const override { return vec(u, v, 0.5); }
};

// ------------------------------------------------------------------ object.rs
struct Material;
struct HitRecord {  // object.rs:32-41
    Vec4 hit_pos, normal, tangent, bitangent;
    double t;
    double u, v;
    bool front_face;
    const Material* material;
    int material_index;
};
inline HitRecord make_hit(const Ray& ray, Vec4 hit_pos, double t, double u, double v, Vec4 outward_normal,
                          Vec4 tangent, Vec4 bitangent, const Material* material, int material_index) {  // object.rs:45-72
    bool front_face = dot(ray.dir, outward_normal) < 0.0;
    Vec4 normal = front_face ? outward_normal : -outward_normal;
    return {hit_pos, normal, tangent, bitangent, t, u, v, front_face, material, material_index};
}
struct Hit {  // object.rs:107-115
    virtual ~Hit() = default;
    virtual bool test(const Ray& ray, Interval t, Rng& rng, HitRecord& out) const = 0;
    virtual Aabb get_bounding_box() const = 0;
    virtual double pdf_value(Vec4 origin, Vec4 dir, Rng& rng) const = 0;
    virtual Vec4 random(Vec4 origin, Rng& rng) const = 0;
};

// ------------------------------------------------------------------ pdf/*.rs
struct PDF {  // pdf.rs:14-18
    virtual ~PDF() = default;
    virtual double value(const Vec4& dir, Rng& rng) const = 0;
    virtual Vec4 generate(Rng& rng) const = 0;
};
struct CosinePDF : PDF {  // cosine.rs:17-33
    Mat4 basis;
    Vec4 w;
    explicit CosinePDF(Vec4 w_) : basis(onb_from_vec(w_)), w(w_) {}
    double value(const Vec4& dir, Rng&) const override {
        double cos_theta = dot(to_unit(dir), w);
        return std::fmax(cos_theta / PI, 0.0);
    }
    Vec4 generate(Rng& rng) const override { return basis * random_cosine(rng); }
};
struct UniformPDF : PDF {  // uniform.rs:18-24
    double value(const Vec4&, Rng&) const override { return 1.0 / (4.0 * PI); }
    Vec4 generate(Rng& rng) const override { return random_unit(rng); }
};
struct HittablePDF : PDF {  // hittable.rs:22-28
    const Hit* object;
    Vec4 origin;
    double value(const Vec4& dir, Rng& rng) const override { return object->pdf_value(origin, dir, rng); }
    Vec4 generate(Rng& rng) const override { return object->random(origin, rng); }
};
struct MixPDF : PDF {  // mix.rs:23-36
    const PDF* first;
    const PDF* second;
    double mix;
    double value(const Vec4& dir, Rng& rng) const override {
        double first_val = first->value(dir, rng);
        double second_val = second->value(dir, rng);
        return first_val * (1.0 - mix) + second_val * mix;
    }
    Vec4 generate(Rng& rng) const override {
        if (rng.uniform() < mix) return second->generate(rng);
        return first->generate(rng);
    }
};

// ------------------------------------------------------------------ material/*.rs
enum class ScatterKind { WithPDF, WithRay, Absorbed, Emissive };  // material.rs:25-36
struct ScatterResult {
    ScatterKind kind;
    Vec4 attenuation;
    std::unique_ptr<PDF> pdf;  // Box<dyn PDF>: one heap allocation per diffuse hit, as in the reference
    Ray scattered;
};
struct Material {  // material.rs:38-47
    virtual ~Material() = default;
    virtual ScatterResult scatter(const Ray& ray, const HitRecord& hit, Rng& rng) const = 0;
    virtual Vec4 emit(const HitRecord&) const { return vec(0.0, 0.0, 0.0); }
    virtual double scattering_pdf(const Ray& ray_in, const Ray& scattered, const HitRecord& hit) const = 0;
};
using ColorTex = std::shared_ptr<Sampler<Vec4>>;
using FloatTex = std::shared_ptr<Sampler<double>>;

struct LambertianDiffuse : Material {  // lambertian.rs:25-43
    ColorTex albedo;
    ScatterResult scatter(const Ray&, const HitRecord& hit, Rng&) const override {
        ScatterResult r{ScatterKind::WithPDF, albedo->sample(hit.u, hit.v, hit.hit_pos), nullptr, {}};
        r.pdf = std::make_unique<CosinePDF>(hit.normal);
        return r;
    }
    double scattering_pdf(const Ray&, const Ray& scattered, const HitRecord& hit) const override {
        double cos_theta = dot(hit.normal, to_unit(scattered.dir));
        return cos_theta < 0.0 ? 0.0 : cos_theta / PI;
    }
};
struct Metal : Material {  // metal.rs:28-48
    ColorTex albedo;
    FloatTex roughness;
    ScatterResult scatter(const Ray& ray, const HitRecord& hit, Rng& rng) const override {
        Vec4 reflected = reflect(ray.dir, hit.normal);
        Vec4 scatter_dir = reflected + random_unit(rng) * roughness->sample(hit.u, hit.v, hit.hit_pos) * length(reflected);
        if (dot(scatter_dir, hit.normal) > 0.0) {
            return {ScatterKind::WithRay, albedo->sample(hit.u, hit.v, hit.hit_pos), nullptr, Ray(hit.hit_pos, scatter_dir)};
        }
        return {ScatterKind::Absorbed, vec(0, 0, 0), nullptr, {}};
    }
    double scattering_pdf(const Ray&, const Ray&, const HitRecord&) const override { return 1.0; }
};
struct Dielectric : Material {  // dielectric.rs:29-58
    double ior;
    ScatterResult scatter(const Ray& ray, const HitRecord& hit, Rng& rng) const override {
        double ior_ratio = hit.front_face ? 1.0 / ior : ior;
        Vec4 unit_dir = to_unit(ray.dir);
        double cos_theta = std::fmin(1.0, dot(-unit_dir, hit.normal));
        double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
        bool tir = ior_ratio * sin_theta > 1.0;
        // `tir || reflectance(..) > rng.gen_range(0.0..1.0)`: no draw on total internal reflection
        bool reflected = tir || reflectance(cos_theta, ior_ratio) > rng.uniform();
        Vec4 scatter_dir = reflected ? reflect(unit_dir, hit.normal) : refract(unit_dir, hit.normal, ior_ratio);
        return {ScatterKind::WithRay, vec(1.0, 1.0, 1.0), nullptr, Ray(hit.hit_pos, scatter_dir)};
    }
    double scattering_pdf(const Ray&, const Ray&, const HitRecord&) const override { return 1.0; }
};
inline Vec4 mapped_normal(const ColorTex& normal_map, const HitRecord& hit) {  // glossy.rs:35-50, normal_debug.rs:23-39
    if (normal_map) {
        Vec4 sampled = normal_map->sample(hit.u, hit.v, hit.hit_pos);
        Mat4 basis = from_columns(hit.tangent, hit.bitangent, hit.normal, {{0.0, 0.0, 0.0, 1.0}});
        return to_unit(basis * (sampled - vec(0.5, 0.5, 0.5)));
    }
    return hit.normal;
}
struct Glossy : Material {  // glossy.rs:54-95
    ColorTex albedo;
    FloatTex roughness;
    ColorTex normal_map;
    double inv_ior;
    ScatterResult scatter(const Ray& ray, const HitRecord& hit, Rng& rng) const override {
        Vec4 normal = mapped_normal(normal_map, hit);
        Vec4 unit_dir = to_unit(ray.dir);
        double cos_theta = std::fmin(1.0, dot(-unit_dir, normal));
        bool specular = reflectance(cos_theta, inv_ior) > rng.uniform();
        if (specular) {
            double rough = roughness->sample(hit.u, hit.v, hit.hit_pos);
            Vec4 reflected = reflect(ray.dir, normal);
            Vec4 scatter_dir = reflected + random_unit(rng) * rough * length(reflected);
            if (dot(scatter_dir, normal) > 0.0)
                return {ScatterKind::WithRay, vec(1.0, 1.0, 1.0), nullptr, Ray(hit.hit_pos, scatter_dir)};
            return {ScatterKind::Absorbed, vec(0, 0, 0), nullptr, {}};
        }
        ScatterResult r{ScatterKind::WithPDF, albedo->sample(hit.u, hit.v, hit.hit_pos), nullptr, {}};
        r.pdf = std::make_unique<CosinePDF>(normal);
        return r;
    }
    double scattering_pdf(const Ray&, const Ray& scattered, const HitRecord& hit) const override {
        Vec4 normal = mapped_normal(normal_map, hit);
        double cos_theta = dot(normal, to_unit(scattered.dir));
        return cos_theta < 0.0 ? 0.0 : cos_theta / PI;
    }
};
struct Emissive : Material {  // emissive.rs:24-38
    ColorTex emission_map;
    ScatterResult scatter(const Ray&, const HitRecord&, Rng&) const override {
        return {ScatterKind::Emissive, vec(0, 0, 0), nullptr, {}};
    }
    Vec4 emit(const HitRecord& hit) const override {
        return hit.front_face ? emission_map->sample(hit.u, hit.v, hit.hit_pos) : vec(0.0, 0.0, 0.0);
    }
    double scattering_pdf(const Ray&, const Ray&, const HitRecord&) const override { return 1.0; }
};
struct Isotropic : Material {  // isotropic.rs:25-37
    ColorTex albedo;
    ScatterResult scatter(const Ray&, const HitRecord& hit, Rng&) const override {
        ScatterResult r{ScatterKind::WithPDF, albedo->sample(hit.u, hit.v, hit.hit_pos), nullptr, {}};
        r.pdf = std::make_unique<UniformPDF>();
        return r;
    }
    double scattering_pdf(const Ray&, const Ray&, const HitRecord&) const override { return 1.0 / (4.0 * PI); }
};
struct NormalDebug : Material {  // normal_debug.rs:42-52
    ColorTex normal_map;
    ScatterResult scatter(const Ray&, const HitRecord&, Rng&) const override {
        return {ScatterKind::Emissive, vec(0, 0, 0), nullptr, {}};
    }
    Vec4 emit(const HitRecord& hit) const override { return mapped_normal(normal_map, hit) * 0.5 + vec(0.5, 0.5, 0.5); }
    double scattering_pdf(const Ray&, const Ray&, const HitRecord&) const override { return 1.0; }
};

// ------------------------------------------------------------------ object/sphere.rs
Vec4 random_to_sphere(double radius, double distance_squared, Rng& rng) {  // sphere.rs:131-145
    double radius_squared = radius * radius;
    double cos_theta_max = std::sqrt(1.0 - radius_squared / distance_squared);
    double r1 = rng.uniform();
    double r2 = rng.uniform();
    double phi = r1 * 2.0 * PI;
    double z = 1.0 + r2 * (cos_theta_max - 1.0);
    double x = det_cos(phi) * std::sqrt(1.0 - z * z);
    double y = det_sin(phi) * std::sqrt(1.0 - z * z);
    return vec(x, y, z);
}
struct Sphere : Hit {
    const Material* material;
    int material_index;
    Vec4 center;
    double radius;
    Aabb bounds;
    Sphere(Vec4 c, double r, const Material* m, int mi) : material(m), material_index(mi), center(c), radius(r) {  // sphere.rs:27-37
        Vec4 rv = vec(r, r, r);
        bounds = {{c - rv, c + rv}};
    }
    bool test_impl(const Ray& ray, Interval t, bool skip_uvs, Rng& rng, HitRecord& out) const {  // sphere.rs:40-94
        rng.cnt.prim_tests++;
        Vec4 center_diff = ray.origin - center;
        double a = length_squared(ray.dir);
        double half_b = dot(ray.dir, center_diff);
        double c = length_squared(center_diff) - radius * radius;
        double discriminant = half_b * half_b - a * c;
        if (discriminant < 0.0) return false;
        double d_sqrt = std::sqrt(discriminant);
        double root = (-half_b - d_sqrt) / a;
        if (root <= t.min || t.max <= root) {
            root = (-half_b + d_sqrt) / a;
            if (root <= t.min || t.max <= root) return false;
        }
        Vec4 hit_pos = ray.at(root);
        Vec4 normal = (hit_pos - center) / radius;
        double u = 0.0, v = 0.0;
        Vec4 tangent = vec(1.0, 0.0, 0.0), bitangent = vec(1.0, 0.0, 0.0);
        if (!skip_uvs) {
            double theta = det_acos(normal[1]);  // f64::acos / atan2 (sphere.rs:69-70) through the shared deterministic functions
            double phi = det_atan2(-normal[2], normal[0]) + PI;
            tangent = vec(-normal[2], 0.0, -normal[0]);
            bitangent = cross(normal, tangent);
            u = phi / (2.0 * PI);
            v = theta / PI;
        }
        out = make_hit(ray, hit_pos, root, u, v, normal, tangent, bitangent, material, material_index);
        return true;
    }
    bool test(const Ray& ray, Interval t, Rng& rng, HitRecord& out) const override { return test_impl(ray, t, false, rng, out); }
    Aabb get_bounding_box() const override { return bounds; }
    double pdf_value(Vec4 origin, Vec4 dir, Rng& rng) const override {  // sphere.rs:106-121
        HitRecord h;
        if (test_impl(Ray(origin, dir), {0.001, INF}, true, rng, h)) {
            double radius_squared = radius * radius;
            double cos_theta_max = std::sqrt(1.0 - radius_squared / length_squared(center - origin));
            double solid_angle = 2.0 * PI * (1.0 - cos_theta_max);
            return 1.0 / solid_angle;
        }
        return 0.0;
    }
    Vec4 random(Vec4 origin, Rng& rng) const override {  // sphere.rs:123-128
        Vec4 dir = center - origin;
        Mat4 basis = onb_from_vec(dir);
        return basis * random_to_sphere(radius, length_squared(dir), rng);
    }
};

// ------------------------------------------------------------------ object/plane.rs
struct Plane : Hit {
    const Material* material;
    int material_index;
    bool render_backface = false;
    Vec4 corner, normal, u, v, inv_u, inv_v;
    double area;
    Aabb bounds;
    Plane(Vec4 center, Vec4 u_, Vec4 v_, const Material* m, int mi) : material(m), material_index(mi), u(u_), v(v_) {  // plane.rs:29-63
        Vec4 u_unit = to_unit(u);
        Vec4 v_unit = to_unit(v);
        Vec4 n = cross(u, v);
        area = length(n) * 4.0;
        normal = to_unit(n);
        Vec4 corners[4] = {center + u + v, center + u - v, center - u + v, center - u - v};
        bounds = ::get_bounding_box(corners, 4);
        corner = corners[3];
        inv_u = u_unit * 0.5 / length(u);
        inv_v = v_unit * 0.5 / length(v);
    }
    bool test(const Ray& ray, Interval t, Rng& rng, HitRecord& out) const override {  // plane.rs:66-101
        rng.cnt.prim_tests++;
        double dot_ray_normal = dot(normal, ray.dir);
        double dd = render_backface ? std::fabs(dot_ray_normal) : -dot_ray_normal;
        if (dd < F64_EPSILON) return false;
        double hit_t = dot(normal, corner - ray.origin) / dot_ray_normal;
        if (hit_t <= t.min || t.max <= hit_t) return false;
        Vec4 hit_pos = ray.at(hit_t);
        Vec4 local_pos = hit_pos - corner;
        double uu = dot(local_pos, inv_u);
        double vv = dot(local_pos, inv_v);
        if (uu < 0.0 || uu > 1.0 || vv < 0.0 || vv > 1.0) return false;
        out = make_hit(ray, hit_pos, hit_t, uu, vv, normal, to_unit(u), to_unit(v), material, material_index);
        return true;
    }
    Aabb get_bounding_box() const override { return bounds; }
    double pdf_value(Vec4 origin, Vec4 dir, Rng& rng) const override {  // plane.rs:107-118
        Ray ray(origin, dir);
        HitRecord hit;
        if (test(ray, {0.001, INF}, rng, hit)) {
            double dist_squared = hit.t * hit.t * length_squared(dir);
            double cosine = std::fabs(dot(dir, hit.normal) / length(dir));
            return dist_squared / (cosine * area);
        }
        return 0.0;
    }
    Vec4 random(Vec4 origin, Rng& rng) const override {  // plane.rs:120-126 (one quarter of the quad, SURVEY B-1)
        double ru = rng.uniform();
        double rv = rng.uniform();
        Vec4 p = corner + u * ru + v * rv;
        return p - origin;
    }
};

// ------------------------------------------------------------------ object/list.rs
struct ObjectList : Hit {
    std::vector<const Hit*> objects;
    Aabb bounds = {{INFINITY_VEC, -INFINITY_VEC}};  // list.rs:30
    bool disable_bounds_check = false;
    void add(const Hit* object) {  // list.rs:51-54
        Aabb two[2] = {bounds, object->get_bounding_box()};
        bounds = combine_bounds(two, 2);
        objects.push_back(object);
    }
    bool test(const Ray& ray, Interval t, Rng& rng, HitRecord& out) const override {  // list.rs:58-74
        if (!disable_bounds_check && !test_bounding_box(bounds, ray, t)) return false;
        bool any = false;
        double closest_t = t.max;
        HitRecord h;
        for (const Hit* object : objects) {
            if (object->test(ray, {t.min, closest_t}, rng, h)) {
                closest_t = h.t;
                out = h;
                any = true;
            }
        }
        return any;
    }
    Aabb get_bounding_box() const override { return bounds; }
    double pdf_value(Vec4 origin, Vec4 dir, Rng& rng) const override {  // list.rs:80-89
        double weight = 1.0 / double(objects.size());
        double sum = 0.0;
        for (const Hit* obj : objects) sum += weight * obj->pdf_value(origin, dir, rng);
        return sum;
    }
    Vec4 random(Vec4 origin, Rng& rng) const override {  // list.rs:91-100
        size_t size = objects.size();
        if (size == 0) return vec(1.0, 0.0, 0.0);
        uint32_t idx = rng.below(uint32_t(size));
        return objects[idx]->random(origin, rng);
    }
};

// ------------------------------------------------------------------ object/mesh.rs + mesh/octree.rs
struct Triangle {  // mesh.rs:15-20
    size_t vert_indices[3];
    size_t normal_indices[3];
    bool has_uv;
    size_t uv_indices[3];
};
constexpr size_t MAX_TRIS_PER_LEAF = 50;  // octree.rs:6
constexpr size_t MAX_DEPTH = 50;          // octree.rs:7
struct OctreeNode {  // octree.rs:9-19
    bool is_leaf;
    std::vector<size_t> leaf;
    std::unique_ptr<OctreeNode> children[8];
    Aabb bounding_box;
};
std::unique_ptr<OctreeNode> octree_new(const std::vector<Vec4>& vertices, const std::vector<Triangle>& triangles,
                                       const std::vector<size_t>* filter, Aabb box, size_t depth) {  // octree.rs:31-210
    std::vector<size_t> all;
    if (!filter) {
        all.resize(triangles.size());
        for (size_t i = 0; i < all.size(); i++) all[i] = i;
        filter = &all;
    }
    auto node = std::make_unique<OctreeNode>();
    node->bounding_box = box;
    const Vec4 b_min = box.b[0], b_max = box.b[1];
    if (filter->size() <= MAX_TRIS_PER_LEAF || depth >= MAX_DEPTH) {
        node->is_leaf = true;
        node->leaf = *filter;
        return node;
    }
    node->is_leaf = false;
    Vec4 midpoint = (b_min + b_max) / 2.0;
    std::vector<size_t> index_lists[8];
    for (size_t index : *filter) {
        const Triangle& tri = triangles[index];
        Vec4 tv[3] = {vertices[tri.vert_indices[0]], vertices[tri.vert_indices[1]], vertices[tri.vert_indices[2]]};
        Aabb tb = get_bounding_box(tv, 3);
        const Vec4& t_min = tb.b[0];
        const Vec4& t_max = tb.b[1];
        bool in_lists[8] = {true, true, true, true, true, true, true, true};
        if (t_min.x() > midpoint.x()) in_lists[0] = in_lists[1] = in_lists[2] = in_lists[3] = false;
        if (t_max.x() < midpoint.x()) in_lists[4] = in_lists[5] = in_lists[6] = in_lists[7] = false;
        if (t_min.y() > midpoint.y()) in_lists[0] = in_lists[1] = in_lists[4] = in_lists[5] = false;
        if (t_max.y() < midpoint.y()) in_lists[2] = in_lists[3] = in_lists[6] = in_lists[7] = false;
        if (t_min.z() > midpoint.z()) in_lists[0] = in_lists[2] = in_lists[4] = in_lists[6] = false;
        if (t_max.z() < midpoint.z()) in_lists[1] = in_lists[3] = in_lists[5] = in_lists[7] = false;
        for (int i = 0; i < 8; i++)
            if (in_lists[i]) index_lists[i].push_back(index);
    }
    double min_x = b_min.x(), min_y = b_min.y(), min_z = b_min.z();
    double max_x = b_max.x(), max_y = b_max.y(), max_z = b_max.z();
    double mid_x = midpoint.x(), mid_y = midpoint.y(), mid_z = midpoint.z();
    Aabb boxes[8] = {
        {{b_min, midpoint}},
        {{point(min_x, min_y, mid_z), point(mid_x, mid_y, max_z)}},
        {{point(min_x, mid_y, min_z), point(mid_x, max_y, mid_z)}},
        {{point(min_x, mid_y, mid_z), point(mid_x, max_y, max_z)}},
        {{point(mid_x, min_y, min_z), point(max_x, mid_y, mid_z)}},
        {{point(mid_x, min_y, mid_z), point(max_x, mid_y, max_z)}},
        {{point(mid_x, mid_y, min_z), point(max_x, max_y, mid_z)}},
        {{midpoint, b_max}},
    };
    for (int i = 0; i < 8; i++) node->children[i] = octree_new(vertices, triangles, &index_lists[i], boxes[i], depth + 1);
    return node;
}

struct TriangleMesh : Hit {
    const Material* material;
    int material_index;
    bool flat_shading = false, hit_back_faces = false;
    std::vector<Vec4> vertices, vertex_uvs, vertex_normals;
    std::vector<Triangle> triangles;
    Aabb bounds;
    std::unique_ptr<OctreeNode> octree;
    void finish() {  // mesh.rs:38-59
        bounds = ::get_bounding_box(vertices.data(), vertices.size());
        octree = octree_new(vertices, triangles, nullptr, bounds, 0);
    }
    bool test_tri(const Triangle& triangle, const Ray& ray, Interval t_int, Rng& rng, HitRecord& out) const {  // mesh.rs:62-163
        rng.cnt.tri_tests++;
        Vec4 v0 = vertices[triangle.vert_indices[0]];
        Vec4 v1 = vertices[triangle.vert_indices[1]];
        Vec4 v2 = vertices[triangle.vert_indices[2]];
        Vec4 edge1 = v1 - v0;
        Vec4 edge2 = v2 - v0;
        Vec4 ray_x_edge2 = cross(ray.dir, edge2);
        double det = dot(edge1, ray_x_edge2);
        double dd = hit_back_faces ? std::fabs(det) : det;
        if (dd < F64_EPSILON) return false;
        double inv_det = 1.0 / det;
        Vec4 b = ray.origin - v0;
        double u = dot(b, ray_x_edge2) * inv_det;
        if (u < 0.0 || u > 1.0) return false;
        Vec4 b_x_edge1 = cross(b, edge1);
        double v = dot(ray.dir, b_x_edge1) * inv_det;
        if (v < 0.0 || u + v > 1.0) return false;
        double t = dot(edge2, b_x_edge1) * inv_det;
        if (t <= t_int.min || t_int.max <= t) return false;
        Vec4 hit_pos = ray.at(t);
        double w = 1.0 - u - v;
        Vec4 normal;
        if (flat_shading) {
            normal = to_unit(cross(edge1, edge2));
        } else {
            Vec4 n0 = vertex_normals[triangle.normal_indices[0]];
            Vec4 n1 = vertex_normals[triangle.normal_indices[1]];
            Vec4 n2 = vertex_normals[triangle.normal_indices[2]];
            normal = n0 * w + n1 * u + n2 * v;  // not normalised (SURVEY B-4)
        }
        Vec4 tangent = vec(1.0, 0.0, 0.0), bitangent = vec(1.0, 0.0, 0.0);
        Vec4 tex = vec(0.0, 0.0, 0.0);
        if (triangle.has_uv) {
            Vec4 uv0 = vertex_uvs[triangle.uv_indices[0]];
            Vec4 uv1 = vertex_uvs[triangle.uv_indices[1]];
            Vec4 uv2 = vertex_uvs[triangle.uv_indices[2]];
            Vec4 duv1 = uv1 - uv0;
            Vec4 duv2 = uv2 - uv0;
            Vec4 edge1perp = cross(normal, edge1);
            Vec4 edge2perp = cross(edge2, normal);
            tangent = edge2perp * duv1[0] + edge1perp * duv2[0];
            bitangent = edge2perp * duv1[1] + edge1perp * duv2[1];
            double inv_max = 1.0 / std::sqrt(std::fmax(length_squared(tangent), length_squared(bitangent)));
            tangent = tangent * -inv_max;
            bitangent = bitangent * inv_max;
            tex = uv0 * w + uv1 * u + uv2 * v;
        }
        out = make_hit(ray, hit_pos, t, tex[0], tex[1], normal, tangent, bitangent, material, material_index);
        return true;
    }
    bool test_octree_node(const OctreeNode& node, const Ray& ray, Interval t, Rng& rng, HitRecord& out) const {  // mesh.rs:165-197
        rng.cnt.node_tests++;
        if (!test_bounding_box(node.bounding_box, ray, t)) return false;
        bool any = false;
        double closest_t = t.max;
        HitRecord h;
        if (node.is_leaf) {
            for (size_t idx : node.leaf) {
                if (test_tri(triangles[idx], ray, {t.min, closest_t}, rng, h)) {
                    closest_t = h.t;
                    out = h;
                    any = true;
                }
            }
        } else {
            for (int i = 0; i < 8; i++) {
                if (test_octree_node(*node.children[i], ray, {t.min, closest_t}, rng, h)) {
                    closest_t = h.t;
                    out = h;
                    any = true;
                }
            }
        }
        return any;
    }
    bool test(const Ray& ray, Interval t, Rng& rng, HitRecord& out) const override { return test_octree_node(*octree, ray, t, rng, out); }  // mesh.rs:201
    Aabb get_bounding_box() const override { return bounds; }
    double pdf_value(Vec4, Vec4, Rng&) const override { return 0.0; }
    Vec4 random(Vec4, Rng&) const override { return vec(1.0, 0.0, 0.0); }
};

// ------------------------------------------------------------------ object/transform.rs
struct Transform : Hit {
    const Hit* object;
    Mat4 transform, inv_transform;
    Aabb bounds;
    bool has_ops;
    Transform(const Hit* obj, const Mat4& m, const Mat4& inv, bool ops) : object(obj), transform(m), inv_transform(inv), has_ops(ops) {
        if (!ops) {
            bounds = obj->get_bounding_box();  // transform.rs:32 (no op applied: bounds copied unpadded)
        } else {
            update_bounds();
        }
    }
    void update_bounds() {  // transform.rs:98-118
        Aabb ob = object->get_bounding_box();
        Vec4 o_min = ob.b[0], o_max = ob.b[1];
        Vec4 d = o_max - o_min;
        double sx = d[0], sy = d[1], sz = d[2];
        Vec4 corners[8] = {o_min,
                           o_min + vec(0.0, 0.0, sz),
                           o_min + vec(0.0, sy, 0.0),
                           o_min + vec(0.0, sy, sz),
                           o_min + vec(sx, 0.0, 0.0),
                           o_min + vec(sx, 0.0, sz),
                           o_min + vec(sx, sy, 0.0),
                           o_max};
        for (auto& c : corners) c = transform * c;  // each corner keeps its own w (o_max has w = -1 for computed boxes)
        bounds = ::get_bounding_box(corners, 8);
    }
    bool test(const Ray& ray, Interval t, Rng& rng, HitRecord& out) const override {  // transform.rs:122-139
        Ray ray_obj(inv_transform * ray.origin, inv_transform * ray.dir);
        if (object->test(ray_obj, t, rng, out)) {
            out.hit_pos = transform * out.hit_pos;
            out.normal = to_unit(transform * out.normal);
            return true;
        }
        return false;
    }
    Aabb get_bounding_box() const override { return bounds; }
    double pdf_value(Vec4, Vec4, Rng&) const override { return 0.0; }
    Vec4 random(Vec4, Rng&) const override { return vec(1.0, 0.0, 0.0); }
};

// ------------------------------------------------------------------ object/bvh.rs, bvh/null_obj.rs
struct NullObject : Hit {
    bool test(const Ray&, Interval, Rng&, HitRecord&) const override { return false; }
    Aabb get_bounding_box() const override { return {{INFINITY_VEC, -INFINITY_VEC}}; }
    double pdf_value(Vec4, Vec4, Rng&) const override { return 0.0; }
    Vec4 random(Vec4, Rng&) const override { return vec(1.0, 0.0, 0.0); }
};
struct BvhNode : Hit {
    const Hit* c0;
    const Hit* c1;
    Aabb bounds;
    BvhNode(const Hit* a, const Hit* b, bool single) : c0(a), c1(b) {  // bvh.rs:48-79
        if (single) {
            bounds = a->get_bounding_box();
        } else {
            Aabb two[2] = {a->get_bounding_box(), b->get_bounding_box()};
            bounds = combine_bounds(two, 2);
        }
    }
    bool test(const Ray& ray, Interval t, Rng& rng, HitRecord& out) const override {  // bvh.rs:84-101
        if (!test_bounding_box(bounds, ray, t)) return false;
        bool any = false;
        double closest_t = t.max;
        HitRecord h;
        if (c0->test(ray, {t.min, closest_t}, rng, h)) {
            closest_t = h.t;
            out = h;
            any = true;
        }
        if (c1->test(ray, {t.min, closest_t}, rng, h)) {
            out = h;
            any = true;
        }
        return any;
    }
    Aabb get_bounding_box() const override { return bounds; }
    double pdf_value(Vec4, Vec4, Rng&) const override { return 0.0; }
    Vec4 random(Vec4, Rng&) const override { return vec(1.0, 0.0, 0.0); }
};

// ------------------------------------------------------------------ object/sky.rs, sun.rs
struct Sky : Hit {
    const Material* material;
    int material_index;
    bool test(const Ray& ray, Interval t, Rng& rng, HitRecord& out) const override {  // sky.rs:28-52
        rng.cnt.prim_tests++;
        double hit_t = INF;
        if (hit_t > t.max) return false;
        Vec4 hit_pos = ray.at(hit_t);
        Vec4 unit_dir = to_unit(ray.dir);
        Vec4 normal = -unit_dir;
        double u = det_atan2(unit_dir.x(), unit_dir.z()) / (2.0 * PI) + 0.5;
        double v = dot(unit_dir, vec(0.0, 1.0, 0.0)) / 2.0 + 0.5;
        out = make_hit(ray, hit_pos, hit_t, u, v, normal, vec(1, 0, 0), vec(1, 0, 0), material, material_index);
        return true;
    }
    Aabb get_bounding_box() const override { return {{point(-F64_MAX, -F64_MAX, -F64_MAX), point(F64_MAX, F64_MAX, F64_MAX)}}; }
    double pdf_value(Vec4, Vec4, Rng&) const override { return 1.0 / (4.0 * PI); }
    Vec4 random(Vec4, Rng& rng) const override { return random_unit(rng); }
};
constexpr double THETA_MAX = 0.001;  // sun.rs:14
struct Sun : Hit {
    Vec4 direction;
    const Material* material;
    int material_index;
    bool test(const Ray& ray, Interval t, Rng& rng, HitRecord& out) const override {  // sun.rs:33-61
        rng.cnt.prim_tests++;
        Vec4 unit_dir = to_unit(ray.dir);
        if (std::fabs(dot(direction, unit_dir) - 1.0) > THETA_MAX) return false;
        double hit_t = F64_MAX;
        if (hit_t >= t.max) return false;
        Vec4 hit_pos = ray.at(hit_t);
        Vec4 normal = -unit_dir;
        out = make_hit(ray, hit_pos, hit_t, 0.0, 0.0, normal, vec(1, 0, 0), vec(1, 0, 0), material, material_index);
        return true;
    }
    Aabb get_bounding_box() const override { return {{point(-F64_MAX, -F64_MAX, -F64_MAX), point(F64_MAX, F64_MAX, F64_MAX)}}; }
    double pdf_value(Vec4, Vec4, Rng&) const override { return 1.0; }
    Vec4 random(Vec4, Rng&) const override { return direction; }
};

// ------------------------------------------------------------------ object/volume.rs
struct Volume : Hit {
    const Hit* boundary;
    const Material* material;
    int material_index;
    double neg_inv_density;
    bool test(const Ray& ray, Interval t, Rng& rng, HitRecord& out) const override {  // volume.rs:33-71
        HitRecord hit_enter, hit_exit;
        if (boundary->test(ray, {-INF, INF}, rng, hit_enter)) {
            if (boundary->test(ray, {hit_enter.t + 0.0001, INF}, rng, hit_exit)) {
                double t_min = std::fmax(hit_enter.t, t.min);
                double t_max = std::fmin(hit_exit.t, t.max);
                if (t_min >= t_max) return false;
                t_min = std::fmax(t_min, 0.0);
                double ray_len = length(ray.dir);
                double dist_inside_boundary = (t_max - t_min) * ray_len;
                double uu = rng.uniform();
                double hit_dist = neg_inv_density * (uu == 0.0 ? -INF : det_log(uu));  // f64::ln(0) = -inf
                if (hit_dist > dist_inside_boundary) return false;
                double tt = t_min + hit_dist / ray_len;
                Vec4 hit_pos = ray.at(tt);
                out = make_hit(ray, hit_pos, tt, 0.0, 0.0, vec(1, 0, 0), vec(1, 0, 0), vec(1, 0, 0), material, material_index);
                return true;
            }
        }
        return false;
    }
    Aabb get_bounding_box() const override { return boundary->get_bounding_box(); }
    double pdf_value(Vec4, Vec4, Rng&) const override { return 0.0; }
    Vec4 random(Vec4, Rng&) const override { return vec(1.0, 0.0, 0.0); }
};

// ------------------------------------------------------------------ scene description -> object graph
thread_local std::string g_err;

struct World {
    std::vector<ColorTex> color_tex;
    std::vector<FloatTex> float_tex;
    std::vector<std::unique_ptr<Material>> materials;
    std::vector<std::unique_ptr<Hit>> nodes;  // one per description node (shared subtrees stay shared)
    std::vector<std::unique_ptr<Hit>> extra;
    const Hit* world = nullptr;
    const Hit* lights = nullptr;
};

struct Builder {
    const RtSceneDesc& d;
    World& w;
    std::vector<int> tex_state;  // 0 = not built, 1 = colour, 2 = float
    bool ok = true;

    bool build_texture(int i) {
        if (i < 0 || uint32_t(i) >= d.n_textures) { g_err = "texture index out of range"; return false; }
        if (tex_state[i]) return true;
        const RtTexture& t = d.textures[i];
        auto kind_of = [&](int k) { return tex_state[k]; };
        switch (t.type) {
            case RT_TEX_CONST_COLOR:
                w.color_tex[i] = std::make_shared<ConstantTexture<Vec4>>(vec(t.v[0], t.v[1], t.v[2]));
                tex_state[i] = 1;
                return true;
            case RT_TEX_CONST_FLOAT:
                w.float_tex[i] = std::make_shared<ConstantTexture<double>>(t.v[0]);
                tex_state[i] = 2;
                return true;
            case RT_TEX_UV_DEBUG:
                w.color_tex[i] = std::make_shared<UvDebugTexture>();
                tex_state[i] = 1;
                return true;
            case RT_TEX_CHECKER:
            case RT_TEX_CHECKER_SOLID: {
                if (!build_texture(t.a) || !build_texture(t.b)) return false;
                if (kind_of(t.a) != kind_of(t.b)) { g_err = "checker inputs differ in type"; return false; }
                bool solid = t.type == RT_TEX_CHECKER_SOLID;
                if (kind_of(t.a) == 1) {
                    if (solid) { auto c = std::make_shared<CheckerboardSolidTexture<Vec4>>(); c->even = w.color_tex[t.a]; c->odd = w.color_tex[t.b]; c->scale = t.scale; w.color_tex[i] = c; }
                    else { auto c = std::make_shared<CheckerboardTexture<Vec4>>(); c->even = w.color_tex[t.a]; c->odd = w.color_tex[t.b]; c->scale = t.scale; w.color_tex[i] = c; }
                    tex_state[i] = 1;
                } else {
                    if (solid) { auto c = std::make_shared<CheckerboardSolidTexture<double>>(); c->even = w.float_tex[t.a]; c->odd = w.float_tex[t.b]; c->scale = t.scale; w.float_tex[i] = c; }
                    else { auto c = std::make_shared<CheckerboardTexture<double>>(); c->even = w.float_tex[t.a]; c->odd = w.float_tex[t.b]; c->scale = t.scale; w.float_tex[i] = c; }
                    tex_state[i] = 2;
                }
                return true;
            }
            case RT_TEX_LERP: {
                if (!build_texture(t.a) || !build_texture(t.b) || !build_texture(t.c)) return false;
                if (kind_of(t.a) != kind_of(t.b) || kind_of(t.c) != 2) { g_err = "lerp inputs have wrong types"; return false; }
                if (kind_of(t.a) == 1) { auto c = std::make_shared<Interpolate<Vec4>>(); c->start = w.color_tex[t.a]; c->end = w.color_tex[t.b]; c->t = w.float_tex[t.c]; w.color_tex[i] = c; tex_state[i] = 1; }
                else { auto c = std::make_shared<Interpolate<double>>(); c->start = w.float_tex[t.a]; c->end = w.float_tex[t.b]; c->t = w.float_tex[t.c]; w.float_tex[i] = c; tex_state[i] = 2; }
                return true;
            }
            case RT_TEX_CHANNEL: {
                if (!build_texture(t.a)) return false;
                if (kind_of(t.a) != 1) { g_err = "channel input must be a colour texture"; return false; }
                auto c = std::make_shared<Channel>(); c->color = w.color_tex[t.a]; c->channel = t.channel; w.float_tex[i] = c; tex_state[i] = 2;
                return true;
            }
            case RT_TEX_IMAGE: {
                if (!t.texels || t.width == 0 || t.height == 0) { g_err = "image texture without texels"; return false; }
                auto c = std::make_shared<ImageTexture>(); c->texels = t.texels; c->width = t.width; c->height = t.height;
                w.color_tex[i] = c; tex_state[i] = 1;
                return true;
            }
            case RT_TEX_NOISE_SOLID: {
                if (!t.perlin_vec || !t.perlin_perm) { g_err = "noise texture without generator tables"; return false; }
                auto c = std::make_shared<NoiseSolidTexture>();
                c->noise.random_vec = t.perlin_vec; c->noise.perm = t.perlin_perm;
                c->scale = vec(t.v[0], t.v[1], t.v[2]); c->samples = t.samples;
                w.float_tex[i] = c; tex_state[i] = 2;
                return true;
            }
            default:
                g_err = "oracle: unknown texture type";
                return false;
        }
    }
    ColorTex color(int i) {
        if (!build_texture(i) || tex_state[i] != 1) { if (g_err.empty()) g_err = "expected colour texture"; ok = false; return nullptr; }
        return w.color_tex[i];
    }
    FloatTex flt(int i) {
        if (!build_texture(i) || tex_state[i] != 2) { if (g_err.empty()) g_err = "expected float texture"; ok = false; return nullptr; }
        return w.float_tex[i];
    }
    bool build_material(uint32_t i) {
        const RtMaterial& m = d.materials[i];
        switch (m.type) {
            case RT_MAT_LAMBERTIAN: { auto p = std::make_unique<LambertianDiffuse>(); p->albedo = color(m.tex_a); w.materials[i] = std::move(p); break; }
            case RT_MAT_METAL: { auto p = std::make_unique<Metal>(); p->albedo = color(m.tex_a); p->roughness = flt(m.tex_b); w.materials[i] = std::move(p); break; }
            case RT_MAT_DIELECTRIC: { auto p = std::make_unique<Dielectric>(); p->ior = m.ior; w.materials[i] = std::move(p); break; }
            case RT_MAT_GLOSSY: {
                auto p = std::make_unique<Glossy>(); p->albedo = color(m.tex_a); p->roughness = flt(m.tex_b);
                if (m.tex_c >= 0) p->normal_map = color(m.tex_c);
                p->inv_ior = 1.0 / m.ior;  // glossy.rs:30
                w.materials[i] = std::move(p); break;
            }
            case RT_MAT_EMISSIVE: { auto p = std::make_unique<Emissive>(); p->emission_map = color(m.tex_a); w.materials[i] = std::move(p); break; }
            case RT_MAT_ISOTROPIC: { auto p = std::make_unique<Isotropic>(); p->albedo = color(m.tex_a); w.materials[i] = std::move(p); break; }
            case RT_MAT_NORMAL_DEBUG: { auto p = std::make_unique<NormalDebug>(); if (m.tex_c >= 0) p->normal_map = color(m.tex_c); w.materials[i] = std::move(p); break; }
            default: g_err = "unknown material type"; return false;
        }
        return ok;
    }
    const Material* mat(int i, int* idx) {
        if (i < 0 || uint32_t(i) >= d.n_materials) { g_err = "material index out of range"; ok = false; return nullptr; }
        *idx = i;
        return w.materials[i].get();
    }
    const Hit* build_node(uint32_t i) {
        if (i >= d.n_nodes) { g_err = "node index out of range"; ok = false; return nullptr; }
        if (w.nodes[i]) return w.nodes[i].get();
        const RtNode& n = d.nodes[i];
        auto child = [&](uint32_t k) -> const Hit* {
            if (k >= n.n_children || n.first_child + k >= d.n_child_indices) { g_err = "child index out of range"; ok = false; return nullptr; }
            return build_node(d.child_indices[n.first_child + k]);
        };
        int mi = -1;
        switch (n.type) {
            case RT_NODE_SPHERE: {
                const Material* m = mat(n.material, &mi);
                w.nodes[i] = std::make_unique<Sphere>(point(n.p[0], n.p[1], n.p[2]), n.p[3], m, mi);
                break;
            }
            case RT_NODE_PLANE: {
                const Material* m = mat(n.material, &mi);
                auto p = std::make_unique<Plane>(point(n.p[0], n.p[1], n.p[2]), vec(n.p[3], n.p[4], n.p[5]), vec(n.p[6], n.p[7], n.p[8]), m, mi);
                p->render_backface = (n.flags & RT_PLANE_RENDER_BACKFACE) != 0;
                w.nodes[i] = std::move(p);
                break;
            }
            case RT_NODE_MESH: {
                if (n.mesh < 0 || uint32_t(n.mesh) >= d.n_meshes) { g_err = "mesh index out of range"; ok = false; return nullptr; }
                const RtMesh& md = d.meshes[n.mesh];
                auto m = std::make_unique<TriangleMesh>();
                m->material = mat(n.material, &mi);
                m->material_index = mi;
                m->flat_shading = (md.flags & RT_MESH_FLAT_SHADING) != 0;
                m->hit_back_faces = (md.flags & RT_MESH_HIT_BACK_FACES) != 0;
                m->vertices.resize(md.n_positions);
                for (uint32_t k = 0; k < md.n_positions; k++) m->vertices[k] = point(md.positions[3 * k], md.positions[3 * k + 1], md.positions[3 * k + 2]);  // obj.rs:32
                m->vertex_normals.resize(md.n_normals);
                for (uint32_t k = 0; k < md.n_normals; k++) m->vertex_normals[k] = vec(md.normals[3 * k], md.normals[3 * k + 1], md.normals[3 * k + 2]);
                m->vertex_uvs.resize(md.n_uvs);
                for (uint32_t k = 0; k < md.n_uvs; k++) m->vertex_uvs[k] = vec(md.uvs[3 * k], md.uvs[3 * k + 1], md.uvs[3 * k + 2]);
                m->triangles.resize(md.n_triangles);
                for (uint32_t k = 0; k < md.n_triangles; k++) {
                    Triangle& t = m->triangles[k];
                    for (int c = 0; c < 3; c++) {
                        t.vert_indices[c] = md.tri_pos[3 * k + c];
                        t.normal_indices[c] = md.tri_nrm[3 * k + c];
                        if (t.vert_indices[c] >= md.n_positions || t.normal_indices[c] >= md.n_normals) { g_err = "triangle index out of range"; ok = false; return nullptr; }
                    }
                    t.has_uv = md.tri_uv && md.tri_uv[3 * k] >= 0 && md.tri_uv[3 * k + 1] >= 0 && md.tri_uv[3 * k + 2] >= 0;
                    for (int c = 0; c < 3; c++) {
                        t.uv_indices[c] = t.has_uv ? size_t(md.tri_uv[3 * k + c]) : 0;
                        if (t.has_uv && t.uv_indices[c] >= md.n_uvs) { g_err = "uv index out of range"; ok = false; return nullptr; }
                    }
                }
                m->finish();
                w.nodes[i] = std::move(m);
                break;
            }
            case RT_NODE_LIST: {
                auto l = std::make_unique<ObjectList>();
                l->disable_bounds_check = (n.flags & RT_LIST_DISABLE_BOUNDS_CHECK) != 0;
                for (uint32_t k = 0; k < n.n_children; k++) {
                    const Hit* c = child(k);
                    if (!c) return nullptr;
                    l->add(c);
                }
                w.nodes[i] = std::move(l);
                break;
            }
            case RT_NODE_TRANSFORM: {
                const Hit* c = child(0);
                if (!c) return nullptr;
                if (n.transform < 0 || uint32_t(n.transform) >= d.n_transforms) { g_err = "transform index out of range"; ok = false; return nullptr; }
                Mat4 m, inv, ident = {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
                std::memcpy(m.m, d.transforms[n.transform].m, sizeof m.m);
                std::memcpy(inv.m, d.transforms[n.transform].inv, sizeof inv.m);
                // "no op applied" is visible only as identity matrices; a Transform whose ops cancel
                // exactly would differ in its (culling-only) box padding.
                bool ops = std::memcmp(m.m, ident.m, sizeof m.m) != 0 || std::memcmp(inv.m, ident.m, sizeof m.m) != 0;
                w.nodes[i] = std::make_unique<Transform>(c, m, inv, ops);
                break;
            }
            case RT_NODE_BVH: {
                const Hit* a = child(0);
                const Hit* b = child(1);
                if (!a || !b) return nullptr;
                bool single = d.nodes[d.child_indices[n.first_child + 1]].type == RT_NODE_NULL;
                w.nodes[i] = std::make_unique<BvhNode>(a, b, single);
                break;
            }
            case RT_NODE_SKY: {
                auto s = std::make_unique<Sky>();
                s->material = mat(n.material, &mi);
                s->material_index = mi;
                w.nodes[i] = std::move(s);
                break;
            }
            case RT_NODE_SUN: {
                auto s = std::make_unique<Sun>();
                s->direction = vec(n.p[0], n.p[1], n.p[2]);  // already unit (sun.rs:27 done by the host)
                s->material = mat(n.material, &mi);
                s->material_index = mi;
                w.nodes[i] = std::move(s);
                break;
            }
            case RT_NODE_VOLUME: {
                auto v = std::make_unique<Volume>();
                v->boundary = child(0);
                if (!v->boundary) return nullptr;
                v->material = mat(n.material, &mi);
                v->material_index = mi;
                v->neg_inv_density = -1.0 / n.p[0];  // volume.rs:23
                w.nodes[i] = std::move(v);
                break;
            }
            case RT_NODE_NULL:
                w.nodes[i] = std::make_unique<NullObject>();
                break;
            default:
                g_err = "unknown node type";
                ok = false;
                return nullptr;
        }
        return ok ? w.nodes[i].get() : nullptr;
    }
};

std::unique_ptr<World> build_world(const RtSceneDesc* d) {
    if (!d || d->abi_version != RT_MI355_ABI_VERSION) { g_err = "bad scene description"; return nullptr; }
    auto w = std::make_unique<World>();
    w->color_tex.resize(d->n_textures);
    w->float_tex.resize(d->n_textures);
    w->materials.resize(d->n_materials);
    w->nodes.resize(d->n_nodes);
    Builder b{*d, *w, std::vector<int>(d->n_textures, 0)};
    for (uint32_t i = 0; i < d->n_materials; i++)
        if (!b.build_material(i)) return nullptr;
    w->world = b.build_node(d->world_root);
    w->lights = b.build_node(d->lights_root);
    if (!w->world || !w->lights || !b.ok) return nullptr;
    return w;
}

// Optional per-bounce trace (debugging / tests): 16 doubles per bounce:
// t, pos.xyz, material index, scatter kind (0 pdf, 1 ray, 2 absorbed, 3 emissive, -1 miss), pdf, scattering_pdf,
// normal.xyz, ray origin.xyz, ray dir.xy (dir.z in slot 15)
thread_local std::vector<double>* g_trace = nullptr;

// ------------------------------------------------------------------ camera.rs
struct Camera {
    RtCameraDesc c;
    RtRenderParams p;
    Vec4 position, first_pixel, pdu, pdv, basis_u, basis_v;
    double inv_sqrt_spt;
    size_t samples_per_pixel;
    Vec4 background;
    Camera(const RtCameraDesc& cd, const RtRenderParams& pp) : c(cd), p(pp) {
        position = point(cd.position[0], cd.position[1], cd.position[2]);
        first_pixel = point(cd.first_pixel[0], cd.first_pixel[1], cd.first_pixel[2]);
        pdu = vec(cd.pixel_delta_u[0], cd.pixel_delta_u[1], cd.pixel_delta_u[2]);
        pdv = vec(cd.pixel_delta_v[0], cd.pixel_delta_v[1], cd.pixel_delta_v[2]);
        basis_u = vec(cd.basis_u[0], cd.basis_u[1], cd.basis_u[2]);
        basis_v = vec(cd.basis_v[0], cd.basis_v[1], cd.basis_v[2]);
        inv_sqrt_spt = 1.0 / double(pp.sqrt_spt);                                  // camera.rs:52
        samples_per_pixel = size_t(pp.sqrt_spt) * pp.sqrt_spt * pp.thread_count;  // camera.rs:50-51
        background = pp.has_background ? vec(pp.background[0], pp.background[1], pp.background[2]) : vec(0, 0, 0);  // camera.rs:55,331
    }
    Vec4 pixel_sample_square(size_t sx, size_t sy, Rng& rng) const {  // camera.rs:334-341
        double rx = rng.uniform();
        double ry = rng.uniform();
        double x = (double(sx) + rx) * inv_sqrt_spt - 0.5;
        double y = (double(sy) + ry) * inv_sqrt_spt - 0.5;
        return pdu * x + pdv * y;
    }
    Vec4 defocus_disk_sample(Rng& rng) const {  // camera.rs:345-349
        Vec4 v = random_in_unit_disk(rng);
        return position + (basis_u * v[0] + basis_v * v[1]) * c.aperture_radius;
    }
    Ray get_ray(size_t px, size_t py, size_t sx, size_t sy, Rng& rng) const {  // camera.rs:260-280
        Vec4 pixel_center = first_pixel + (pdu * double(px)) + (pdv * double(py));
        Vec4 pixel_sample = pixel_center + pixel_sample_square(sx, sy, rng);
        Vec4 ray_origin = c.has_aperture ? defocus_disk_sample(rng) : position;
        Vec4 ray_direction = pixel_sample - ray_origin;
        return Ray(ray_origin, ray_direction);
    }
    Vec4 ray_color(const Ray& ray, const Hit* object, HittablePDF& lights_pdf, size_t depth, Rng& rng) const {  // camera.rs:282-332
        if (depth == 0) return vec(0.0, 0.0, 0.0);
        rng.cnt.rays++;
        HitRecord hit;
        if (object->test(ray, {0.001, INF}, rng, hit)) {
            Vec4 from_emission = hit.material->emit(hit);
            ScatterResult sr = hit.material->scatter(ray, hit, rng);
            size_t trace_at = 0;
            if (g_trace) {
                trace_at = g_trace->size();
                g_trace->insert(g_trace->end(), {hit.t, hit.hit_pos[0], hit.hit_pos[1], hit.hit_pos[2],
                                                 double(hit.material_index), double(int(sr.kind)), 0.0, 0.0,
                                                 hit.normal[0], hit.normal[1], hit.normal[2],
                                                 ray.origin[0], ray.origin[1], ray.origin[2], ray.dir[0], ray.dir[1]});
                g_trace->push_back(ray.dir[2]);
            }
            switch (sr.kind) {
                case ScatterKind::WithPDF: {
                    lights_pdf.origin = hit.hit_pos;
                    MixPDF mix_pdf;
                    mix_pdf.first = sr.pdf.get();
                    mix_pdf.second = &lights_pdf;
                    mix_pdf.mix = p.light_bias;
                    Ray scattered(hit.hit_pos, mix_pdf.generate(rng));
                    double pdf = mix_pdf.value(scattered.dir, rng);
                    double scattering_pdf = hit.material->scattering_pdf(ray, scattered, hit);
                    if (g_trace) { (*g_trace)[trace_at + 6] = pdf; (*g_trace)[trace_at + 7] = scattering_pdf; }
                    Vec4 scatter_color = ray_color(scattered, object, lights_pdf, depth - 1, rng);
                    Vec4 from_scatter = (scatter_color * sr.attenuation * scattering_pdf) / pdf;
                    return from_emission + from_scatter;
                }
                case ScatterKind::WithRay: {
                    Vec4 scatter_color = ray_color(sr.scattered, object, lights_pdf, depth - 1, rng);
                    Vec4 from_scatter = scatter_color * sr.attenuation;
                    return from_emission + from_scatter;
                }
                case ScatterKind::Absorbed:
                    return vec(0.0, 0.0, 0.0);
                case ScatterKind::Emissive:
                    return from_emission;
            }
        }
        if (g_trace) g_trace->insert(g_trace->end(), {INF, 0, 0, 0, -1, -1, 0, 0, 0, 0, 0,
                                                      ray.origin[0], ray.origin[1], ray.origin[2], ray.dir[0], ray.dir[1], ray.dir[2]});
        return background;
    }
};

bool row_owned(uint32_t y, const RtRenderParams& p) {
    if (p.band_rows == 0 || p.n_parts <= 1) return true;
    return (y / p.band_rows) % p.n_parts == p.part;
}

}  // namespace

// ------------------------------------------------------------------ C ABI
extern "C" {

const char* oracle_last_error(void) { return g_err.c_str(); }

int oracle_render(const RtSceneDesc* scene, const RtCameraDesc* camera, const RtRenderParams* params,
                  double* rgba_out, OracleStats* stats) {
    g_err.clear();
    if (!scene || !camera || !params || !rgba_out) { g_err = "NULL argument"; return RT_E_INVALID; }
    if (params->sqrt_spt == 0 || params->thread_count == 0) { g_err = "sqrt_spt and thread_count must be positive"; return RT_E_INVALID; }
    auto world = build_world(scene);
    if (!world) return RT_E_INVALID;
    Camera cam(*camera, *params);
    const uint32_t W = camera->image_width, H = camera->image_height;
    std::vector<uint32_t> rows;
    for (uint32_t y = 0; y < H; y++)
        if (row_owned(y, *params)) rows.push_back(y);
    const size_t npix = rows.size() * size_t(W);
    const uint32_t T = params->thread_count;
    std::vector<std::vector<Vec4>> thread_bufs(T);
    std::vector<Counters> counters(T);
    std::vector<std::thread> threads;
    auto t0 = std::chrono::steady_clock::now();
    // camera.rs:197-241: one OS thread per replica, each renders the whole (owned) frame
    for (uint32_t tid = 0; tid < T; tid++) {
        threads.emplace_back([&, tid]() {
            std::vector<Vec4>& buf = thread_bufs[tid];
            buf.assign(npix, vec(0.0, 0.0, 0.0));  // buffer.rs:15-28
            HittablePDF lights_pdf;
            lights_pdf.object = world->lights;
            lights_pdf.origin = point(0.0, 0.0, 0.0);
            Rng rng;
            const uint32_t S = params->sqrt_spt;
            for (size_t r = 0; r < rows.size(); r++) {
                uint32_t y = rows[r];
                for (uint32_t x = 0; x < W; x++) {
                    Vec4 color = vec(0.0, 0.0, 0.0);
                    for (uint32_t sy = 0; sy < S; sy++) {
                        for (uint32_t sx = 0; sx < S; sx++) {
                            rng.key(params->seed, tid, uint64_t(y) * W + x, sy * S + sx);
                            Ray ray = cam.get_ray(x, y, sx, sy, rng);
                            color = color + cam.ray_color(ray, world->world, lights_pdf, params->max_depth, rng);
                        }
                    }
                    color = color / double(cam.samples_per_pixel);  // camera.rs:229
                    buf[r * W + x] = color;
                }
            }
            counters[tid] = rng.cnt;
        });
    }
    // camera.rs:243-255: join in order, add each thread buffer into the (zeroed) main buffer
    std::vector<Vec4> main_buf(npix, vec(0.0, 0.0, 0.0));
    for (uint32_t tid = 0; tid < T; tid++) {
        threads[tid].join();
        for (size_t i = 0; i < npix; i++) main_buf[i] = main_buf[i] + thread_bufs[tid][i];
        std::vector<Vec4>().swap(thread_bufs[tid]);
    }
    auto t1 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < npix; i++) {
        rgba_out[4 * i + 0] = main_buf[i][0];
        rgba_out[4 * i + 1] = main_buf[i][1];
        rgba_out[4 * i + 2] = main_buf[i][2];
        rgba_out[4 * i + 3] = 0.0;
    }
    if (stats) {
        OracleStats s{};
        for (auto& c : counters) {
            s.rays += c.rays; s.node_tests += c.node_tests; s.tri_tests += c.tri_tests; s.prim_tests += c.prim_tests;
        }
        s.samples = uint64_t(npix) * cam.samples_per_pixel;
        s.seconds = std::chrono::duration<double>(t1 - t0).count();
        s.os_threads = T;
        *stats = s;
    }
    return RT_OK;
}

int oracle_node_bounds(const RtSceneDesc* scene, uint32_t node, double* out6) {
    g_err.clear();
    auto world = build_world(scene);
    if (!world) return RT_E_INVALID;
    if (node >= scene->n_nodes || !world->nodes[node]) { g_err = "node not reachable from world/lights"; return RT_E_INVALID; }
    Aabb b = world->nodes[node]->get_bounding_box();
    for (int i = 0; i < 3; i++) { out6[i] = b.b[0][i]; out6[3 + i] = b.b[1][i]; }
    return RT_OK;
}

static void octree_walk(const OctreeNode& n, size_t depth, uint64_t* o) {
    if (depth > o[4]) o[4] = depth;
    if (n.is_leaf) {
        o[1]++;
        if (n.leaf.empty()) o[2]++;
        o[3] += n.leaf.size();
        if (n.leaf.size() > o[5]) o[5] = n.leaf.size();
    } else {
        o[0]++;
        for (int i = 0; i < 8; i++) octree_walk(*n.children[i], depth + 1, o);
    }
}

int oracle_octree_stats(const RtSceneDesc* scene, uint32_t mesh, uint64_t* out6) {
    g_err.clear();
    auto world = build_world(scene);
    if (!world) return RT_E_INVALID;
    for (uint32_t i = 0; i < scene->n_nodes; i++) {
        if (scene->nodes[i].type == RT_NODE_MESH && uint32_t(scene->nodes[i].mesh) == mesh && world->nodes[i]) {
            auto* m = static_cast<const TriangleMesh*>(world->nodes[i].get());
            for (int k = 0; k < 6; k++) out6[k] = 0;
            octree_walk(*m->octree, 0, out6);
            return RT_OK;
        }
    }
    g_err = "mesh not reachable";
    return RT_E_INVALID;
}

int oracle_test_bounding_box(const double b6[6], const double o[3], const double d[3], double t_min, double t_max) {
    Aabb b = {{point(b6[0], b6[1], b6[2]), point(b6[3], b6[4], b6[5])}};
    Ray r(point(o[0], o[1], o[2]), vec(d[0], d[1], d[2]));
    return test_bounding_box(b, r, {t_min, t_max}) ? 1 : 0;
}

int oracle_world_hit(const RtSceneDesc* scene, const double o[3], const double d[3], double t_min, double t_max, double* out) {
    g_err.clear();
    auto world = build_world(scene);
    if (!world) return RT_E_INVALID;
    Rng rng;
    Ray r(point(o[0], o[1], o[2]), vec(d[0], d[1], d[2]));
    HitRecord h;
    if (!world->world->test(r, {t_min, t_max}, rng, h)) return 0;
    out[0] = h.t;
    for (int i = 0; i < 3; i++) { out[1 + i] = h.hit_pos[i]; out[4 + i] = h.normal[i]; }
    out[7] = h.u; out[8] = h.v; out[9] = h.front_face ? 1.0 : 0.0; out[10] = double(h.material_index);
    return 1;
}

int oracle_lights_pdf_value(const RtSceneDesc* scene, const double o[3], const double d[3], double* out) {
    g_err.clear();
    auto world = build_world(scene);
    if (!world) return RT_E_INVALID;
    Rng rng;
    *out = world->lights->pdf_value(point(o[0], o[1], o[2]), vec(d[0], d[1], d[2]), rng);
    return RT_OK;
}

int oracle_lights_random(const RtSceneDesc* scene, const double o[3], uint64_t seed, uint32_t n, double* out3n) {
    g_err.clear();
    auto world = build_world(scene);
    if (!world) return RT_E_INVALID;
    Rng rng;
    rng.key(seed, 0, 0, 0);
    for (uint32_t i = 0; i < n; i++) {
        Vec4 d = world->lights->random(point(o[0], o[1], o[2]), rng);
        out3n[3 * i] = d[0]; out3n[3 * i + 1] = d[1]; out3n[3 * i + 2] = d[2];
    }
    return RT_OK;
}

// Samples texture `tex` of the scene at (u, v, p): colour textures fill out3, float textures out3[0].
int oracle_texture_sample(const RtSceneDesc* scene, uint32_t tex, double u, double v, const double p[3], double* out3) {
    g_err.clear();
    if (!scene || scene->abi_version != RT_MI355_ABI_VERSION) { g_err = "bad scene description"; return RT_E_INVALID; }
    World w;
    w.color_tex.resize(scene->n_textures);
    w.float_tex.resize(scene->n_textures);
    Builder b{*scene, w, std::vector<int>(scene->n_textures, 0)};
    if (!b.build_texture(int(tex))) return RT_E_INVALID;
    Vec4 pt = point(p[0], p[1], p[2]);
    if (b.tex_state[tex] == 1) {
        Vec4 c = w.color_tex[tex]->sample(u, v, pt);
        out3[0] = c[0]; out3[1] = c[1]; out3[2] = c[2];
    } else {
        out3[0] = w.float_tex[tex]->sample(u, v, pt);
        out3[1] = out3[2] = 0.0;
    }
    return RT_OK;
}

void oracle_detmath(double x, double* out3) {
    det_sincos(x, &out3[0], &out3[1]);
    out3[2] = x > 0 ? det_log(x) : 0.0;
}

void oracle_detmath_inv(double y, double x, double* out3) {
    out3[0] = det_atan(y);
    out3[1] = det_atan2(y, x);
    out3[2] = det_acos(y);
}

double oracle_reflectance(double cos_theta, double ior_ratio) { return reflectance(cos_theta, ior_ratio); }

void oracle_onb_from_vec(const double w[3], double* out9) {
    Mat4 m = onb_from_vec(vec(w[0], w[1], w[2]));
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) out9[3 * c + r] = m.m[4 * r + c];
}

void oracle_refract(const double v[3], const double n[3], double ior_ratio, double* out3) {
    Vec4 r = refract(vec(v[0], v[1], v[2]), vec(n[0], n[1], n[2]), ior_ratio);
    out3[0] = r[0]; out3[1] = r[1]; out3[2] = r[2];
}

void oracle_rng_uniforms(uint64_t seed, uint32_t tid, uint64_t pixel, uint32_t stratum, uint32_t n, double* out) {
    Rng rng;
    rng.key(seed, tid, pixel, stratum);
    for (uint32_t i = 0; i < n; i++) out[i] = rng.uniform();
}

void oracle_rng_raw(uint64_t seed, uint32_t tid, uint64_t pixel, uint32_t stratum, uint32_t n, uint64_t* out) {
    Rng rng;
    rng.key(seed, tid, pixel, stratum);
    for (uint32_t i = 0; i < n; i++) out[i] = rng.next();
}

int oracle_trace_sample(const RtSceneDesc* scene, const RtCameraDesc* camera, const RtRenderParams* params,
                        uint32_t tid, uint32_t x, uint32_t y, uint32_t sx, uint32_t sy, double* rgb_out,
                        double* trace_out, uint32_t max_bounces) {
    g_err.clear();
    auto world = build_world(scene);
    if (!world) return RT_E_INVALID;
    Camera cam(*camera, *params);
    HittablePDF lights_pdf;
    lights_pdf.object = world->lights;
    lights_pdf.origin = point(0.0, 0.0, 0.0);
    Rng rng;
    rng.key(params->seed, tid, uint64_t(y) * camera->image_width + x, sy * params->sqrt_spt + sx);
    Ray ray = cam.get_ray(x, y, sx, sy, rng);
    std::vector<double> trace;
    g_trace = &trace;
    Vec4 c = cam.ray_color(ray, world->world, lights_pdf, params->max_depth, rng);
    g_trace = nullptr;
    rgb_out[0] = c[0]; rgb_out[1] = c[1]; rgb_out[2] = c[2];
    uint32_t n = uint32_t(trace.size() / 17);
    for (uint32_t i = 0; i < n && i < max_bounces; i++)
        for (int k = 0; k < 17; k++) trace_out[17 * i + k] = trace[17 * i + k];
    return int(n);
}

void oracle_get_ray(const RtCameraDesc* camera, const RtRenderParams* params, uint32_t tid, uint32_t x, uint32_t y,
                    uint32_t sx, uint32_t sy, double* out6) {
    Camera cam(*camera, *params);
    Rng rng;
    rng.key(params->seed, tid, uint64_t(y) * camera->image_width + x, sy * params->sqrt_spt + sx);
    Ray r = cam.get_ray(x, y, sx, sy, rng);
    for (int i = 0; i < 3; i++) { out6[i] = r.origin[i]; out6[3 + i] = r.dir[i]; }
}

}  // extern "C"
