// rt_device.h — device functions of the render path, templated on the arithmetic type R
// (double = the reference's precision, float = fast mode).  Hand-written for gfx950:
// 64-lane waves, per-lane BVH stack in LDS, no CUDA compatibility layer.
//
// Every function cites the reference code it restates.  Arithmetic is written in the
// reference's operation order; no fast-math, IEEE semantics for inf/NaN are relied upon
// exactly like the Rust code does (e.g. 1/0 = inf in Ray::new, src/ray.rs:20).
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>

#include "../../include/rt_detmath.h"
#include "../../include/rt_mi355.h"
#include "rt_scene.h"

namespace rt {

#define RT_DEV __device__ __forceinline__
// Notes from tuning: (1) a single `__noinline__` helper (to_unit) raised the shade kernel
// from 129 to 217 VGPRs through the call ABI; (2) restructuring shade() so that every heavy routine has one
// call site halved its code (71 KB -> 36 KB) but also raised it to 211 VGPRs and bought no time.  Everything
// is therefore inlined and shade() keeps the reference's per-material structure.

template <typename R> struct Lim;
template <> struct Lim<double> {
    static RT_DEV double inf() { return __builtin_huge_val(); }
    static RT_DEV double max() { return DBL_MAX; }
    static RT_DEV double eps() { return DBL_EPSILON; }  // f64::EPSILON (mesh.rs:77, plane.rs:74)
};
template <> struct Lim<float> {
    static RT_DEV float inf() { return __builtin_huge_valf(); }
    static RT_DEV float max() { return FLT_MAX; }
    static RT_DEV float eps() { return FLT_EPSILON; }
};

template <typename R> RT_DEV R pi() { return R(3.14159265358979323846264338327950288); }

// ------------------------------------------------------------------ vec4.rs (xyz part)
template <typename R>
struct V3 {
    R x, y, z;
};
template <typename R> RT_DEV V3<R> mk(R x, R y, R z) { return {x, y, z}; }
template <typename R> RT_DEV V3<R> ld3(const R* p) { return {p[0], p[1], p[2]}; }
template <typename R> RT_DEV V3<R> operator+(V3<R> a, V3<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename R> RT_DEV V3<R> operator-(V3<R> a, V3<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename R> RT_DEV V3<R> operator*(V3<R> a, V3<R> b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
template <typename R> RT_DEV V3<R> operator*(V3<R> a, R s) { return {a.x * s, a.y * s, a.z * s}; }
template <typename R> RT_DEV V3<R> operator/(V3<R> a, R s) { return {a.x / s, a.y / s, a.z / s}; }
template <typename R> RT_DEV V3<R> operator-(V3<R> a) { return {-a.x, -a.y, -a.z}; }
template <typename R> RT_DEV R dot(V3<R> a, V3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   // vec4.rs:109
template <typename R> RT_DEV R length_squared(V3<R> a) { return a.x * a.x + a.y * a.y + a.z * a.z; }  // vec4.rs:105
template <typename R> RT_DEV R length(V3<R> a) { return sqrt(length_squared(a)); }                    // vec4.rs:101
template <typename R> RT_DEV V3<R> cross(V3<R> a, V3<R> b) {                                          // vec4.rs:113
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename R> RT_DEV V3<R> to_unit(V3<R> a) { return a / length(a); }                         // vec4.rs:122
template <typename R> RT_DEV V3<R> reflect(V3<R> v, V3<R> n) { return v - n * (R(2) * dot(v, n)); }   // vec4.rs:135
template <typename R> RT_DEV V3<R> refract(V3<R> v, V3<R> n, R ior_ratio) {                           // vec4.rs:140-147
    R cos_theta = fmin(R(1), dot(-v, n));
    V3<R> perp = (v + (n * cos_theta)) * ior_ratio;
    V3<R> parallel = n * -sqrt(R(1) - length_squared(perp));
    return perp + parallel;
}

// Mat4 * Vec4 for affine matrices (mat4.rs:342-353); w = 1 for points, 0 for directions.
// The `m3 * w` term is kept even for w = 0 so that signed zeros come out as in the reference.
template <typename R> RT_DEV V3<R> xform_apply(const R* m, V3<R> v, R w) {
    return {m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * w,
            m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7] * w,
            m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] * w};
}
// from_columns(u, v, w, (0,0,0,1)) * vec(x, y, z) (mat4.rs:37-44, 342-353)
template <typename R> RT_DEV V3<R> basis_apply(V3<R> u, V3<R> v, V3<R> w, V3<R> r) {
    const R z = R(0);
    return {u.x * r.x + v.x * r.y + w.x * r.z + z * z,
            u.y * r.x + v.y * r.y + w.y * r.z + z * z,
            u.z * r.x + v.z * r.y + w.z * r.z + z * z};
}

// utils.rs:17-28
template <typename R> RT_DEV void onb_from_vec(V3<R> w, V3<R>& u, V3<R>& v) {
    V3<R> a = fabs(w.x) > R(0.9) ? mk<R>(0, 1, 0) : mk<R>(1, 0, 0);
    v = to_unit(cross(w, a));
    u = cross(w, v);
}
// utils.rs:31-36 (powi(5) = x2 = x*x; x4 = x2*x2; x4*x) with r0 = ((1 - ior_ratio) / (1 + ior_ratio))^2 taken from the
// material's constants (MaterialParams: the same operations, done once on the host)
template <typename R> RT_DEV R reflectance_r0(R cos_theta, R r0) {
    R x = R(1) - cos_theta;
    R x2 = x * x;
    R x4 = x2 * x2;
    return r0 + (R(1) - r0) * (x4 * x);
}
template <typename R> RT_DEV R reflectance(R cos_theta, R ior_ratio) {
    R r0 = (R(1) - ior_ratio) / (R(1) + ior_ratio);
    r0 = r0 * r0;
    R x = R(1) - cos_theta;
    R x2 = x * x;
    R x4 = x2 * x2;
    return r0 + (R(1) - r0) * (x4 * x);
}

// ------------------------------------------------------------------ RNG (DESIGN.md "RNG")
// Keyed SplitMix64 stream: one stream per (seed, replica, pixel, stratum); draw order follows
// SURVEY Appendix A.  Identical to oracle/oracle.cpp `struct Rng`.
struct Rng {
    uint64_t s;
    static RT_DEV uint64_t mix(uint64_t z) {
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    RT_DEV void key(uint64_t seed, uint32_t tid, uint64_t pixel, uint32_t stratum) {
        uint64_t k = mix(seed + 0x9E3779B97F4A7C15ull * (uint64_t(tid) + 1));
        k = mix(k ^ (pixel * 0xD1B54A32D192ED03ull + 0x8CB92BA72F3D8DD7ull));
        k = mix(k ^ (uint64_t(stratum) * 0xA0761D6478BD642Full + 0xE7037ED1A0B428DBull));
        s = k;
    }
    RT_DEV uint64_t next() {
        s += 0x9E3779B97F4A7C15ull;
        return mix(s);
    }
    RT_DEV uint32_t below(uint32_t n) { return uint32_t(((next() >> 32) * uint64_t(n)) >> 32); }
};
template <typename R> RT_DEV R rng_uniform(Rng& g);
template <> RT_DEV double rng_uniform<double>(Rng& g) { return double(g.next() >> 11) * (1.0 / 9007199254740992.0); }
template <> RT_DEV float rng_uniform<float>(Rng& g) { return float(g.next() >> 40) * (1.0f / 16777216.0f); }
// sin / cos / ln of the sampling routines: f64 uses the deterministic functions shared with the
// oracle (include/rt_detmath.h), f32 the device libm.
RT_DEV void sincos_r(double x, double& s, double& c) { det_sincos(x, &s, &c); }
RT_DEV void sincos_r(float x, float& s, float& c) { s = sinf(x); c = cosf(x); }
RT_DEV double log_r(double x) { return det_log(x); }
RT_DEV float log_r(float x) { return logf(x); }
template <typename R> RT_DEV R rng_normal(Rng& g) {  // Box-Muller, cosine branch
    R u1 = rng_uniform<R>(g);
    R u2 = rng_uniform<R>(g);
    R r = sqrt(R(-2) * log_r(R(1) - u1));
    R s, c;
    sincos_r(R(2) * pi<R>() * u2, s, c);
    return r * c;
}
template <typename R> RT_DEV V3<R> random_unit(Rng& g) {  // vec4.rs:42-48
    R x = rng_normal<R>(g);
    R y = rng_normal<R>(g);
    R z = rng_normal<R>(g);
    return to_unit(mk<R>(x, y, z));
}
// reflected + random_unit(rng) * fuzz * length(reflected)  (metal.rs:33-35, glossy.rs:66-68).  For fuzz == 0 - a polished
// metal, the clear coat of `glossy (..) (constant 0)` in the reference's own scenes - the random term is (n * 0) * len = +-0
// per component: n is a unit vector of three finite normal samples, so with a finite length the sum is `reflected` itself bit
// for bit unless a component of `reflected` is a zero (-0 + +0 = +0).  Then the three Box-Muller samples (3 x ln, sqrt,
// sin / cos: ~800 f64 instructions that a wave executes for ALL its lanes as soon as one lane reflects specularly) are not
// computed; the generator still moves on by their six draws (SplitMix64: six increments).  Not covered: all three normal
// samples exactly zero (u1 = 0 three times in a row, 2^-159), where the reference normalises a zero vector and gets NaN.
template <typename R> RT_DEV V3<R> fuzzy_reflection(V3<R> reflected, R fuzz, Rng& g) {
    const R lim = sizeof(R) == 8 ? R(1e150) : R(1e18);  // below it length_squared cannot overflow
    const bool plain = fuzz == R(0) && reflected.x != R(0) && reflected.y != R(0) && reflected.z != R(0) &&
                       fabs(reflected.x) < lim && fabs(reflected.y) < lim && fabs(reflected.z) < lim;
#ifdef RT_WHATIF_NO_FUZZ  // timing experiment only (wrong frames): what would k_wf_shade cost if NO vertex computed the three normal samples?
    if (true) {
#else
    if (plain) {
#endif
        g.s += 6ull * 0x9E3779B97F4A7C15ull;
        return reflected;
    }
    return reflected + random_unit<R>(g) * fuzz * length(reflected);
}
template <typename R> RT_DEV V3<R> random_cosine(Rng& g) {  // vec4.rs:50-61
    R r1 = rng_uniform<R>(g);
    R r2 = rng_uniform<R>(g);
    R phi = r1 * R(2) * pi<R>();
    R sqrt_r2 = sqrt(r2);
    R sn, cs;
    sincos_r(phi, sn, cs);
    R x = cs * sqrt_r2;
    R y = sn * sqrt_r2;
    R z = sqrt(R(1) - r2);
    return mk<R>(x, y, z);
}

// Inverse trigonometric functions of the UV maps (sphere.rs:30-38, sky.rs:40-46), kept OUT of line on purpose: inlined, hipcc
// materialises the thirteen f64 polynomial coefficients of acos (and atan2's) in VGPR pairs at the top of the kernel - they are
// loop-invariant - and keeps them live through the whole of k_wf_shade, 26 registers of a kernel whose occupancy is set by its
// registers, for code that only sphere and sky hits execute.  A call costs those hits a few dozen cycles.
// f64: the deterministic functions shared with the oracle (include/rt_detmath.h): u and v feed texel lookups and checker
// parities, where a one-ulp difference between libms would flip a whole texel; f32 (no bit parity): the device libm.
__attribute__((noinline)) RT_DEV double uv_acos(double x) { return det_acos(x); }
__attribute__((noinline)) RT_DEV double uv_atan2(double y, double x) { return det_atan2(y, x); }
__attribute__((noinline)) RT_DEV float uv_acos(float x) { return acosf(x); }
__attribute__((noinline)) RT_DEV float uv_atan2(float y, float x) { return atan2f(y, x); }

// ------------------------------------------------------------------ ray.rs
template <typename R>
struct Ray {
    V3<R> o, d, inv;
};
template <typename R> RT_DEV Ray<R> make_ray(V3<R> o, V3<R> d) {  // ray.rs:19-33
    return {o, d, mk<R>(R(1) / d.x, R(1) / d.y, R(1) / d.z)};
}
template <typename R> RT_DEV V3<R> ray_at(const Ray<R>& r, R t) { return r.o + (r.d * t); }  // ray.rs:35

// aabb.rs:50-87, Williams et al.; `sign[i] = inv_dir[i] < 0` (ray.rs:21-25)
template <typename R> RT_DEV bool test_bounding_box(const Bounds<R>& b, const Ray<R>& ray, R t_lo, R t_hi) {
    bool sx = ray.inv.x < R(0), sy = ray.inv.y < R(0), sz = ray.inv.z < R(0);
    R t_min = ((sx ? b.hi[0] : b.lo[0]) - ray.o.x) * ray.inv.x;
    R t_max = ((sx ? b.lo[0] : b.hi[0]) - ray.o.x) * ray.inv.x;
    R ty_min = ((sy ? b.hi[1] : b.lo[1]) - ray.o.y) * ray.inv.y;
    R ty_max = ((sy ? b.lo[1] : b.hi[1]) - ray.o.y) * ray.inv.y;
    if ((t_min > ty_max) || (ty_min > t_max)) return false;
    if (ty_min > t_min) t_min = ty_min;
    if (ty_max < t_max) t_max = ty_max;
    R tz_min = ((sz ? b.hi[2] : b.lo[2]) - ray.o.z) * ray.inv.z;
    R tz_max = ((sz ? b.lo[2] : b.hi[2]) - ray.o.z) * ray.inv.z;
    if ((t_min > tz_max) || (tz_min > t_max)) return false;
    if (tz_min > t_min) t_min = tz_min;
    if (tz_max < t_max) t_max = tz_max;
    return t_min < t_hi && t_max > t_lo;
}

// ------------------------------------------------------------------ primitives
// sphere.rs:40-62: nearest root in (t_lo, t_hi), un-normalised direction.  INCL: t_hi itself is accepted as well (the scene
// program decides ties by the reference's visiting order, see hit_takes_over).
template <typename R, bool INCL = false> RT_DEV bool sphere_test(const SpherePrim<R>& s, const Ray<R>& ray, R t_lo, R t_hi, R& t_out) {
    V3<R> center_diff = ray.o - ld3(s.center);
    R a = length_squared(ray.d);
    R half_b = dot(ray.d, center_diff);
    R c = length_squared(center_diff) - s.radius * s.radius;
    R discriminant = half_b * half_b - a * c;
    if (discriminant < R(0)) return false;
    R d_sqrt = sqrt(discriminant);
    R root = (-half_b - d_sqrt) / a;
    if (root <= t_lo || (INCL ? t_hi < root : t_hi <= root)) {
        root = (-half_b + d_sqrt) / a;
        if (root <= t_lo || (INCL ? t_hi < root : t_hi <= root)) return false;
    }
    t_out = root;
    return true;
}

// plane.rs:66-89
template <typename R, bool INCL = false> RT_DEV bool plane_test(const PlanePrim<R>& p, const Ray<R>& ray, R t_lo, R t_hi, R& t_out, R& u_out, R& v_out) {
    V3<R> normal = ld3(p.normal);
    R dot_ray_normal = dot(normal, ray.d);
    R dd = p.backface ? fabs(dot_ray_normal) : -dot_ray_normal;
    if (dd < Lim<R>::eps()) return false;
    V3<R> corner = ld3(p.corner);
    R hit_t = dot(normal, corner - ray.o) / dot_ray_normal;
    if (hit_t <= t_lo || (INCL ? t_hi < hit_t : t_hi <= hit_t)) return false;
    V3<R> hit_pos = ray_at(ray, hit_t);
    V3<R> local_pos = hit_pos - corner;
    R u = dot(local_pos, ld3(p.inv_u));
    R v = dot(local_pos, ld3(p.inv_v));
    if (u < R(0) || u > R(1) || v < R(0) || v > R(1)) return false;
    t_out = hit_t;
    u_out = u;
    v_out = v;
    return true;
}

// ------------------------------------------------------------------ closest-hit state
template <typename R>
struct Best {
    R t;          // closest_t so far (Interval max handed to every later test)
    int32_t pc;   // op that produced it, -1 = none
    int32_t tri;  // mesh hits: triangle slot (absolute index into tris/attrs)
    R u, v;       // mesh: barycentrics; plane: (u, v)
};

// Closest-hit bookkeeping of the scene program for sphere / quad ops.  Their tests run with the upper end of the interval
// INCLUDED; a candidate at exactly the current closest distance takes over only if the reference would have visited it
// first (its tests use strict `t < closest`, list.rs:58-74, bvh.rs:84-101: the first one visited wins).  Op::skip of a
// primitive op is its RANK in the reference's depth-first order: the ops of a rebuilt primitive group (rt_compile.cpp)
// are not in that order; everywhere else rank order is program order and this rule changes nothing.
template <typename R> RT_DEV bool hit_takes_over(const SceneView<R>& sc, R t, const Op& op, const Best<R>& best) {
    if (!(t == best.t)) return true;  // strictly nearer (or NaN, which the reference's interval test lets through as well)
    return best.pc >= 0 && op.skip < sc.ops[best.pc].skip;
}

struct LaneCounters {
    uint32_t rays = 0, mesh_rays = 0, node_visits = 0, tri_tests = 0, prim_tests = 0;
};

// mesh.rs:62-107 Moeller-Trumbore with the reference's cull and interval rules
template <typename R, bool STATS>
RT_DEV void mesh_traverse(const SceneView<R>& sc, const MeshInst& mi, const Ray<R>& ray, R t_lo, Best<R>& best, int32_t pc,
                          int* stack, int stride, LaneCounters& cnt) {
    const BvhNode<R>* nodes = sc.nodes + mi.node_base;
    const TriRec<R>* tris = sc.tris + mi.tri_base;
    const bool hit_back = (mi.flags & RT_MESH_HIT_BACK_FACES) != 0;
    // Slab tests as t = b * inv - o * inv (culling only, any conservative test is admissible).
    // A zero direction component gives inv = +-inf and inf - inf = NaN in that form, so the
    // inverse used HERE is clamped to a huge finite value: a ray parallel to a slab is then
    // "inside forever" or "outside forever", which is exact.
    const R big = sizeof(R) == 8 ? R(1e150) : R(1e18);
    const V3<R> inv = {fabs(ray.inv.x) > big ? copysign(big, ray.inv.x) : ray.inv.x,
                       fabs(ray.inv.y) > big ? copysign(big, ray.inv.y) : ray.inv.y,
                       fabs(ray.inv.z) > big ? copysign(big, ray.inv.z) : ray.inv.z};
    const V3<R> oi = ray.o * inv;
    int sp = 0;
    int32_t cur = 0;
    if (STATS) cnt.mesh_rays++;
    for (;;) {
        if (cur >= 0) {
            const BvhNode<R>& n = nodes[cur];
            if (STATS) cnt.node_visits++;
            R t0x = n.lo0[0] * inv.x - oi.x, t1x = n.hi0[0] * inv.x - oi.x;
            R t0y = n.lo0[1] * inv.y - oi.y, t1y = n.hi0[1] * inv.y - oi.y;
            R t0z = n.lo0[2] * inv.z - oi.z, t1z = n.hi0[2] * inv.z - oi.z;
            R near0 = fmax(fmax(fmin(t0x, t1x), fmin(t0y, t1y)), fmax(fmin(t0z, t1z), t_lo));
            R far0 = fmin(fmin(fmax(t0x, t1x), fmax(t0y, t1y)), fmin(fmax(t0z, t1z), best.t));
            R s0x = n.lo1[0] * inv.x - oi.x, s1x = n.hi1[0] * inv.x - oi.x;
            R s0y = n.lo1[1] * inv.y - oi.y, s1y = n.hi1[1] * inv.y - oi.y;
            R s0z = n.lo1[2] * inv.z - oi.z, s1z = n.hi1[2] * inv.z - oi.z;
            R near1 = fmax(fmax(fmin(s0x, s1x), fmin(s0y, s1y)), fmax(fmin(s0z, s1z), t_lo));
            R far1 = fmin(fmin(fmax(s0x, s1x), fmax(s0y, s1y)), fmin(fmax(s0z, s1z), best.t));
            int32_t c0 = n.c0, c1 = n.c1;
            bool h0 = (near0 <= far0) && c0 != kEmptyChild;
            bool h1 = (near1 <= far1) && c1 != kEmptyChild;
            if (h0 && h1) {
                bool first0 = near0 <= near1;
                stack[sp * stride] = first0 ? c1 : c0;
                sp++;
                cur = first0 ? c0 : c1;
                continue;
            }
            if (h0) { cur = c0; continue; }
            if (h1) { cur = c1; continue; }
        } else {
            uint32_t code = uint32_t(~cur);
            uint32_t first = code >> 3, count = (code & 7u) + 1u;
            for (uint32_t i = 0; i < count; i++) {
                const TriRec<R>& tr = tris[first + i];
                if (STATS) cnt.tri_tests++;
                V3<R> edge1 = ld3(tr.e1), edge2 = ld3(tr.e2);
                V3<R> ray_x_edge2 = cross(ray.d, edge2);
                R det = dot(edge1, ray_x_edge2);
                R dd = hit_back ? fabs(det) : det;
                if (dd < Lim<R>::eps()) continue;
                R inv_det = R(1) / det;
                V3<R> b = ray.o - ld3(tr.v0);
                R u = dot(b, ray_x_edge2) * inv_det;
                if (u < R(0) || u > R(1)) continue;
                V3<R> b_x_edge1 = cross(b, edge1);
                R v = dot(ray.d, b_x_edge1) * inv_det;
                if (v < R(0) || u + v > R(1)) continue;
                R t = dot(edge2, b_x_edge1) * inv_det;
                if (t <= t_lo || best.t <= t) continue;
                best.t = t;
                best.pc = pc;
                best.tri = int32_t(mi.tri_base + first + i);
                best.u = u;
                best.v = v;
            }
        }
        if (sp == 0) break;
        sp--;
        cur = stack[sp * stride];
    }
}

// Ray in the space of `chain` (outermost transform first): Transform::test, transform.rs:124-127
template <typename R> RT_DEV Ray<R> ray_in_chain(const SceneView<R>& sc, const Ray<R>& wray, int32_t chain) {
    int32_t b = sc.chain_offsets[chain], e = sc.chain_offsets[chain + 1];
    if (b == e) return wray;
    V3<R> o = wray.o, d = wray.d;
    for (int32_t i = b; i < e; i++) {
        const Xform<R>& x = sc.xforms[sc.chain_items[i]];
        o = xform_apply(x.inv, o, R(1));
        d = xform_apply(x.inv, d, R(0));
    }
    return make_ray(o, d);
}

// Scene tables are never written by a kernel.  Reading them through a CONSTANT-address-space pointer tells the compiler so:
// a load whose address is wave-uniform then becomes a scalar load (s_load into SGPRs, scalar cache) even behind the kernel's
// own stores to other buffers, where it otherwise has to assume a clobber and issues a vector load.
template <typename T> RT_DEV const __attribute__((address_space(4))) T* as_const_mem(const T* p) {
    return (const __attribute__((address_space(4))) T*)(p);
}
// ray_in_chain for a WAVE-UNIFORM chain index (k_wf_mesh enters one mesh op at a time): chain and matrices through scalar loads.
template <typename R> RT_DEV Ray<R> ray_in_chain_uniform(const SceneView<R>& sc, const Ray<R>& wray, int32_t chain) {
    const auto* offs = as_const_mem(sc.chain_offsets);
    const auto* items = as_const_mem(sc.chain_items);
    const auto* xf = as_const_mem(sc.xforms);
    int32_t b = offs[chain], e = offs[chain + 1];
    if (b == e) return wray;
    V3<R> o = wray.o, d = wray.d;
    for (int32_t i = b; i < e; i++) {
        R inv[12];
        const int32_t x = items[i];
#pragma unroll
        for (int k = 0; k < 12; k++) inv[k] = xf[x].inv[k];
        o = xform_apply(inv, o, R(1));
        d = xform_apply(inv, d, R(0));
    }
    return make_ray(o, d);
}

// Volume::test (volume.rs:33-71) inside the scene program:  VOL_BEGIN <boundary ops> VOL_MID <boundary ops> VOL_END.
// While a volume searches its boundary the caller's search state (closest hit so far, lower end of its interval) waits in a
// frame; a volume INSIDE a boundary (its test runs once in each of the outer volume's two searches, draws included, in program
// order) takes the next frame: kMaxVolDepth levels, kept in registers (every index below is a compile-time constant).
template <typename R>
struct VolFrames {  // two levels, spelled out member by member: arrays indexed by `depth` end up in scratch memory
    Best<R> saved0, saved1;
    R t_lo0, t_lo1;
    R enter0, enter1;  // t of the boundary's entry hit (volume.rs:34)
    int depth = 0;
};
static_assert(kMaxVolDepth == 2, "VolFrames spells out two levels");
// volume.rs:34: boundary.test(ray, Interval::UNIVERSE)
template <typename R> RT_DEV void vol_begin(VolFrames<R>& f, Best<R>& best, R& t_lo) {
    if (f.depth == 0) { f.saved0 = best; f.t_lo0 = t_lo; }
    else { f.saved1 = best; f.t_lo1 = t_lo; }
    f.depth++;
    best.t = Lim<R>::inf();
    best.pc = -1;
    t_lo = -Lim<R>::inf();
}
// volume.rs:35-37: no entry hit -> the volume is missed (returns true: jump to Op::skip); else second search over (t_enter + 0.0001, inf)
template <typename R> RT_DEV bool vol_mid(VolFrames<R>& f, Best<R>& best, R& t_lo) {
    if (best.pc < 0) {
        f.depth--;
        if (f.depth == 0) { best = f.saved0; t_lo = f.t_lo0; }
        else { best = f.saved1; t_lo = f.t_lo1; }
        return true;
    }
    if (f.depth == 1) f.enter0 = best.t;
    else f.enter1 = best.t;
    t_lo = best.t + R(0.0001);
    best.t = Lim<R>::inf();
    best.pc = -1;
    return false;
}
// volume.rs:38-68: clamp to the caller's interval, draw the free-flight distance; a scattering event becomes the closest hit
template <typename R> RT_DEV void vol_end(const SceneView<R>& sc, VolFrames<R>& f, Best<R>& best, R& t_lo, const Ray<R>& cur, const Op& op, int32_t pc, Rng& rng) {
    const bool has_exit = best.pc >= 0;
    const R t_exit = best.t;
    R vol_enter;
    f.depth--;
    if (f.depth == 0) { best = f.saved0; t_lo = f.t_lo0; vol_enter = f.enter0; }
    else { best = f.saved1; t_lo = f.t_lo1; vol_enter = f.enter1; }
    if (has_exit) {
        R t_min = fmax(vol_enter, t_lo);
        R t_max = fmin(t_exit, best.t);
        if (!(t_min >= t_max)) {
            t_min = fmax(t_min, R(0));
            R ray_len = length(cur.d);
            R dist_inside = (t_max - t_min) * ray_len;
            R uu = rng_uniform<R>(rng);
            R hit_dist = sc.volumes[op.arg].neg_inv_density * (uu == R(0) ? -Lim<R>::inf() : log_r(uu));
            if (!(hit_dist > dist_inside)) {
                best.t = t_min + hit_dist / ray_len;
                best.pc = pc;
                best.tri = -1;
                best.u = R(0);
                best.v = R(0);
            }
        }
    }
}

// world.test(ray, Interval(t_lo, inf)) — closest hit over the whole scene program.
// Reproduces ObjectList::test (list.rs:58-74), BoundingVolumeHierarchyNode::test
// (bvh.rs:84-101) and Transform::test (transform.rs:122-139): depth-first, fixed order,
// every test sees the interval (t_lo, closest_t so far).
// VOL: the program may contain volume ops (OP_VOL_*), which draw from the path's RNG during the search
// (volume.rs:47); `rng` may be null otherwise.
template <typename R, bool STATS, bool VOL = false>
RT_DEV void world_test(const SceneView<R>& sc, const Ray<R>& wray, const R t_lo_outer, Best<R>& best, int* stack, int stride, LaneCounters& cnt,
                       Rng* rng = nullptr) {
    best.t = Lim<R>::inf();
    best.pc = -1;
    best.tri = -1;
    best.u = R(0);
    best.v = R(0);
    Ray<R> cur = wray;
    int32_t pc = 0;
    R t_lo = t_lo_outer;   // lower end of the current search interval (changes only inside a volume's boundary tests)
    VolFrames<R> vol;      // the callers' search states while volumes test their boundaries
    if (STATS) cnt.rays++;
    for (;;) {
        const Op op = sc.ops[pc];
        if (op.type == OP_END) break;
        switch (op.type) {
            case OP_VOL_BEGIN:
                if constexpr (VOL) vol_begin(vol, best, t_lo);
                break;
            case OP_VOL_MID:
                if constexpr (VOL) {
                    if (vol_mid(vol, best, t_lo)) {
                        pc = op.skip;
                        continue;
                    }
                }
                break;
            case OP_VOL_END:
                if constexpr (VOL) vol_end(sc, vol, best, t_lo, cur, op, pc, *rng);
                break;
            case OP_BOUNDS:
                if (!test_bounding_box(sc.bounds[op.arg], cur, t_lo, best.t)) {
                    pc = op.skip;
                    continue;
                }
                break;
            case OP_XFORM_PUSH: {
                const Xform<R>& x = sc.xforms[op.arg];
                cur = make_ray(xform_apply(x.inv, cur.o, R(1)), xform_apply(x.inv, cur.d, R(0)));
                break;
            }
            case OP_XFORM_POP:
                cur = ray_in_chain(sc, wray, op.chain);
                break;
            case OP_SPHERE: {
                R t;
                if (STATS) cnt.prim_tests++;
                if (sphere_test<R, true>(sc.spheres[op.arg], cur, t_lo, best.t, t) && hit_takes_over(sc, t, op, best)) {
                    best.t = t;
                    best.pc = pc;
                }
                break;
            }
            case OP_PLANE: {
                R t, u, v;
                if (STATS) cnt.prim_tests++;
                if (plane_test<R, true>(sc.planes[op.arg], cur, t_lo, best.t, t, u, v) && hit_takes_over(sc, t, op, best)) {
                    best.t = t;
                    best.pc = pc;
                    best.u = u;
                    best.v = v;
                }
                break;
            }
            case OP_MESH:
                mesh_traverse<R, STATS>(sc, sc.meshes[op.arg], cur, t_lo, best, pc, stack, stride, cnt);
                break;
            case OP_SKY:  // sky.rs:28-33: hit at t = inf unless `inf > t.max`
                if (STATS) cnt.prim_tests++;
                if (!(Lim<R>::inf() > best.t)) {
                    best.t = Lim<R>::inf();
                    best.pc = pc;
                }
                break;
            case OP_SUN: {  // sun.rs:33-43
                if (STATS) cnt.prim_tests++;
                const SunPrim<R>& s = sc.suns[op.arg];
                V3<R> unit_dir = to_unit(cur.d);
                if (!(fabs(dot(ld3(s.direction), unit_dir) - R(1)) > R(0.001)) && !(Lim<R>::max() >= best.t)) {
                    best.t = Lim<R>::max();
                    best.pc = pc;
                }
                break;
            }
            default:
                break;
        }
        pc++;
    }
}

// ------------------------------------------------------------------ small tables in LDS
// Copies the first `staged` bytes of a packed small-table blob into LDS (whole workgroup, 16 B per lane per trip) and
// returns a view whose pointers to the tables that lie completely inside that prefix point into LDS; the others keep
// pointing to global memory.  Must be called by every thread of the block.
// ALL: the whole blob is staged (staged == lay.total_bytes): every table pointer is an LDS pointer and the compiler KNOWS it
// (ds_read instructions).  Otherwise the choice is made at run time and the tables are read with flat instructions.
template <typename R, bool ALL>
RT_DEV SceneView<R> scene_tables_to_lds(const SceneView<R>& g, const SmallLayout& lay, const char* blob, char* lds, uint32_t staged) {
    const uint32_t n16 = (staged + 15u) / 16u;
    const uint4* src = reinterpret_cast<const uint4*>(blob);
    uint4* dst = reinterpret_cast<uint4*>(lds);
    for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
    SceneView<R> v = g;
#define RT_REMAP(field, T, tbl) if (ALL || lay.end[tbl] <= staged) v.field = reinterpret_cast<const T*>(lds + lay.begin[tbl])
    RT_REMAP(ops, Op, ST_OPS);
    RT_REMAP(bounds, Bounds<R>, ST_BOUNDS);
    RT_REMAP(chain_offsets, int32_t, ST_CHAIN_OFFSETS);
    RT_REMAP(chain_items, int32_t, ST_CHAIN_ITEMS);
    RT_REMAP(xforms, Xform<R>, ST_XFORMS);
    RT_REMAP(spheres, SpherePrim<R>, ST_SPHERES);
    RT_REMAP(planes, PlanePrim<R>, ST_PLANES);
    RT_REMAP(suns, SunPrim<R>, ST_SUNS);
    RT_REMAP(meshes, MeshInst, ST_MESHES);
    RT_REMAP(materials, MaterialRec, ST_MATERIALS);
    RT_REMAP(material_params, MaterialParams<R>, ST_MATERIAL_PARAMS);
    RT_REMAP(textures, TextureRec<R>, ST_TEXTURES);
    RT_REMAP(lights, LightRec, ST_LIGHTS);
#undef RT_REMAP
    return v;
}

// ------------------------------------------------------------------ HitRecord (object.rs:32-72)
// ------------------------------------------------------------------ textures (texture/*.rs)
template <typename R> RT_DEV uint32_t as_u32_sat(R x) {  // Rust `as u32`
    if (!(x > R(0))) return 0u;
    if (x >= R(4294967295.0)) return 4294967295u;
    return uint32_t(x);
}
template <typename R> RT_DEV int32_t as_i32_sat(R x) {
    if (x != x) return 0;
    if (x <= R(-2147483648.0)) return INT32_MIN;
    if (x >= R(2147483647.0)) return INT32_MAX;
    return int32_t(x);
}
RT_DEV double sin_r(double x) { return det_sin(x); }
RT_DEV float sin_r(float x) { return sinf(x); }

// noise/perlin.rs:81-101 (sample) + :57-77 (trilinear_interpolation), arithmetic in the reference's order
template <typename R> RT_DEV R perlin_sample(const SceneView<R>& sc, uint32_t gen, V3<R> p) {
    const R u = p.x - floor(p.x), v = p.y - floor(p.y), w = p.z - floor(p.z);
    const uint32_t i = uint32_t(as_i32_sat(floor(p.x))), j = uint32_t(as_i32_sat(floor(p.y))), k = uint32_t(as_i32_sat(floor(p.z)));
    const R uu = u * u * (R(3) - R(2) * u), vv = v * v * (R(3) - R(2) * v), ww = w * w * (R(3) - R(2) * w);
    const uint32_t* perm = sc.perlin_perm + size_t(gen) * 768;
    const R* vec = sc.perlin_vec + size_t(gen) * 768;
    R acc = R(0);
    // one corner per trip, NOT unrolled: eight unrolled corners hoist 48 loads and push the whole shade kernel
    // to 256 VGPRs; noise textures are rare and the order of the additions is the reference's either way
#pragma clang loop unroll(disable)
    for (uint32_t corner = 0; corner < 8; corner++) {
        const uint32_t di = corner >> 2, dj = (corner >> 1) & 1u, dk = corner & 1u;
        uint32_t idx = perm[(di + i) & 255u] ^ perm[256u + ((dj + j) & 255u)] ^ perm[512u + ((dk + k) & 255u)];
        V3<R> c = mk<R>(vec[3 * idx], vec[3 * idx + 1], vec[3 * idx + 2]);
        R fi = R(di), fj = R(dj), fk = R(dk);
        V3<R> v_weight = mk<R>(u - fi, v - fj, w - fk);
        acc += (fi * uu + (R(1) - fi) * (R(1) - uu)) * (fj * vv + (R(1) - fj) * (R(1) - vv)) *
               (fk * ww + (R(1) - fk) * (R(1) - ww)) * dot(c, v_weight);
    }
    return acc;
}
// noise/perlin.rs:103-113 + texture/noise.rs:28-37
template <typename R> RT_DEV R noise_solid_sample(const SceneView<R>& sc, const TextureRec<R>& tx, V3<R> p) {
    V3<R> q = mk<R>(p.x * tx.v[0], p.y * tx.v[1], p.z * tx.v[2]);
    const R qz = q.z;
    R acc = R(0), weight = R(1);
#pragma clang loop unroll(disable)
    for (int32_t s = 0; s < tx.aux; s++) {
        acc += weight * perlin_sample(sc, tx.data, q);
        weight *= R(0.5);
        q = q * R(2);
    }
    return R(0.5) * (R(1) + sin_r(qz + R(10) * fabs(acc)));
}
// texture/image.rs:37-53: repeat, nearest neighbour; texels are the f32 values of Buffer::from_image
template <typename R> RT_DEV V3<R> image_sample(const SceneView<R>& sc, const TextureRec<R>& tx, R u, R v) {
    u = u - floor(u);
    v = v - floor(v);
    R w = R(tx.width) - R(0.001), h = R(tx.height) - R(0.001);
    uint32_t x = as_u32_sat(u * w), y = as_u32_sat(v * h);
    if (x >= tx.width) x = tx.width - 1;  // unreachable in f64 (u < 1); guards the f32 kernels
    if (y >= tx.height) y = tx.height - 1;
    const float* t = sc.texels + (size_t(tx.data) + size_t(y) * tx.width + x) * 3;
    return mk<R>(R(t[0]), R(t[1]), R(t[2]));
}

// Programs made of constants, uv_debug and checkers only (the common case) need no value stack: walk from the
// root (the program's last op) down the checker's chosen input to a leaf.  CHECKER ops carry the op indices of
// their inputs' roots for this (aux = even, data = odd).
template <typename R> RT_DEV V3<R> eval_texture_simple(const SceneView<R>& sc, int32_t prog, R u, R v, V3<R> p) {
    int32_t t = (prog & ((1 << kTexProgShift) - 1)) + (prog >> kTexProgShift) - 1;
    for (;;) {
        const TextureRec<R>& tx = sc.textures[t];
        if (tx.type == RT_TEX_CHECKER) {  // checkerboard.rs:34-44
            uint32_t iu = as_u32_sat(u * R(2) / tx.scale);
            uint32_t iv = as_u32_sat(v * R(2) / tx.scale);
            t = ((iu + iv) % 2u) == 0u ? tx.aux : int32_t(tx.data);
        } else if (tx.type == RT_TEX_CHECKER_SOLID) {  // checkerboard.rs:74-85
            int32_t ix = as_i32_sat(floor(p.x / tx.scale));
            int32_t iy = as_i32_sat(floor(p.y / tx.scale));
            int32_t iz = as_i32_sat(floor(p.z / tx.scale));
            int32_t sum = int32_t(uint32_t(ix) + uint32_t(iy) + uint32_t(iz));
            t = (sum % 2) == 0 ? tx.aux : int32_t(tx.data);
        } else if (tx.type == RT_TEX_UV_DEBUG) {
            return mk<R>(u, v, R(0.5));  // uv_debug.rs:11-13
        } else {
            return ld3(tx.v);  // constant.rs:30
        }
    }
}

// Runs a postfix texture program (rt_scene.h).  The top four values of the stack live in registers (pushes and pops are
// register moves, no indexed array): enough for every texture the reference's scenes build.  Deeper expressions - the
// reference's recursion (interpolate.rs:29) has no limit - spill the values below the top four into a private array
// (scratch memory; kTexStackMax live values in all), touched only by programs that need it.  Floats travel in .x.
template <typename R> RT_DEV V3<R> eval_texture(const SceneView<R>& sc, int32_t prog, R u, R v, V3<R> p) {
    const int32_t first = prog & ((1 << kTexProgShift) - 1), count = prog >> kTexProgShift;
    V3<R> s0 = mk<R>(0, 0, 0), s1 = s0, s2 = s0, s3 = s0;  // s0 = top
    V3<R> deep[kTexStackMax - kTexStackDepth];                // deep[k] = the value k + 5 from the top... kept in push order
    int live = 0;                                              // values on the stack
#pragma clang loop unroll(disable)
    for (int32_t k = 0; k < count; k++) {
        const TextureRec<R>& tx = sc.textures[first + k];
        switch (tx.type) {
            case RT_TEX_CHECKER: {  // checkerboard.rs:34-44; s1 = even, s0 = odd
                uint32_t iu = as_u32_sat(u * R(2) / tx.scale);
                uint32_t iv = as_u32_sat(v * R(2) / tx.scale);
                s0 = ((iu + iv) % 2u) == 0u ? s1 : s0;
                s1 = s2;
                s2 = s3;
                if (live > 4) s3 = deep[live - 5];
                live -= 1;
                break;
            }
            case RT_TEX_CHECKER_SOLID: {  // checkerboard.rs:74-85
                int32_t ix = as_i32_sat(floor(p.x / tx.scale));
                int32_t iy = as_i32_sat(floor(p.y / tx.scale));
                int32_t iz = as_i32_sat(floor(p.z / tx.scale));
                int32_t sum = int32_t(uint32_t(ix) + uint32_t(iy) + uint32_t(iz));
                s0 = (sum % 2) == 0 ? s1 : s0;
                s1 = s2;
                s2 = s3;
                if (live > 4) s3 = deep[live - 5];
                live -= 1;
                break;
            }
            case RT_TEX_LERP: {  // interpolate.rs:29-39; s2 = start, s1 = end, s0 = t
                const R t = s0.x;
                V3<R> r;
                if (t == R(0)) r = s2;
                else if (t == R(1)) r = s1;
                else r = s2 * (R(1) - t) + s1 * t;
                s0 = r;
                s1 = s3;
                if (live > 4) s2 = deep[live - 5];
                if (live > 5) s3 = deep[live - 6];
                live -= 2;
                break;
            }
            case RT_TEX_CHANNEL:  // channel.rs:22-25 (index 3 = the colour's w, always 0)
                s0 = mk<R>(tx.aux == 0 ? s0.x : (tx.aux == 1 ? s0.y : (tx.aux == 2 ? s0.z : R(0))), R(0), R(0));
                break;
            default: {  // leaves push
                V3<R> val;
                if (tx.type == RT_TEX_UV_DEBUG) val = mk<R>(u, v, R(0.5));  // uv_debug.rs:11-13
                else if (tx.type == RT_TEX_IMAGE) val = image_sample(sc, tx, u, v);
                else if (tx.type == RT_TEX_NOISE_SOLID) val = mk<R>(noise_solid_sample(sc, tx, p), R(0), R(0));
                else val = ld3(tx.v);  // constant.rs:30 (a float constant keeps its value in v[0])
                if (live >= 4) deep[live - 4] = s3;
                s3 = s2;
                s2 = s1;
                s1 = s0;
                s0 = val;
                live += 1;
                break;
            }
        }
    }
    return s0;
}

template <typename R>
struct HitInfo {
    V3<R> pos, normal;
    R u, v;
    V3<R> tex_a;  // the material's colour texture (albedo / emission) sampled at the hit
    R tex_b;      // its float texture (roughness)
    int32_t material;
    bool front_face;
};

// TEX: the scene uses lerp / image / noise / channel textures or normal maps (CompiledScene::needs_tex_interpreter):
// only then is the tangent frame built and the texture interpreter compiled in (it costs ~100 VGPRs in f64).
template <typename R, bool TEX>
RT_DEV HitInfo<R> resolve_hit(const SceneView<R>& sc, const Ray<R>& wray, const Best<R>& best) {
    const Op op = sc.ops[best.pc];
    const Ray<R> ray = ray_in_chain(sc, wray, op.chain);
    HitInfo<R> h;
    V3<R> outward;
    // object-space tangent frame (HitRecord::tangent / bitangent, object.rs:36-37): only normal-mapped
    // materials read it, and no Transform touches it on the way up (transform.rs:132-133)
    V3<R> tangent = mk<R>(1, 0, 0), bitangent = mk<R>(1, 0, 0);
    h.u = R(0);
    h.v = R(0);
    switch (op.type) {
        case OP_SPHERE: {  // sphere.rs:64-93
            const SpherePrim<R>& s = sc.spheres[op.arg];
            h.pos = ray_at(ray, best.t);
            outward = (h.pos - ld3(s.center)) / s.radius;
            h.material = s.material;
            if (sc.materials[s.material].needs_uv) {
                R theta = uv_acos(outward.y);
                R phi = uv_atan2(-outward.z, outward.x) + pi<R>();
                if constexpr (TEX) {
                    tangent = mk<R>(-outward.z, R(0), -outward.x);  // sphere.rs:83-84
                    bitangent = cross(outward, tangent);
                }
                h.u = phi / (R(2) * pi<R>());
                h.v = theta / pi<R>();
            }
            break;
        }
        case OP_PLANE: {  // plane.rs:81-100
            const PlanePrim<R>& p = sc.planes[op.arg];
            h.pos = ray_at(ray, best.t);
            outward = ld3(p.normal);
            h.material = p.material;
            h.u = best.u;
            h.v = best.v;
            if (TEX && sc.materials[p.material].has_normal_map) {  // plane.rs:96-97
                tangent = to_unit(ld3(p.u));
                bitangent = to_unit(ld3(p.v));
            }
            break;
        }
        case OP_MESH: {  // mesh.rs:103-162
            const MeshInst& mi = sc.meshes[op.arg];
            const TriAttr<R>& at = sc.attrs[best.tri];
            h.pos = ray_at(ray, best.t);
            R w = R(1) - best.u - best.v;
            if (mi.flags & RT_MESH_FLAT_SHADING) {
                const TriRec<R>& tr = sc.tris[best.tri];
                outward = to_unit(cross(ld3(tr.e1), ld3(tr.e2)));
            } else {
                outward = ld3(at.n0) * w + ld3(at.n1) * best.u + ld3(at.n2) * best.v;  // not normalised (SURVEY B-4)
            }
            if (at.has_uv) {
                if (TEX && sc.materials[mi.material].has_normal_map) {  // mesh.rs:127-145
                    const TriRec<R>& tr = sc.tris[best.tri];
                    V3<R> edge1 = ld3(tr.e1), edge2 = ld3(tr.e2);
                    R du1 = at.uv1[0] - at.uv0[0], dv1 = at.uv1[1] - at.uv0[1];
                    R du2 = at.uv2[0] - at.uv0[0], dv2 = at.uv2[1] - at.uv0[1];
                    V3<R> edge1perp = cross(outward, edge1);
                    V3<R> edge2perp = cross(edge2, outward);
                    tangent = edge2perp * du1 + edge1perp * du2;
                    bitangent = edge2perp * dv1 + edge1perp * dv2;
                    R inv_max = R(1) / sqrt(fmax(length_squared(tangent), length_squared(bitangent)));
                    tangent = tangent * -inv_max;
                    bitangent = bitangent * inv_max;
                }
                h.u = at.uv0[0] * w + at.uv1[0] * best.u + at.uv2[0] * best.v;
                h.v = at.uv0[1] * w + at.uv1[1] * best.u + at.uv2[1] * best.v;
            }
            h.material = mi.material;
            break;
        }
        case OP_SKY: {  // sky.rs:35-51
            h.pos = ray_at(ray, Lim<R>::inf());
            V3<R> unit_dir = to_unit(ray.d);
            outward = -unit_dir;
            h.material = op.arg;
            if (sc.materials[op.arg].needs_uv) {
                h.u = uv_atan2(unit_dir.x, unit_dir.z) / (R(2) * pi<R>()) + R(0.5);
                h.v = dot(unit_dir, mk<R>(0, 1, 0)) / R(2) + R(0.5);
            }
            break;
        }
        case OP_VOL_END: {  // volume.rs:55-66: position on the ray, everything else arbitrary
            h.pos = ray_at(ray, best.t);
            outward = mk<R>(1, 0, 0);
            h.material = sc.volumes[op.arg].material;
            break;
        }
        default: {  // OP_SUN, sun.rs:45-60
            const SunPrim<R>& s = sc.suns[op.arg];
            h.pos = ray_at(ray, Lim<R>::max());
            outward = -to_unit(ray.d);
            h.material = s.material;
            break;
        }
    }
    h.front_face = dot(ray.d, outward) < R(0);  // object.rs:55, decided in object space
    h.normal = h.front_face ? outward : -outward;
    // Transform::test on the way back up (transform.rs:132-133), innermost first
    int32_t b = sc.chain_offsets[op.chain], e = sc.chain_offsets[op.chain + 1];
    for (int32_t i = e - 1; i >= b; i--) {
        const Xform<R>& x = sc.xforms[sc.chain_items[i]];
        h.pos = xform_apply(x.m, h.pos, R(1));
        h.normal = to_unit(xform_apply(x.m, h.normal, R(0)));
    }
    // The material's textures are sampled HERE, once, in one non-unrolled loop: a single inlined copy of the
    // texture interpreter (image fetch, Perlin turbulence) instead of one per use in shade().  Samplers are
    // pure functions of (u, v, p), so sampling a texture the material then does not read changes nothing.
    //   slot 0: normal map -> Glossy / NormalDebug use the mapped normal wherever they use the normal
    //           (glossy.rs:35-50, normal_debug.rs:23-39): columns (tangent, bitangent, normal) x (sample - 0.5)
    //   slot 1: tex_a (albedo / emission), slot 2: tex_b (roughness)
    const MaterialRec& mat = sc.materials[h.material];
    h.tex_a = mk<R>(0, 0, 0);
    h.tex_b = R(0);
    if constexpr (!TEX) {
        if (mat.tex_a >= 0) h.tex_a = eval_texture_simple(sc, mat.tex_a, h.u, h.v, h.pos);
        if (mat.tex_b >= 0) h.tex_b = eval_texture_simple(sc, mat.tex_b, h.u, h.v, h.pos).x;
    } else {
#pragma clang loop unroll(disable)
        for (int slot = 0; slot < 3; slot++) {
            const int32_t prog = slot == 0 ? (mat.has_normal_map ? mat.tex_c : -1) : (slot == 1 ? mat.tex_a : mat.tex_b);
            if (prog < 0) continue;
            const V3<R> val = eval_texture(sc, prog, h.u, h.v, h.pos);
            if (slot == 0) {
                V3<R> smp = val - mk<R>(R(0.5), R(0.5), R(0.5));
                V3<R> m = mk<R>(tangent.x * smp.x + bitangent.x * smp.y + h.normal.x * smp.z,
                                tangent.y * smp.x + bitangent.y * smp.y + h.normal.y * smp.z,
                                tangent.z * smp.x + bitangent.z * smp.y + h.normal.z * smp.z);
                h.normal = to_unit(m);
            } else if (slot == 1) {
                h.tex_a = val;
            } else {
                h.tex_b = val.x;
            }
        }
    }
    return h;
}

// ------------------------------------------------------------------ lights (pdf/hittable.rs + Hit::pdf_value / random)
// plane.rs:107-118
template <typename R, bool STATS> RT_DEV R plane_pdf_value(const PlanePrim<R>& p, V3<R> origin, V3<R> dir, LaneCounters& cnt) {
    Ray<R> ray = make_ray(origin, dir);
    R t, u, v;
    if (STATS) cnt.prim_tests++;
    if (plane_test(p, ray, R(0.001), Lim<R>::inf(), t, u, v)) {
        V3<R> n = ld3(p.normal);
        bool front = dot(dir, n) < R(0);
        V3<R> hn = front ? n : -n;  // hit.normal() is the face-forwarded normal
        R dist_squared = t * t * length_squared(dir);
        R cosine = fabs(dot(dir, hn) / length(dir));
        return dist_squared / (cosine * p.area);
    }
    return R(0);
}
// sphere.rs:106-121
template <typename R, bool STATS> RT_DEV R sphere_pdf_value(const SpherePrim<R>& s, V3<R> origin, V3<R> dir, LaneCounters& cnt) {
    Ray<R> ray = make_ray(origin, dir);
    R t;
    if (STATS) cnt.prim_tests++;
    if (sphere_test(s, ray, R(0.001), Lim<R>::inf(), t)) {
        R radius_squared = s.radius * s.radius;
        R cos_theta_max = sqrt(R(1) - radius_squared / length_squared(ld3(s.center) - origin));
        R solid_angle = R(2) * pi<R>() * (R(1) - cos_theta_max);
        return R(1) / solid_angle;
    }
    return R(0);
}
template <typename R, bool STATS> RT_DEV R light_pdf_value(const SceneView<R>& sc, const LightRec& l, V3<R> origin, V3<R> dir, LaneCounters& cnt) {
    switch (l.kind) {
        case LIGHT_PLANE: return plane_pdf_value<R, STATS>(sc.planes[l.index], origin, dir, cnt);
        case LIGHT_SPHERE: return sphere_pdf_value<R, STATS>(sc.spheres[l.index], origin, dir, cnt);
        case LIGHT_SKY: return R(1) / (R(4) * pi<R>());  // sky.rs:61-63
        case LIGHT_SUN: return R(1);                      // sun.rs:70-72
        default: return R(0);                             // Transform / mesh / bvh / volume
    }
}
// lights.pdf_value(origin, dir): ObjectList (list.rs:80-89: weight = 1 / n, sum of weight * member.pdf_value in member order,
// starting from 0) or the single object.  FULL kernel variants: members may be ObjectLists themselves, to any depth the scene
// compiler accepts (kMaxLightDepth); the reference's recursion becomes an explicit stack of (list, next member, partial sum),
// so that every partial sum is formed in the reference's order.
template <typename R, bool STATS, bool FULL> RT_DEV R lights_pdf_value(const SceneView<R>& sc, V3<R> origin, V3<R> dir, LaneCounters& cnt) {
    if (!sc.lights_is_list) return light_pdf_value<R, STATS>(sc, sc.lights[0], origin, dir, cnt);
    if constexpr (!FULL) {
        const R weight = sc.inv_n_lights;
        R sum = R(0);
        for (int32_t i = 0; i < sc.n_lights; i++) sum += weight * light_pdf_value<R, STATS>(sc, sc.lights[i], origin, dir, cnt);
        return sum;
    } else {
        int32_t first[kMaxLightDepth], count[kMaxLightDepth], next[kMaxLightDepth];
        R sum[kMaxLightDepth], weight[kMaxLightDepth];
        int depth = 0;
        first[0] = 0; count[0] = sc.n_lights; next[0] = 0; sum[0] = R(0); weight[0] = sc.inv_n_lights;
        for (;;) {
            if (next[depth] == count[depth]) {  // this list is summed up: hand the value to its parent
                if (depth == 0) return sum[0];
                const R value = sum[depth];
                depth--;
                sum[depth] += weight[depth] * value;
                next[depth]++;
                continue;
            }
            const LightRec l = sc.lights[first[depth] + next[depth]];
            if (l.kind == LIGHT_LIST && depth + 1 < kMaxLightDepth) {
                depth++;
                first[depth] = l.index & ((1 << kLightListShift) - 1);
                count[depth] = l.index >> kLightListShift;
                next[depth] = 0;
                sum[depth] = R(0);
                weight[depth] = R(1) / R(count[depth]);
                continue;
            }
            sum[depth] += weight[depth] * light_pdf_value<R, STATS>(sc, l, origin, dir, cnt);
            next[depth]++;
        }
    }
}
template <typename R> RT_DEV V3<R> light_random(const SceneView<R>& sc, const LightRec& l, V3<R> origin, Rng& rng) {
    switch (l.kind) {
        case LIGHT_PLANE: {  // plane.rs:120-126: corner + u*U + v*V covers one quarter (SURVEY B-1)
            const PlanePrim<R>& p = sc.planes[l.index];
            R ru = rng_uniform<R>(rng);
            R rv = rng_uniform<R>(rng);
            V3<R> pt = ld3(p.corner) + ld3(p.u) * ru + ld3(p.v) * rv;
            return pt - origin;
        }
        case LIGHT_SPHERE: {  // sphere.rs:123-145
            const SpherePrim<R>& s = sc.spheres[l.index];
            V3<R> dir = ld3(s.center) - origin;
            V3<R> bu, bv;
            onb_from_vec(dir, bu, bv);
            R radius_squared = s.radius * s.radius;
            R cos_theta_max = sqrt(R(1) - radius_squared / length_squared(dir));
            R r1 = rng_uniform<R>(rng);
            R r2 = rng_uniform<R>(rng);
            R phi = r1 * R(2) * pi<R>();
            R z = R(1) + r2 * (cos_theta_max - R(1));
            R sn, cs;
            sincos_r(phi, sn, cs);
            R x = cs * sqrt(R(1) - z * z);
            R y = sn * sqrt(R(1) - z * z);
            return basis_apply(bu, bv, dir, mk<R>(x, y, z));
        }
        case LIGHT_SKY: return random_unit<R>(rng);           // sky.rs:65-67
        case LIGHT_SUN: return ld3(sc.suns[l.index].direction);  // sun.rs:74-76
        default: return mk<R>(1, 0, 0);
    }
}
template <typename R, bool FULL> RT_DEV V3<R> lights_random(const SceneView<R>& sc, V3<R> origin, Rng& rng) {
    if (!sc.lights_is_list) return light_random(sc, sc.lights[0], origin, rng);
    if (sc.n_lights == 0) return mk<R>(1, 0, 0);  // list.rs:93-95
    uint32_t idx = rng.below(uint32_t(sc.n_lights));
    LightRec l = sc.lights[idx];
    if constexpr (FULL) {
        while (l.kind == LIGHT_LIST) {  // the member is an ObjectList: list.rs:91-100 again, as deep as the lists nest
            const int32_t first = l.index & ((1 << kLightListShift) - 1), count = l.index >> kLightListShift;
            if (count == 0) return mk<R>(1, 0, 0);
            l = sc.lights[first + int32_t(rng.below(uint32_t(count)))];
        }
    }
    return light_random(sc, l, origin, rng);
}

// ------------------------------------------------------------------ camera.rs:260-280, 334-349
template <typename R>
RT_DEV void camera_ray(const CameraView<R>& cam, uint32_t px, uint32_t py, uint32_t sx, uint32_t sy, Rng& rng, V3<R>& origin_out, V3<R>& dir_out) {
    V3<R> pdu = ld3(cam.pdu), pdv = ld3(cam.pdv);
    V3<R> pixel_center = ld3(cam.first_pixel) + (pdu * R(px)) + (pdv * R(py));
    R rx = rng_uniform<R>(rng);
    R ry = rng_uniform<R>(rng);
    R x = (R(sx) + rx) * cam.inv_sqrt_spt - R(0.5);
    R y = (R(sy) + ry) * cam.inv_sqrt_spt - R(0.5);
    V3<R> pixel_sample = pixel_center + (pdu * x + pdv * y);
    V3<R> origin = ld3(cam.position);
    if (cam.has_aperture) {
        // random_in_unit_disk normalises a 2-D Gaussian: samples lie ON the unit circle (SURVEY B-2)
        R dx = rng_normal<R>(rng);
        R dy = rng_normal<R>(rng);
        R len = sqrt(dx * dx + dy * dy + R(0) * R(0));
        dx = dx / len;
        dy = dy / len;
        origin = origin + (ld3(cam.basis_u) * dx + ld3(cam.basis_v) * dy) * cam.aperture_radius;
    }
    origin_out = origin;
    dir_out = pixel_sample - origin;
}
template <typename R>
RT_DEV Ray<R> get_ray(const CameraView<R>& cam, uint32_t px, uint32_t py, uint32_t sx, uint32_t sy, Rng& rng) {
    V3<R> o, d;
    camera_ray(cam, px, py, sx, sy, rng, o, d);
    return make_ray(o, d);
}

// ------------------------------------------------------------------ one path vertex (camera.rs:282-332)
// The reference's recursion returns, for one camera sample, a single PRODUCT: every vertex either ends the path with a
// terminal value T (emission camera.rs:327, background :331, black for Absorbed :326 and for depth == 0 :290) or
// multiplies what its continuation returns by a weight (att * s_pdf / pdf, :312, or att, :320); `from_emission` of a
// scattering material is the zero vector (material.rs:42-44).  The iterative form carries the running product
// `throughput` W forward and sets  L = W * T  at the terminal.  IEEE special values come out like the reference's
// nested evaluation: a factor that is NaN, or a zero factor together with an infinite one, gives NaN in either order
// (e.g. the 0/0 weight of a light sample that misses every light and points below the surface).
template <typename R>
struct PathState {
    Ray<R> ray;
    V3<R> throughput;
    V3<R> radiance;  // written once, at the terminal
    uint32_t depth;  // remaining depth, like the `depth` argument of ray_color
};

// Early ends that cannot change the sample's value.  (1) W is NaN in every channel: so is W * T whatever follows.
// (2) W is zero in every channel and the scene is one whose vertices cannot produce an infinite or NaN weight
// (SceneView::stop_on_zero_weight, decided by the scene compiler): the continuation returns something finite and
// L = 0.  Every other path is traced to its end like the reference does.
template <typename R> RT_DEV bool path_goes_on(const SceneView<R>& sc, PathState<R>& ps) {
    const V3<R> w = ps.throughput;
    if (w.x != w.x && w.y != w.y && w.z != w.z) {
        ps.radiance = w;
        return false;
    }
    if (sc.stop_on_zero_weight && w.x == R(0) && w.y == R(0) && w.z == R(0)) {
        ps.radiance = w;
        return false;
    }
    return true;
}
// Black terminal (Absorbed, depth exhausted): W * 0 is 0 unless W holds an infinity or a NaN.
template <typename R> RT_DEV void end_black(PathState<R>& ps) { ps.radiance = ps.throughput * R(0); }

// Shades the closest hit; returns true if the path continues with ps.ray updated.
template <typename R, bool STATS, bool FULL>
RT_DEV bool shade_hit(const SceneView<R>& sc, const ParamsView<R>& prm, PathState<R>& ps, const HitInfo<R>& hit, Rng& rng, LaneCounters& cnt);

template <typename R, bool STATS, bool TEX = false>
RT_DEV bool shade(const SceneView<R>& sc, const ParamsView<R>& prm, PathState<R>& ps, const Best<R>& best, Rng& rng, LaneCounters& cnt) {
    if (best.pc < 0) {  // camera.rs:331 background
        ps.radiance = ps.throughput * ld3(prm.background);
        return false;
    }
    const HitInfo<R> hit = resolve_hit<R, TEX>(sc, ps.ray, best);
    return shade_hit<R, STATS, TEX>(sc, prm, ps, hit, rng, cnt);
}

// The part of shade() after the hit has been resolved (k_wf_shade calls the two halves itself, so that the
// path's throughput / RNG are loaded only after resolve_hit: they are not live across its loops).
// FULL: the full-feature kernel variant (TEX): additionally evaluates ObjectLists nested inside `lights`.
//
// Every continuing case only produces the new DIRECTION; the new ray (origin = hit.pos for every material) is
// built once, behind the switch.  This is deliberate: with `ps.ray = make_ray(hit.pos, dir)` written inside each
// case, hipcc (ROCm 7.2, clang 22) produced a k_wf_shade in which the Dielectric lanes that reflect by the
// Schlick coin (not total internal reflection) keep two temporaries of the reflectance / RNG code in the
// registers of origin.x / origin.y: after structurisation the join behind the refract block takes those
// registers from an undefined value on the edge that skips it (profiles/r02/shade_dielectric_isa_excerpt.s,
// found with the pool trace RT_WF_TRACE).  The bit-exact wavefront-vs-megakernel tests guard this.
template <typename R, bool STATS, bool FULL>
RT_DEV bool shade_hit(const SceneView<R>& sc, const ParamsView<R>& prm, PathState<R>& ps, const HitInfo<R>& hit, Rng& rng, LaneCounters& cnt) {
    const MaterialRec mat = sc.materials[hit.material];
    V3<R> attenuation = mk<R>(0, 0, 0);
    V3<R> pdf_w = mk<R>(0, 0, 0);  // CosinePDF::w (the shading normal), cosine.rs:17-22
    V3<R> dir = mk<R>(0, 0, 0);    // direction of the continuation ray
    bool with_pdf = false;
    bool uniform_pdf = false;
    bool weight_changed = false;   // the throughput was multiplied by something other than (1,1,1)
    switch (mat.type) {
        case RT_MAT_EMISSIVE: {  // emissive.rs:24-34; camera.rs:327
            if (hit.front_face) ps.radiance = ps.throughput * hit.tex_a;
            else end_black(ps);  // emissive.rs:31-33
            return false;
        }
        case RT_MAT_NORMAL_DEBUG: {  // normal_debug.rs:42-48
            ps.radiance = ps.throughput * (hit.normal * R(0.5) + mk<R>(R(0.5), R(0.5), R(0.5)));
            return false;
        }
        case RT_MAT_LAMBERTIAN:  // lambertian.rs:25-33
            attenuation = hit.tex_a;
            pdf_w = hit.normal;
            with_pdf = true;
            break;
        case RT_MAT_ISOTROPIC:  // isotropic.rs:25-33
            attenuation = hit.tex_a;
            with_pdf = true;
            uniform_pdf = true;
            break;
        case RT_MAT_METAL: {  // metal.rs:28-44
            V3<R> reflected = reflect(ps.ray.d, hit.normal);
            dir = fuzzy_reflection(reflected, hit.tex_b, rng);
            if (!(dot(dir, hit.normal) > R(0))) { end_black(ps); return false; }  // Absorbed (camera.rs:326)
            ps.throughput = ps.throughput * hit.tex_a;
            weight_changed = true;
            break;
        }
        case RT_MAT_DIELECTRIC: {  // dielectric.rs:29-54, attenuation (1,1,1)
            const MaterialParams<R>& mp = sc.material_params[hit.material];
            R ior_ratio = hit.front_face ? mp.inv_ior_r : mp.ior;  // dielectric.rs:31
            V3<R> unit_dir = to_unit(ps.ray.d);
            R cos_theta = fmin(R(1), dot(-unit_dir, hit.normal));
            R sin_theta = sqrt(R(1) - cos_theta * cos_theta);
            bool tir = ior_ratio * sin_theta > R(1);
            bool reflected = tir || reflectance_r0(cos_theta, hit.front_face ? mp.r0_front : mp.r0_back) > rng_uniform<R>(rng);  // no draw on TIR
            dir = reflected ? reflect(unit_dir, hit.normal) : refract(unit_dir, hit.normal, ior_ratio);
            break;
        }
        case RT_MAT_GLOSSY: {  // glossy.rs:54-83
            V3<R> normal = hit.normal;
            V3<R> unit_dir = to_unit(ps.ray.d);
            R cos_theta = fmin(R(1), dot(-unit_dir, normal));
            bool specular = reflectance_r0(cos_theta, sc.material_params[hit.material].r0_glossy) > rng_uniform<R>(rng);
            if (specular) {  // attenuation (1,1,1)
                R roughness = hit.tex_b;
                V3<R> reflected = reflect(ps.ray.d, normal);
                dir = fuzzy_reflection(reflected, roughness, rng);
                if (!(dot(dir, normal) > R(0))) { end_black(ps); return false; }  // Absorbed
                break;
            }
            attenuation = hit.tex_a;
            pdf_w = normal;
            with_pdf = true;
            break;
        }
        default:
            end_black(ps);
            return false;
    }
    if (with_pdf) {
        // ScatteredWithPDF: camera.rs:298-315 with MixPDF (mix.rs:23-36)
        if (rng_uniform<R>(rng) < prm.light_bias) {
            dir = lights_random<R, FULL>(sc, hit.pos, rng);
        } else if (uniform_pdf) {
            dir = random_unit<R>(rng);  // uniform.rs:22-24
        } else {
            V3<R> bu, bv;
            onb_from_vec(pdf_w, bu, bv);
            dir = basis_apply(bu, bv, pdf_w, random_cosine<R>(rng));  // cosine.rs:31-33
        }
        R first_val;
        R scattering_pdf;
        if (uniform_pdf) {
            first_val = R(1) / (R(4) * pi<R>());       // uniform.rs:18-20
            scattering_pdf = R(1) / (R(4) * pi<R>());  // isotropic.rs:35-37
        } else {
            V3<R> unit = to_unit(dir);
            // cosine.rs:26-29 takes dot(unit, w) / pi, lambertian.rs:35-43 / glossy.rs:86-95 dot(w, unit) / pi: the products commute
            // and the sums run over x, y, z in both, so it is ONE value and one division
            const R cos_theta = dot(pdf_w, unit);
            const R cos_over_pi = cos_theta / pi<R>();
            first_val = fmax(cos_over_pi, R(0));
            scattering_pdf = cos_theta < R(0) ? R(0) : cos_over_pi;
        }
        R second_val = lights_pdf_value<R, STATS, FULL>(sc, hit.pos, dir, cnt);
        R pdf = first_val * (R(1) - prm.light_bias) + second_val * prm.light_bias;
        // (scatter_color * attenuation * scattering_pdf) / pdf, camera.rs:312
        V3<R> w = (attenuation * scattering_pdf) / pdf;
        ps.throughput = ps.throughput * w;
        weight_changed = true;
    }
    ps.ray = make_ray(hit.pos, dir);  // the ONE place where the continuation ray is built (see above)
    return weight_changed ? path_goes_on(sc, ps) : true;
}

}  // namespace rt
