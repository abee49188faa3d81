/*
 * oracle.h — C ABI of the CPU oracle (liboracle.so).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; nothing under rust_raytracer_amd/ links, imports
 * or falls back to it.
 *
 * Parity status: **pinned only statistically**.  The reference (Rust) cannot be built in
 * this image (no cargo/rustc), has no tests or golden vectors, and seeds every RNG from OS
 * entropy (src/camera.rs:208), so bit-level pins do not exist.  The oracle is pinned by
 * (a) hand-derived known-answer values from the cited source lines (tests/test_oracle_kat.py),
 * (b) the three renders the reference ships in samples/ (PNG files), compared after the restated
 * output stage on 8x8 box-filtered blocks (tests/test_oracle_vs_samples.py).
 * The RNG (rand/rand_pcg/rand_distr crates, not vendored) is replaced by the repo's keyed
 * SplitMix64 stream: "parity unpinned" at that boundary, see DESIGN.md.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "../include/rt_mi355.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OracleStats {
    uint64_t rays;          /* world.test() calls from ray_color (camera.rs:294)     */
    uint64_t node_tests;    /* test_bounding_box calls on octree nodes (mesh.rs:166) */
    uint64_t tri_tests;     /* test_tri calls (mesh.rs:62)                           */
    uint64_t prim_tests;    /* sphere / plane / sky / sun tests incl. pdf_value      */
    uint64_t samples;
    double   seconds;       /* wall time of the render loop                          */
    uint32_t os_threads;
    uint32_t _pad;
} OracleStats;

/* Restatement of Camera::render (src/camera.rs:189-256) on the scene description of
 * rt_mi355.h.  One OS thread per params->thread_count replica, like the reference.
 * Output layout and row partition semantics are those of rt_render. */
int oracle_render(const RtSceneDesc* scene, const RtCameraDesc* camera,
                  const RtRenderParams* params, double* rgba_out, OracleStats* stats);

/* Bounding box the oracle computes for node `node` with the reference's rules
 * (used to cross-check the host builder). out6 = min xyz, max xyz. */
int oracle_node_bounds(const RtSceneDesc* scene, uint32_t node, double* out6);

/* Octree statistics of mesh `mesh` built with the reference's rule (octree.rs:31-210):
 * out[0]=branches, [1]=leaves, [2]=empty leaves, [3]=triangle references, [4]=max depth,
 * [5]=max triangles in a leaf. */
int oracle_octree_stats(const RtSceneDesc* scene, uint32_t mesh, uint64_t* out6);

/* Single-function probes for known-answer tests.  Each mirrors one reference function. */
/* aabb.rs:50-87 */
int oracle_test_bounding_box(const double bounds6[6], const double origin[3], const double dir[3],
                             double t_min, double t_max);
/* closest hit of `world` for one ray with Interval(t_min, t_max): returns 1 on hit and fills
 * out[0]=t, [1..3]=pos, [4..6]=normal, [7]=u, [8]=v, [9]=front_face, [10]=material index */
int oracle_world_hit(const RtSceneDesc* scene, const double origin[3], const double dir[3],
                     double t_min, double t_max, double* out11);
/* lights.pdf_value(origin, dir) (list.rs:80-89, plane.rs:107-118, sphere.rs:106-121) */
int oracle_lights_pdf_value(const RtSceneDesc* scene, const double origin[3], const double dir[3], double* out);
/* n draws of lights.random(origin) (list.rs:91-100, plane.rs:120-126, sphere.rs:123-128) from one
 * stream keyed by `seed`: out3n = n direction vectors */
int oracle_lights_random(const RtSceneDesc* scene, const double origin[3], uint64_t seed, uint32_t n, double* out3n);
/* include/rt_detmath.h: out3 = det_sin(x), det_cos(x), det_log(x) */
void oracle_detmath(double x, double* out3);
/* out3 = det_atan(y), det_atan2(y, x), det_acos(y) */
void oracle_detmath_inv(double y, double x, double* out3);
/* utils.rs:31-36 */
/* texture/*.rs Sampler::sample of texture `tex` at (u, v, p) */
int oracle_texture_sample(const RtSceneDesc* scene, uint32_t tex, double u, double v, const double p[3], double* out3);
double oracle_reflectance(double cos_theta, double ior_ratio);
/* utils.rs:17-28: out9 = columns u, v, w */
void oracle_onb_from_vec(const double w[3], double* out9);
/* vec4.rs:140-147 */
void oracle_refract(const double v[3], const double n[3], double ior_ratio, double* out3);
/* The keyed RNG: fills out[n] with the first n uniform draws of stream (seed, tid, pixel, stratum). */
void oracle_rng_uniforms(uint64_t seed, uint32_t tid, uint64_t pixel, uint32_t stratum, uint32_t n, double* out);
void oracle_rng_raw(uint64_t seed, uint32_t tid, uint64_t pixel, uint32_t stratum, uint32_t n, uint64_t* out);
/* Camera::get_ray for sample (x, y, sx, sy) of replica tid: out6 = origin, dir (camera.rs:260-280) */
void oracle_get_ray(const RtCameraDesc* camera, const RtRenderParams* params, uint32_t tid,
                    uint32_t x, uint32_t y, uint32_t sx, uint32_t sy, double* out6);

/* ray_color for ONE sample with a per-bounce trace (17 doubles per bounce: t, pos xyz, material,
 * scatter kind 0 pdf / 1 ray / 2 absorbed / 3 emissive / -1 miss, mix pdf, scattering pdf,
 * normal xyz, ray origin xyz, ray dir xyz).
 * Returns the number of bounces recorded. */
int oracle_trace_sample(const RtSceneDesc* scene, const RtCameraDesc* camera, const RtRenderParams* params,
                        uint32_t tid, uint32_t x, uint32_t y, uint32_t sx, uint32_t sy, double* rgb_out,
                        double* trace_out, uint32_t max_bounces);

const char* oracle_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
