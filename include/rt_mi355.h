/*
 * rt_mi355.h — C ABI of the MI355X-native render path (librt_mi355.so).
 *
 * Drop-in boundary: this library replaces ONE call of the reference,
 *
 *     camera.render(world, lights, &mut buf)          reference src/main.rs:75
 *     pub fn render(self, world: Arc<dyn Hit>,
 *                   lights: Arc<dyn Hit>, buf: &mut Buffer)   src/camera.rs:189
 *
 * i.e. the per-pixel stratified sample loop, closest-hit scene evaluation,
 * material scattering and the light-biased mixture-PDF sampler.  Everything the
 * reference passes as Rust trait objects (`Arc<dyn Hit>`, `Arc<dyn Material>`,
 * `Arc<dyn Sampler>`) is handed over as the flat tables below: the tree of
 * `Hit` nodes is kept as a tree (node table + child-index table), so a host
 * only has to walk its own object graph once and copy numbers.  The library
 * compiles that tree into its own device layout (scene program, SAH BVHs).
 *
 * Plain C: pointers and sizes only, no C++/torch/HIP types in signatures
 * (the optional stream argument is an opaque `void*` = hipStream_t).
 * All matrices are row-major 4x4 (reference src/mat4.rs:10), all reals are
 * f64 like the reference (src/vec4.rs:10).  Errors: every entry point returns
 * RT_OK (0) or a negative RtStatus; rt_last_error() gives the message for the
 * calling thread.  Nothing throws or aborts across this boundary (the
 * reference panics: Cargo.toml:18 `panic = "abort"`, src/camera.rs:244).
 */
#ifndef RT_MI355_H
#define RT_MI355_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_MI355_ABI_VERSION 2  /* 2: RtRenderStats grew (per-kernel times, state bytes, iterations, replica groups) */

typedef enum RtStatus {
    RT_OK = 0,
    RT_E_INVALID = -1,      /* malformed description (bad index, NULL, size)  */
    RT_E_UNSUPPORTED = -2,  /* valid reference scene feature the kernels lack */
    RT_E_DEVICE = -3,       /* HIP runtime failure / no gfx950 device         */
    RT_E_NOMEM = -4
} RtStatus;

/* ---- Hit tree ----------------------------------------------------------- */
/* One entry per reference `dyn Hit` object (src/object.rs:107-115).         */
typedef enum RtNodeType {
    RT_NODE_SPHERE = 1,     /* src/object/sphere.rs     p = center[3], radius            */
    RT_NODE_PLANE = 2,      /* src/object/plane.rs      p = center[3], u[3], v[3] (half-vectors) */
    RT_NODE_MESH = 3,       /* src/object/mesh.rs       mesh = index into meshes         */
    RT_NODE_LIST = 4,       /* src/object/list.rs       children = members, in order     */
    RT_NODE_TRANSFORM = 5,  /* src/object/transform.rs  1 child, transform = index       */
    RT_NODE_BVH = 6,        /* src/object/bvh.rs        2 children (child 1 may be NULL node) */
    RT_NODE_SKY = 7,        /* src/object/sky.rs        material = embedded Emissive     */
    RT_NODE_SUN = 8,        /* src/object/sun.rs        p = direction[3] (normalised by callee) */
    RT_NODE_VOLUME = 9,     /* src/object/volume.rs     1 child = boundary, p[0] = density */
    RT_NODE_NULL = 10       /* src/object/bvh/null_obj.rs                                */
} RtNodeType;

#define RT_PLANE_RENDER_BACKFACE 1u      /* Plane::render_backface, plane.rs:17  */
#define RT_LIST_DISABLE_BOUNDS_CHECK 1u  /* ObjectList::disable_bounds_check, list.rs:23 */

typedef struct RtNode {
    uint32_t type;          /* RtNodeType */
    uint32_t flags;
    int32_t  material;      /* index into materials, -1 if none */
    int32_t  mesh;          /* RT_NODE_MESH: index into meshes */
    int32_t  transform;     /* RT_NODE_TRANSFORM: index into transforms */
    uint32_t first_child;   /* offset into child_indices */
    uint32_t n_children;
    uint32_t _pad;
    double   bounds[6];     /* Hit::get_bounding_box(): min xyz, max xyz (object.rs:110) */
    double   p[12];         /* per-type parameters, see RtNodeType */
} RtNode;

/* Transform::transform / inv_transform after all ops (transform.rs:14-20).  */
typedef struct RtTransform {
    double m[16];
    double inv[16];
} RtTransform;

/* TriangleMesh (mesh.rs:22-35) as loaded by loaders/obj.rs: indexed arrays. */
#define RT_MESH_FLAT_SHADING 1u
#define RT_MESH_HIT_BACK_FACES 2u
typedef struct RtMesh {
    const double*   positions;   /* n_positions * 3 */
    const double*   normals;     /* n_normals * 3 (unit length, obj.rs:48)   */
    const double*   uvs;         /* n_uvs * 3 (u, v, w), may be NULL         */
    const uint32_t* tri_pos;     /* n_triangles * 3 indices into positions   */
    const uint32_t* tri_nrm;     /* n_triangles * 3 indices into normals     */
    const int32_t*  tri_uv;      /* n_triangles * 3 indices into uvs, or -1 (Triangle::uv_indices None); may be NULL */
    uint32_t n_positions, n_normals, n_uvs, n_triangles;
    uint32_t flags;
    uint32_t _pad;
} RtMesh;

/* ---- Materials and textures -------------------------------------------- */
typedef enum RtMaterialType {
    RT_MAT_LAMBERTIAN = 1,   /* material/lambertian.rs  tex_a = albedo                 */
    RT_MAT_METAL = 2,        /* material/metal.rs       tex_a = albedo, tex_b = roughness */
    RT_MAT_DIELECTRIC = 3,   /* material/dielectric.rs  ior                            */
    RT_MAT_GLOSSY = 4,       /* material/glossy.rs      tex_a, tex_b, ior, tex_c = normal map or -1 */
    RT_MAT_EMISSIVE = 5,     /* material/emissive.rs    tex_a = emission map           */
    RT_MAT_ISOTROPIC = 6,    /* material/isotropic.rs   tex_a = albedo                 */
    RT_MAT_NORMAL_DEBUG = 7  /* material/normal_debug.rs tex_c = normal map or -1      */
} RtMaterialType;

typedef struct RtMaterial {
    uint32_t type;
    int32_t  tex_a, tex_b, tex_c;
    double   ior;
} RtMaterial;

typedef enum RtTextureType {
    RT_TEX_CONST_COLOR = 1,    /* texture/constant.rs   v = rgb                        */
    RT_TEX_CONST_FLOAT = 2,    /* texture/constant.rs   v[0] = k                       */
    RT_TEX_CHECKER = 3,        /* texture/checkerboard.rs:34  a = even, b = odd, scale */
    RT_TEX_CHECKER_SOLID = 4,  /* texture/checkerboard.rs:74                           */
    RT_TEX_LERP = 5,           /* texture/interpolate.rs  a, b, c = t                  */
    RT_TEX_IMAGE = 6,          /* texture/image.rs:37-53  texels, width, height (repeat, nearest) */
    RT_TEX_NOISE_SOLID = 7,    /* texture/noise.rs:33-38  v = scale vector, samples, perlin_* tables */
    RT_TEX_CHANNEL = 8,        /* texture/channel.rs  a = colour texture, channel      */
    RT_TEX_UV_DEBUG = 9        /* texture/uv_debug.rs                                  */
} RtTextureType;

typedef struct RtTexture {
    uint32_t type;
    int32_t  a, b, c;
    uint32_t channel;
    uint32_t samples;           /* NOISE_SOLID: turbulence octaves (noise.rs:27, default 7)            */
    double   v[3];              /* CONST_*: the value; NOISE_SOLID: NoiseSolidTexture::scale (noise.rs:16) */
    double   scale;
    /* IMAGE: what Buffer::from_image keeps (buffer.rs:30-48): width*height RGB triples as decoded by
     * `into_rgb32f` (8-bit: x/255, 16-bit: x/65535, in f32), row 0 = top row of the file.            */
    const float* texels;
    uint32_t width, height;
    /* NOISE_SOLID: the generator's tables (noise/perlin.rs:13-18). The reference fills them from its
     * entropy-seeded RNG (perlin.rs:21-36); the caller passes its own.                                */
    const double*   perlin_vec;   /* 256 unit vectors, x y z            */
    const uint32_t* perlin_perm;  /* perm_x[256], perm_y[256], perm_z[256], values 0..255 */
} RtTexture;

typedef struct RtSceneDesc {
    uint32_t abi_version;         /* RT_MI355_ABI_VERSION */
    uint32_t n_nodes;
    const RtNode* nodes;
    uint32_t n_child_indices;
    uint32_t n_transforms;
    const uint32_t* child_indices;
    const RtTransform* transforms;
    uint32_t n_meshes;
    uint32_t n_materials;
    const RtMesh* meshes;
    const RtMaterial* materials;
    uint32_t n_textures;
    uint32_t world_root;          /* node index of `world`  (main.rs:75 arg 1) */
    const RtTexture* textures;
    uint32_t lights_root;         /* node index of `lights` (main.rs:75 arg 2) */
    uint32_t flags;               /* RT_SCENE_* */
} RtSceneDesc;

/* RtSceneDesc.flags */
#define RT_SCENE_BVH_ON_DEVICE 1u /* build the mesh BVHs on the GPU (LBVH: milliseconds instead of ~0.7 s per 870k
                                     triangles, slower traversal); default: binned SAH on the host.  The tree only
                                     culls, so the rendered values do not depend on the builder.               */

/* ---- Camera: the fields of `Camera` after init() (camera.rs:19-44,86-130) */
typedef struct RtCameraDesc {
    uint32_t image_width, image_height;
    double position[3];
    double first_pixel[3];
    double pixel_delta_u[3];
    double pixel_delta_v[3];
    double basis_u[3];
    double basis_v[3];
    uint32_t has_aperture;        /* aperture_radius.is_some() */
    uint32_t _pad;
    double aperture_radius;
} RtCameraDesc;

typedef enum RtPrecision { RT_PRECISION_F64 = 0, RT_PRECISION_F32 = 1 } RtPrecision;
typedef enum RtPipeline { RT_PIPELINE_AUTO = 0, RT_PIPELINE_MEGAKERNEL = 1, RT_PIPELINE_WAVEFRONT = 2 } RtPipeline;

/* Render parameters: CameraConfig (config.rs:46-52) + the additions a
 * deterministic, shardable renderer needs (seed, row partition, precision). */
typedef struct RtRenderParams {
    uint32_t sqrt_spt;            /* Camera::sqrt_spt: strata per axis per replica         */
    uint32_t thread_count;        /* Camera::thread_count: number of sample replicas (NOT OS threads) */
    uint32_t max_depth;           /* config.rs:76 default 20                               */
    uint32_t has_background;      /* Camera::background_color.is_some()                    */
    double   light_bias;          /* config.rs:77 default 0.25                             */
    double   background[3];
    uint64_t seed;                /* new: the reference seeds from OS entropy (camera.rs:208) */
    /* Row partition for multi-GPU rendering: rows are grouped in bands of
     * `band_rows`; band b belongs to part (b % n_parts).  band_rows == 0 or
     * n_parts <= 1 means "whole frame".  Output holds only the owned rows,
     * packed in increasing y.                                               */
    uint32_t band_rows;
    uint32_t n_parts;
    uint32_t part;
    uint32_t precision;           /* RtPrecision: arithmetic type of the kernels */
    uint32_t pipeline;            /* RtPipeline */
    uint32_t collect_stats;       /* count node visits / triangle tests / rays (slower) */
} RtRenderParams;

/* Counters and timings of the last rt_render* call on a scene. */
typedef struct RtRenderStats {
    double   kernel_ms;           /* HIP-event time of all render kernels of the call (their own stream) */
    double   traversal_kernel_ms; /* of which: dominant kernel (megakernel, or wavefront intersect)      */
    uint32_t n_launches;          /* launches of the dominant kernel                                     */
    uint32_t pipeline_used;       /* RtPipeline actually run                                             */
    uint64_t samples;             /* W * owned_rows * spp                                                */
    uint64_t rays;                /* world.test() calls (closest-hit casts); valid if collect_stats      */
    uint64_t mesh_rays;           /* casts that entered a mesh BVH                                       */
    uint64_t node_visits;         /* mesh-BVH nodes fetched                                              */
    uint64_t tri_tests;           /* Moller-Trumbore tests                                               */
    uint64_t prim_tests;          /* sphere/quad/sky/sun tests incl. light-pdf re-intersections          */
    uint64_t bytes_node;          /* bytes per BVH node in the layout used                               */
    uint64_t bytes_tri;           /* bytes per triangle record                                           */
    uint64_t bytes_attr;          /* bytes of shading attributes fetched per mesh hit                    */
    uint64_t bytes_state;         /* bytes of path state the traversal kernel moves per ray it handles (0: megakernel) */
    /* wavefront scheduler only (0 otherwise): HIP-event sums per kernel of the iteration loop, and the
     * algorithmic path-state bytes (read + written) per ray that passes through each of them              */
    double   prims_kernel_ms;     /* k_wf_prims: scene program over spheres / quads / sky / sun            */
    double   shade_kernel_ms;     /* k_wf_shade: scatter, pdf, regeneration, queue compaction              */
    uint64_t bytes_state_prims;
    uint64_t bytes_state_shade;
    uint32_t n_iterations;        /* wavefront iterations (one bounce of every live path each)             */
    uint32_t n_replica_groups;    /* groups the replicas were rendered in (per-sample buffer budget)       */
    uint32_t n_tail_compactions;  /* times the live paths were moved together at the end of a group (k_wf_compact) */
    uint32_t _reserved;
} RtRenderStats;

typedef struct RtScene RtScene;

/* Number of usable gfx950 devices (0 if none; never fails). */
int rt_device_count(void);

/* Deep-copies `desc`, builds the device scene (scene program, per-mesh SAH
 * BVH) and uploads it to device `device`.  Replaces the construction of the
 * `Arc<dyn Hit>` graph as far as the render path is concerned.              */
int rt_scene_create(const RtSceneDesc* desc, int device, RtScene** out);
void rt_scene_destroy(RtScene* scene);

/* Number of rows the partition in `params` assigns to this part. */
uint32_t rt_owned_rows(uint32_t image_height, const RtRenderParams* params);

/* The replacement for Camera::render (camera.rs:189-256).  Synchronous.
 * rgba_out: caller-allocated, owned_rows * width * 4 doubles, row-major,
 * overwritten with the per-pixel mean linear radiance (r, g, b, 0): exactly
 * what the reference leaves in `buf` (camera.rs:229-231, 247-253).          */
int rt_render(const RtScene* scene, const RtCameraDesc* camera,
              const RtRenderParams* params, double* rgba_out);

/* Same, but the output stays in HBM: d_rgba_out is a device pointer on the
 * scene's device (owned_rows * width * 4 doubles); work is enqueued on
 * `stream` (hipStream_t, NULL = the library's own stream) and the call
 * returns after the kernels complete.                                       */
int rt_render_device(const RtScene* scene, const RtCameraDesc* camera,
                     const RtRenderParams* params, double* d_rgba_out, void* stream);

int rt_get_stats(const RtScene* scene, RtRenderStats* out);

/* Diagnostic probe (tests): traces ONE sample (replica tid, pixel x,y, stratum sx,sy) on the
 * device; rgb_out[3] = its radiance, trace_out[17*max_bounces] = per bounce: t, pos xyz,
 * material index, scene-program op type, triangle slot, 0, normal xyz, ray origin xyz, ray dir xyz.  Returns the bounce count (>= 0)
 * or a negative RtStatus. */
int rt_debug_trace_sample(const RtScene* scene, const RtCameraDesc* camera, const RtRenderParams* params,
                          uint32_t tid, uint32_t x, uint32_t y, uint32_t sx, uint32_t sy,
                          double* rgb_out, double* trace_out, uint32_t max_bounces);

/* Diagnostic probe (tests): the specular reflection of Metal / Glossy (metal.rs:33-35, glossy.rs:66-68) as the kernels compute
 * it - with the shortcut for a fuzz / roughness of exactly 0 - next to the plain expression, on `n` inputs (reflected[3 n],
 * fuzz[n], generator state[n]): out[6 i ..] = the kernels' direction, then the plain one; state_out[2 i ..] = the generator
 * after each.  The two must agree bit for bit. */
int rt_debug_fuzzy_reflection(int device, uint32_t n, const double* reflected, const double* fuzz, const uint64_t* state, double* out, uint64_t* state_out);

/* Diagnostic (host only, needs no device): compiles `desc` like rt_scene_create and reports how the scene
 * compiler classified it.  RT_SCENE_INFO_ZERO_WEIGHT_STOP: the light set cannot give an infinite or NaN weight,
 * so paths whose weight is exactly 0 are ended early (otherwise they are traced to the end like
 * camera.rs:310-314 does); _TEX_INTERPRETER: full-feature kernel variants; _VOLUMES: combined intersect kernel. */
#define RT_SCENE_INFO_ZERO_WEIGHT_STOP 1u
#define RT_SCENE_INFO_TEX_INTERPRETER 2u
#define RT_SCENE_INFO_VOLUMES 4u
int rt_scene_info(const RtSceneDesc* desc, uint32_t* flags_out);
/* Same, plus mesh statistics of the compiled scene (distinct meshes, host-built BVH):
 * out[0] triangle records, out[1] BVH2 nodes, out[2] 4-wide nodes, out[3] BVH2 depth, out[4] worst-case 4-wide traversal stack,
 * out[5] scene-program ops, out[6] sphere / quad groups re-built as SAH trees, out[7] primitives in them. */
int rt_scene_mesh_stats(const RtSceneDesc* desc, uint64_t out[8]);
/* Same, plus the compiled scene program itself (tests of the scene compiler without a GPU): ops_out[4 * i ..] = type, arg,
 * skip, chain of op i (rt_scene.h OpType; up to `capacity` ops are written, *n_ops_out = the program's length; ops_out may be
 * NULL).  info[0] mesh ops, [1] primitive groups with a 4-wide BVH, [2] their nodes, [3] their worst-case stack, [4] entries of
 * the light table (nested lists flattened to a tree), [5] volumes, [6] the wavefront scheduler's kernel plan: bit 0 split
 * intersect (k_wf_prims + k_wf_mesh), bit 1 volumes inside k_wf_prims, bit 2 multi-mesh form of k_wf_mesh, bit 3 group
 * BVHs in k_wf_prims; [7] primitives in group BVHs. */
int rt_scene_program(const RtSceneDesc* desc, int32_t* ops_out, uint32_t capacity, uint32_t* n_ops_out, uint64_t info[8]);

/* Frame pipelining (no counterpart in the reference, which renders one image per process run): while `flag` is set
 * (non-NULL), every rt_render / rt_render_device of this scene object stores 1 to `*flag` as soon as the render can no
 * longer fill the GPU — all samples started, pool slots running empty (wavefront scheduler, last replica group) — and
 * at the latest when the call returns, with or without an error.  A second RtScene created from the same description,
 * driven from another host thread on another stream, can start the NEXT frame at that moment: its full launches run
 * underneath the first render's tail (a chain of small, latency-bound launches: 10 % of a 1/8-frame share).  The caller
 * clears `*flag` before each render.  rust_raytracer_amd.api.FramePipeline and bench.py use it. */
int rt_scene_set_tail_flag(RtScene* scene, int32_t* flag);

/* Message for the last non-RT_OK status on this thread ("" if none). */
const char* rt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355_H */
