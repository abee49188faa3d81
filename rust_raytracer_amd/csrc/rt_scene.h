// rt_scene.h — device-side scene layout (plain PODs, templated on the arithmetic type R).
//
// The reference keeps the scene as a graph of `Arc<dyn Hit>` trait objects that `test()`
// walks recursively (src/object/*.rs).  On the GPU the same graph is flattened ONCE into
//   * a linear "scene program" (Op[]) that reproduces the reference's depth-first visiting
//     order, its bounds culling (skip pointers) and its closest-hit interval shrinking,
//   * primitive tables (spheres, quads, sky/sun, mesh instances) in R,
//   * one SAH BVH2 per triangle mesh with both child boxes stored in the parent node
//     (replaces the reference's octree, src/object/mesh/octree.rs; same closest hit),
//   * material / texture / light tables.
#pragma once
#include <cstdint>

namespace rt {

enum OpType : int32_t {
    OP_END = 0,
    OP_BOUNDS = 1,      // ObjectList / BVH node bounds test (list.rs:59, bvh.rs:85): arg = bounds idx, skip = pc when culled
    OP_XFORM_PUSH = 2,  // Transform::test entry (transform.rs:124-127): arg = transform idx
    OP_XFORM_POP = 3,   // leaving a Transform: ray goes back to the parent space (chain = parent chain)
    OP_SPHERE = 4,      // arg = sphere idx
    OP_PLANE = 5,       // arg = plane idx
    OP_MESH = 6,        // arg = mesh instance idx
    OP_SKY = 7,         // arg = material idx
    OP_SUN = 8,         // arg = sun idx
    // Volume::test (volume.rs:33-71) = two closest-hit searches over the boundary's sub-program, then the
    // free-flight draw:   VOL_BEGIN  <boundary ops>  VOL_MID  <boundary ops again>  VOL_END
    OP_VOL_BEGIN = 9,   // save the search state; boundary test over Interval::UNIVERSE
    OP_VOL_MID = 10,    // no entry hit: restore, jump to `skip`; else second test over (t_enter + 0.0001, inf)
    OP_VOL_END = 11,    // arg = volume idx: clamp to the caller's interval, draw the scattering distance
    // A re-built primitive group (rt_compile.cpp: a list / object-BVH subtree of spheres and quads) exists in TWO forms:
    // as skip-pointer ops (OP_BOUNDS tree, the form every interpreter understands) and as a 4-wide quantised BVH over the
    // same primitives (SceneView::group_*).  OP_GROUP stands in front of the op form: arg = group idx, skip = pc behind the
    // subtree.  k_wf_prims<GROUPS> searches the BVH (nearest child first, per-lane stack in LDS) and jumps to `skip`; the
    // other interpreters treat the op as a no-op and walk the ops behind it.  Same closest hit either way: both forms test
    // the same primitives with the same exact arithmetic, boxes only cull, ties go by rank (hit_takes_over).
    OP_GROUP = 12,
};

struct Op {
    int32_t type;
    int32_t arg;
    int32_t skip;   // OP_BOUNDS / OP_VOL_MID: pc to continue at when culled; every op that can produce a hit: its RANK (>= 1) in the
                    // reference's depth-first visiting order (ties at equal t go to the lower rank, rt_device.h hit_takes_over)
    int32_t chain;  // index into chain_offsets: transforms enclosing this op, outermost first
};

template <typename R>
struct Bounds {  // reference AxisAlignedBoundingBox, aabb.rs:7
    R lo[3], hi[3];
};

template <typename R>
struct Xform {  // Transform::transform / inv_transform (row-major, rows 0..2; row 3 is 0,0,0,1)
    R m[12];
    R inv[12];
};

template <typename R>
struct SpherePrim {  // sphere.rs:18-24
    R center[3];
    R radius;
    int32_t material;
    int32_t _pad;
};

template <typename R>
struct PlanePrim {  // plane.rs:14-27 (fields as computed by Plane::new, plane.rs:29-63)
    R corner[3], normal[3], u[3], v[3], inv_u[3], inv_v[3];
    R area;
    int32_t material;
    int32_t backface;
};

template <typename R>
struct SunPrim {  // sun.rs:17-20
    R direction[3];
    int32_t material;
    int32_t _pad;
};

template <typename R>
struct VolumeRec {  // volume.rs:15-19
    R neg_inv_density;
    int32_t material;
    int32_t _pad;
};

template <typename R>
struct GroupRec {
    uint32_t root;        // root node in SceneView::group_nodes
    uint32_t _pad[3];
    R lo[3], hi[3];       // box of the group's primitives (exact, rounded outward in f32 builds): the culling ray starts on it
};
struct GroupPrimRef {
    int32_t pc;           // the primitive's op inside the group's op form (type, arg, rank, chain): also the `pc` of its hits
    int32_t guard_first;  // guard boxes (reference ancestor boxes that do not contain the primitive, SURVEY B-8): indices
    int32_t guard_count;  // into SceneView::group_guards -> SceneView::bounds, tested with the reference's Williams test
    int32_t _pad;
};

struct MeshInst {
    uint32_t node_base;  // first BVH node of this mesh in nodes[]
    uint32_t tri_base;   // first triangle record in tris[] / attrs[]
    int32_t material;
    uint32_t flags;      // RT_MESH_* | MESH_HAS_UV
    uint32_t n_tris;
    uint32_t max_depth;
    uint32_t node4_base;  // root of this mesh's 4-wide BVH in nodes4[] / nodes4q[] (their child references are absolute)
    uint32_t _pad;
};
constexpr uint32_t MESH_HAS_UV = 0x100u;

// What k_wf_mesh needs to enter mesh op m of the scene program (SceneView::mesh_op_recs[m], program order), in one record:
// the kernel enters one mesh op at a time for all the lanes that wait for it, so the record is read with scalar loads.
template <typename R>
struct alignas(16) MeshOpRec {
    int32_t pc;           // the OP_MESH op
    int32_t chain;        // its transform chain (Op::chain)
    uint32_t node4_base;  // root of the mesh's 4-wide BVH
    uint32_t flags;       // MeshInst::flags | chain length << 16
    R lo[3], hi[3];       // SceneView::mesh_bounds of the instance
    R inv[12];            // chain length 1 (a mesh inside one Transform, the usual case): that transform's inverse, so that
                          // entering the mesh costs ONE scalar-load round trip instead of record -> chain -> item -> matrix
};


// BVH2 node: the two children's boxes live in the parent, so one fetch decides both.
// child >= 0: inner node index (relative to node_base); child < 0: leaf,
// ~child = (first_tri << 3) | (count - 1); kEmptyChild: no child.
constexpr int32_t kEmptyChild = INT32_MIN;
template <typename R>
struct alignas(sizeof(R) == 8 ? 128 : 64) BvhNode {
    R lo0[3], hi0[3], lo1[3], hi1[3];
    int32_t c0, c1;
};

// 4-wide node with CONSERVATIVE f32 child boxes (structure of arrays over the four children; 128 B =
// one cache line).  Boxes only cull: they are padded by 2^-19 x (largest |coordinate| of the mesh) and
// rounded outward, which covers every rounding error of the f32 slab test for rays whose origin has
// been moved onto the root box (rt_wavefront.h, k_wf_mesh), so no triangle the exact test in R would
// hit is ever skipped.  Used by the wavefront mesh kernel for both arithmetic types.
struct alignas(128) BvhNode4f {
    float lox[4], loy[4], loz[4], hix[4], hiy[4], hiz[4];
    int32_t child[4];  // >= 0 inner node (ABSOLUTE index into nodes4), < 0 leaf: ~child = (first triangle slot, absolute, << 3) | (count - 1); kEmptyChild
    uint32_t _pad[4];
};

// The same node in 64 B (four 16-B loads instead of seven): child boxes as 8-bit grid coordinates relative to a
// per-node origin and per-axis power-of-two cell size (the compressed wide BVH node of Ylitie, Karras, Laine 2017,
// 4-wide here).  A wave whose lanes read 64 different nodes pays one L1 cycle per lane and LOAD INSTRUCTION
// (tools/ubench/micro_l1.hip: 64 clk per 16-B wave load, 449 clk per visit of a 7-load record, 291 of a 4-load one), which is
// what bounds k_wf_mesh.  Decoded plane = org + q * cell, q_lo rounded down and q_hi up on the grid from the boxes
// of BvhNode4f (padding included), so the quantised box contains the padded f32 box; the decode's own rounding
// (one fma per plane) stays inside that padding (see the node builder in rt_kernels.hip).
struct alignas(64) BvhNode4q {
    float org[3];      // grid origin (<= every child's lo)
    float cell[3];     // 2^e per axis
    uint32_t qlo[3];   // per axis: child k's lower grid coordinate in byte k
    uint32_t qhi[3];   // upper; an empty child has qlo = 255 > qhi = 0
    int32_t child[4];  // as in BvhNode4f
};
static_assert(sizeof(BvhNode4q) == 64, "BvhNode4q must be 64 bytes");

// Triangle record for the intersection test: v0 and the two edges (mesh.rs:69-70 computes the
// edges per test; v1 - v0 done once on the host in the same arithmetic gives the same bits).
template <typename R>
struct alignas(16) TriRec {
    R v0[3], e1[3], e2[3];
    R _pad;
};
// Shading attributes, fetched only for the final closest hit.
template <typename R>
struct alignas(16) TriAttr {
    R n0[3], n1[3], n2[3];  // vertex normals (mesh.rs:110-116)
    R uv0[2], uv1[2], uv2[2];
    int32_t has_uv;
    int32_t _pad;
};

struct MaterialRec {  // material/*.rs
    int32_t type;     // RtMaterialType
    int32_t tex_a, tex_b, tex_c;
    int32_t needs_uv; // any reachable texture reads (u, v)
    int32_t has_normal_map;  // tex_c is a normal map (glossy.rs:35-50, normal_debug.rs:23-39): tangent frame needed
};
template <typename R>
struct MaterialParams {
    R ior;      // Dielectric::ior
    R inv_ior;  // Glossy::inv_ior = 1 / ior (glossy.rs:30), computed in f64 then rounded
    // Per-material constants of the Schlick term (utils.rs:31-36: r0 = ((1 - x) / (1 + x))^2 for x = the ior ratio), computed once
    // on the host in R with the kernels' own operation order - the same IEEE operations on the same operands, so the same bits
    // as evaluating them at every vertex (one f64 division less per Glossy vertex, two per Dielectric vertex):
    R inv_ior_r;     // R(1) / ior in R arithmetic: the front-face ratio of Dielectric (dielectric.rs:31)
    R r0_glossy;     // x = inv_ior
    R r0_front;      // x = inv_ior_r
    R r0_back;       // x = ior
};

// Texture expressions are compiled to POSTFIX programs (rt_compile.cpp): a material slot holds
// (first op | op count << 20); operands are evaluated before their operator onto a value stack (top four in registers):
//   CONST_COLOR / CONST_FLOAT / UV_DEBUG / IMAGE / NOISE_SOLID   push a value
//   CHECKER / CHECKER_SOLID   [even][odd] -> the one the parity rule selects (checkerboard.rs:34-44, 74-85)
//   LERP                      [start][end][t] -> interpolate.rs:29-39
//   CHANNEL                   [colour] -> colour[channel] (channel.rs:22-25)
// Every sampler is a pure function of (u, v, p), so evaluating both inputs of a checker / lerp and
// selecting afterwards gives the reference's value.
template <typename R>
struct TextureRec {  // texture/*.rs
    int32_t type;    // RtTextureType
    int32_t aux;     // CHANNEL: channel index; NOISE_SOLID: turbulence samples
    uint32_t data;   // IMAGE: first texel (in RGB triples) in SceneView::texels; NOISE_SOLID: generator index
    uint32_t width, height;  // IMAGE
    int32_t _pad;
    R v[3];          // CONST_*: value; NOISE_SOLID: scale vector
    R scale;         // CHECKER*
};
constexpr int kMaxVolDepth = 2;     // a volume and a volume inside its boundary (VolFrames, rt_device.h: one register frame per level)
constexpr int kTexStackDepth = 4;   // values of eval_texture's stack kept in registers
constexpr int kTexStackMax = 16;    // live values a texture program may need (the rest spill to a private array)
constexpr int32_t kTexProgShift = 20;  // program id = first op | (op count << 20)

enum LightKind : int32_t { LIGHT_OTHER = 0, LIGHT_PLANE = 1, LIGHT_SPHERE = 2, LIGHT_SKY = 3, LIGHT_SUN = 4, LIGHT_LIST = 5 };
struct LightRec {
    int32_t kind;
    int32_t index;  // into planes / spheres / suns; LIGHT_LIST (an ObjectList inside `lights`, e.g. an emissive box):
                    // first member in lights[] | member count << 20 (the members of a list are contiguous; members that are
                    // lists themselves point further back: a tree, as deep as kMaxLightDepth)
};
constexpr int32_t kLightListShift = 20;
constexpr int kMaxLightDepth = 8;  // nesting levels of ObjectLists inside `lights` (explicit evaluation stack in lights_pdf_value)

// All the small tables (everything except BVH nodes, triangle records and attributes) are ALSO uploaded as contiguous
// blobs, so that a workgroup can stage them in LDS with one cooperative copy and walk the scene program / materials /
// lights without global-memory round trips.  Two blobs, the same tables in two orders: what k_wf_prims reads most first
// (ops, chains, transforms, boxes, primitives) and what k_wf_shade reads most first (ops, chains, transforms, materials,
// lights, textures, primitives).  A kernel stages a PREFIX of whole tables (as much as its LDS budget allows: the default
// scene's 455 spheres, each with its own material and texture, make 90 KB of tables) and reads the rest from global memory.
enum SmallTable : int {
    ST_OPS, ST_BOUNDS, ST_CHAIN_OFFSETS, ST_CHAIN_ITEMS, ST_XFORMS, ST_SPHERES, ST_PLANES, ST_SUNS, ST_MESHES, ST_MATERIALS,
    ST_MATERIAL_PARAMS, ST_TEXTURES, ST_LIGHTS, ST_COUNT
};
struct SmallLayout {
    uint32_t begin[ST_COUNT], end[ST_COUNT];  // byte range of every table inside the blob (begin 16-B aligned)
    uint32_t total_bytes;
};

// Device pointers.  Passed to kernels by value.
template <typename R>
struct SceneView {
    const Op* ops;
    const Bounds<R>* bounds;
    const int32_t* chain_offsets;  // chain c = chain_items[chain_offsets[c] .. chain_offsets[c+1])
    const int32_t* chain_items;
    const Xform<R>* xforms;
    const SpherePrim<R>* spheres;
    const PlanePrim<R>* planes;
    const SunPrim<R>* suns;
    const MeshInst* meshes;
    const VolumeRec<R>* volumes;   // global memory (not in the LDS blob)
    const BvhNode<R>* nodes;
    const BvhNode4f* nodes4;       // 4-wide f32 nodes (k_wf_mesh with RT_WF_NODES=0: A/B control)
    const BvhNode4q* nodes4q;      // the same nodes, quantised to 64 B (k_wf_mesh)
    const Bounds<R>* mesh_bounds;  // per mesh instance: exact box of its triangles (object space)
    const int32_t* mesh_ops;       // pcs of the OP_MESH ops in program order: k_wf_prims defers them, k_wf_mesh serves them one after the other
    const MeshOpRec<R>* mesh_op_recs;  // the same ops as k_wf_mesh wants them
    int32_t n_mesh_ops;
    const GroupRec<R>* groups;         // re-built primitive groups (OP_GROUP)
    const BvhNode4q* group_nodes;      // their 4-wide quantised BVHs (absolute child references; leaves index group_prims)
    const GroupPrimRef* group_prims;
    const int32_t* group_guards;
    int32_t n_group_nodes;
    int32_t group_stack_levels;        // worst-case traversal stack of any group (0: no groups)
    const TriRec<R>* tris;
    const TriAttr<R>* attrs;
    const MaterialRec* materials;
    const MaterialParams<R>* material_params;
    const TextureRec<R>* textures;   // postfix ops
    const float* texels;             // all image textures, RGB f32 triples (image.rs / buffer.rs:30-48)
    const R* perlin_vec;             // per generator: 256 x (x, y, z)   (perlin.rs:13-18)
    const uint32_t* perlin_perm;     // per generator: perm_x, perm_y, perm_z (3 x 256)
    const LightRec* lights;
    int32_t n_lights;
    R inv_n_lights;          // R(1) / R(n_lights): the weight of list.rs:81, computed once on the host in R
    int32_t lights_is_list;  // lights root is an ObjectList (list.rs:80-100) vs a single object
    int32_t stop_on_zero_weight;  // CompiledScene::zero_weight_stop: a path whose throughput is exactly 0 may end (rt_device.h, path_goes_on)
    int32_t stack_entries;   // per-lane LDS traversal stack size
    int32_t n_ops;
    const char* small_blob;        // the small tables packed in k_wf_prims' order (global memory)
    SmallLayout lay;
    const char* small_blob_shade;  // ... and in k_wf_shade's order
    SmallLayout lay_shade;
};

template <typename R>
struct CameraView {  // camera.rs:19-44
    R position[3], first_pixel[3], pdu[3], pdv[3], basis_u[3], basis_v[3];
    R aperture_radius;
    R inv_sqrt_spt;
    int32_t has_aperture;
    uint32_t width, height;
    uint32_t sqrt_spt, thread_count, max_depth;
};

template <typename R>
struct ParamsView {
    R light_bias;
    R background[3];
    uint64_t seed;
    uint32_t band_rows, n_parts, part;  // row partition (see RtRenderParams)
    uint32_t owned_rows;
    double inv_spp;  // unused by the math (division is used), kept for diagnostics
    double spp;
};

struct DeviceCounters {
    unsigned long long rays, mesh_rays, node_visits, tri_tests, prim_tests;
    // k_wf_mesh lane utilisation (collect_stats): wave-level iterations of the node / triangle / refill code and
    // the lanes that were active in them (utilisation = lanes / (64 * waves)); printed with RT_WF_DEBUG=1
    unsigned long long node_wave_iters, tri_wave_iters, refill_wave_iters, refill_lanes, pops_culled;
};

}  // namespace rt
