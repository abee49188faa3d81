// rt_compile.h — RtSceneDesc (tree of Hit nodes) -> flat host tables in f64, ready to be
// rounded to the kernel's arithmetic type and uploaded.
#pragma once
#include <string>
#include <vector>

#include "../../include/rt_mi355.h"
#include "rt_bvh.h"
#include "rt_scene.h"

namespace rt {

struct CompiledScene {
    std::vector<Op> ops;
    std::vector<Bounds<double>> bounds;
    std::vector<int32_t> chain_offsets, chain_items;
    std::vector<Xform<double>> xforms;
    std::vector<SpherePrim<double>> spheres;
    std::vector<PlanePrim<double>> planes;
    std::vector<SunPrim<double>> suns;
    std::vector<MeshInst> meshes;
    std::vector<VolumeRec<double>> volumes;  // OP_VOL_* (world_test<VOL> / k_wf_intersect<VOL>)
    std::vector<BuildNode> nodes;            // all meshes, node indices relative to MeshInst::node_base
    std::vector<BuildNode4> nodes4;          // 4-wide collapse, ABSOLUTE child / triangle references; a mesh's root is node MeshInst::node4_base
    std::vector<int32_t> mesh_ops;           // pcs of the OP_MESH ops outside volumes, in program order (k_wf_prims defers them, k_wf_mesh serves them)
    std::vector<Bounds<double>> mesh_bounds; // per MeshInst
    uint32_t max_bvh4_stack = 1;
    std::vector<TriRec<double>> tris;        // leaf order
    std::vector<TriAttr<double>> attrs;      // same order
    std::vector<MaterialRec> materials;
    std::vector<MaterialParams<double>> material_params;
    std::vector<TextureRec<double>> textures;  // postfix ops of every texture program (see rt_scene.h)
    std::vector<float> texels;                 // image textures, concatenated
    std::vector<double> perlin_vec;            // noise generators, 768 doubles each
    std::vector<uint32_t> perlin_perm;         // 768 each
    std::vector<LightRec> lights;
    bool needs_tex_interpreter = false;        // any lerp / image / noise / channel op or normal map
    int32_t lights_is_list = 0;
    int32_t n_top_lights = 0;                  // entries of `lights` itself (members of nested lists follow in `lights`)
    uint32_t max_bvh_depth = 1;
    // A path whose running weight is exactly zero contributes 0 unless a LATER vertex produces an infinite or NaN
    // weight (0 * inf = NaN in the reference's `(L * att * s_pdf) / pdf`, camera.rs:312).  Such weights need pdf == 0
    // (or a NaN direction) at a pdf-sampled vertex, i.e. a light sample that no light's pdf_value covers.  True when
    // the light set rules that out (see Compiler::decide_zero_weight_stop); the kernels then end zero-weight paths,
    // otherwise they trace them to the end like the reference.
    bool zero_weight_stop = false;
    uint32_t n_rebuilt_groups = 0, n_rebuilt_prims = 0;  // object-BVH / list subtrees re-built as SAH trees (rt_compile.cpp)
    // the same groups as 4-wide BVHs for k_wf_prims<GROUPS> (OP_GROUP, rt_scene.h)
    std::vector<GroupRec<double>> groups;
    std::vector<BuildNode4> group_nodes4;   // absolute child references; leaf codes index group_prims
    std::vector<GroupPrimRef> group_prims;
    std::vector<int32_t> group_guards;
    uint32_t max_group_stack = 0;
};

struct CompileOptions {
    bool bvh_on_device = false;  // build mesh BVHs with rt_bvh_device.hip (the HIP device must already be selected)
    bool rebuild_prim_groups = true;  // re-build sphere / quad subtrees of >= 12 primitives (RT_PRIM_REBUILD=0 keeps the reference's tree)
};

// Which kernels the wavefront scheduler runs for a compiled scene (rt_kernels.hip render_wavefront; also reported by
// rt_scene_program so that the choice is testable without a GPU).
struct WavefrontPlan {
    bool split;        // k_wf_prims (+ k_wf_mesh when the program has mesh ops) instead of the combined k_wf_intersect
    bool vol_prims;    // ... with the volumes inside k_wf_prims<VOL>
    bool multi_mesh;   // k_wf_mesh<MULTI>: more than one mesh op
    bool groups;       // k_wf_prims<GROUPS>: re-built primitive groups searched through their 4-wide BVH
};
WavefrontPlan plan_wavefront(const CompiledScene& cs);

// Returns RT_OK or a negative RtStatus with `err` set.
int compile_scene(const RtSceneDesc* desc, CompiledScene* out, std::string* err, const CompileOptions& opt = CompileOptions());

}  // namespace rt
