// SceneBuilder: host-side construction of the Hit tree in the flat form of
// include/rt_mi355.h, computing every node's bounding box with the reference's rules
// (src/aabb.rs:11-48 and each object's constructor).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../../include/rt_mi355.h"
#include "hmath.h"

namespace rth {

struct MeshData {
    std::vector<double> positions, normals, uvs;
    std::vector<uint32_t> tri_pos, tri_nrm;
    std::vector<int32_t> tri_uv;
    bool any_uv = false;
};

class SceneBuilder {
public:
    // --- textures (src/texture/*.rs) ---
    int tex_const_color(double r, double g, double b);
    int tex_const_float(double k);
    int tex_checker(int even, int odd, double scale, bool solid);
    int tex_lerp(int a, int b, int t);
    int tex_channel(int color, uint32_t channel);
    int tex_uv_debug();
    int tex_image(std::vector<float> rgb, uint32_t width, uint32_t height);  // image.rs:29-35
    int noise_perlin(SceneRng& rng);                                         // perlin.rs:21-36 (returns a generator id)
    int tex_noise_solid(int generator, double scale, uint32_t samples);      // scene.rs:555-565
    bool tex_is_color(int tex) const;

    // --- materials (src/material/*.rs) ---
    int mat_lambertian(int albedo);
    int mat_metal(int albedo, int rough);
    int mat_dielectric(double ior);
    int mat_glossy(int albedo, int rough, double ior, int normal_map);
    int mat_emissive(int emission);
    int mat_isotropic(int albedo);
    int mat_normal_debug(int normal_map);

    // --- objects (src/object/*.rs); every function returns a node index ---
    int sphere(V4 center, double radius, int material);                  // sphere.rs:27-37
    int plane(V4 center, V4 u, V4 v, int material, bool backface, std::string* err);  // plane.rs:29-63
    int box(V4 center, V4 size, int material);                           // obj_box.rs:8-48
    int mesh(std::unique_ptr<MeshData> data, int material);              // mesh.rs:38-59
    int list_new();                                                      // list.rs:27-33
    void list_add(int list, int object);                                 // list.rs:51-54
    int transform_new(int object);                                       // transform.rs:27-37
    void transform_translate(int t, double x, double y, double z);       // transform.rs:49-57
    void transform_rotate(int t, int axis, double theta);                // transform.rs:59-87
    void transform_scale(int t, double x, double y, double z);           // transform.rs:89-96
    int bvh(std::vector<int> objects, const bool axes[3], SceneRng& rng);  // bvh.rs:32-80
    int sky(int emission_tex);                                           // sky.rs:21-24
    int sun(int emission_tex, V4 direction);                             // sun.rs:23-29
    int volume(int boundary, int material, double density);              // volume.rs:22-30
    int null_object();                                                   // bvh/null_obj.rs

    Aabb bounds_of(int node) const;
    size_t node_count() const { return nodes_.size(); }

    // Freezes the tables into an RtSceneDesc whose pointers stay valid while `this` lives.
    const RtSceneDesc* finish(int world, int lights, uint32_t flags = 0);

private:
    int push_node(RtNode n);
    void set_bounds(int node, const Aabb& b);
    void transform_update_bounds(int t);  // transform.rs:98-118

    std::vector<RtNode> nodes_;
    std::vector<std::vector<uint32_t>> children_;  // per node
    std::vector<uint32_t> child_indices_;
    std::vector<RtTransform> transforms_;
    std::vector<std::unique_ptr<MeshData>> mesh_data_;
    std::vector<RtMesh> meshes_;
    std::vector<RtMaterial> materials_;
    std::vector<RtTexture> textures_;
    std::vector<std::unique_ptr<std::vector<float>>> images_;
    struct Perlin { std::vector<double> vec; std::vector<uint32_t> perm; };
    std::vector<std::unique_ptr<Perlin>> perlins_;
    RtSceneDesc desc_{};
};

Aabb combine_bounds(const Aabb* boxes, size_t n);          // aabb.rs:11-27
Aabb get_bounding_box(const V4* points, size_t n);         // aabb.rs:29-45

}  // namespace rth
