// SceneBuilder implementation.  Bounding boxes reproduce the reference bit for bit,
// including its quirks:
//   * every combine/from-points call pads by 0.001 (aabb.rs:9,26,44), so a list that was
//     built with n `add` calls carries n paddings around its first member (list.rs:52);
//   * `bounds_max = -bounds_min` (aabb.rs:13,31) negates the homogeneous w too, so the
//     max corner of every *computed* box has w = -1 and Transform::update_bounds
//     (transform.rs:98-118) applies the translation to that one corner with the wrong sign.
#include "scene_builder.h"

#include <algorithm>
#include <cstring>

namespace rth {

static const V4 kEpsilonVec = {0.001, 0.001, 0.001, 0.0};  // aabb.rs:9

Aabb combine_bounds(const Aabb* boxes, size_t n) {
    V4 lo = {kInf, kInf, kInf, 1.0};  // constants.rs:3
    V4 hi = -lo;
    double* l = &lo.x;
    double* h = &hi.x;
    for (size_t k = 0; k < n; k++) {
        const double* bl = &boxes[k].lo.x;
        const double* bh = &boxes[k].hi.x;
        for (int i = 0; i < 3; i++) {
            if (bl[i] < l[i]) l[i] = bl[i];
            if (bh[i] > h[i]) h[i] = bh[i];
        }
    }
    return {lo - kEpsilonVec, hi + kEpsilonVec};
}

Aabb get_bounding_box(const V4* pts, size_t n) {
    V4 lo = {kInf, kInf, kInf, 1.0};
    V4 hi = -lo;
    double* l = &lo.x;
    double* h = &hi.x;
    for (size_t k = 0; k < n; k++) {
        const double* p = &pts[k].x;
        for (int i = 0; i < 3; i++) {
            if (p[i] < l[i]) l[i] = p[i];
            if (p[i] > h[i]) h[i] = p[i];
        }
    }
    return {lo - kEpsilonVec, hi + kEpsilonVec};
}

// The w components of a node's box are kept in p[10], p[11] (unused by every node type)
// so that bounds_of() can return the reference's full Vec4 corners.
int SceneBuilder::push_node(RtNode n) {
    nodes_.push_back(n);
    children_.emplace_back();
    return int(nodes_.size()) - 1;
}

void SceneBuilder::set_bounds(int node, const Aabb& b) {
    RtNode& n = nodes_[node];
    n.bounds[0] = b.lo.x; n.bounds[1] = b.lo.y; n.bounds[2] = b.lo.z;
    n.bounds[3] = b.hi.x; n.bounds[4] = b.hi.y; n.bounds[5] = b.hi.z;
    n.p[10] = b.lo.w;
    n.p[11] = b.hi.w;
}

Aabb SceneBuilder::bounds_of(int node) const {
    const RtNode& n = nodes_[node];
    return {{n.bounds[0], n.bounds[1], n.bounds[2], n.p[10]},
            {n.bounds[3], n.bounds[4], n.bounds[5], n.p[11]}};
}

static RtNode blank_node(uint32_t type) {
    RtNode n;
    std::memset(&n, 0, sizeof n);
    n.type = type;
    n.material = -1;
    n.mesh = -1;
    n.transform = -1;
    return n;
}

// ---- textures ----
int SceneBuilder::tex_const_color(double r, double g, double b) {
    RtTexture t{}; t.type = RT_TEX_CONST_COLOR; t.a = t.b = t.c = -1; t.v[0] = r; t.v[1] = g; t.v[2] = b; t.scale = 1.0;
    textures_.push_back(t);
    return int(textures_.size()) - 1;
}
int SceneBuilder::tex_const_float(double k) {
    RtTexture t{}; t.type = RT_TEX_CONST_FLOAT; t.a = t.b = t.c = -1; t.v[0] = k; t.scale = 1.0;
    textures_.push_back(t);
    return int(textures_.size()) - 1;
}
int SceneBuilder::tex_checker(int even, int odd, double scale, bool solid) {
    RtTexture t{}; t.type = solid ? RT_TEX_CHECKER_SOLID : RT_TEX_CHECKER; t.a = even; t.b = odd; t.c = -1; t.scale = scale;
    textures_.push_back(t);
    return int(textures_.size()) - 1;
}
int SceneBuilder::tex_lerp(int a, int b, int tt) {
    RtTexture t{}; t.type = RT_TEX_LERP; t.a = a; t.b = b; t.c = tt; t.scale = 1.0;
    textures_.push_back(t);
    return int(textures_.size()) - 1;
}
int SceneBuilder::tex_channel(int color, uint32_t channel) {
    RtTexture t{}; t.type = RT_TEX_CHANNEL; t.a = color; t.b = t.c = -1; t.channel = channel; t.scale = 1.0;
    textures_.push_back(t);
    return int(textures_.size()) - 1;
}
int SceneBuilder::tex_uv_debug() {
    RtTexture t{}; t.type = RT_TEX_UV_DEBUG; t.a = t.b = t.c = -1; t.scale = 1.0;
    textures_.push_back(t);
    return int(textures_.size()) - 1;
}
int SceneBuilder::tex_image(std::vector<float> rgb, uint32_t width, uint32_t height) {
    images_.push_back(std::make_unique<std::vector<float>>(std::move(rgb)));
    RtTexture t{}; t.type = RT_TEX_IMAGE; t.a = t.b = t.c = -1; t.scale = 1.0;
    t.texels = images_.back()->data(); t.width = width; t.height = height;
    textures_.push_back(t);
    return int(textures_.size()) - 1;
}
// PerlinNoise3D::new (perlin.rs:21-36): 256 random unit vectors, then three Fisher-Yates permutations
// (perlin.rs:50-55: for i in (1..256).rev() { swap(i, gen_range(0..=i)) }).  Same draw order, our RNG.
int SceneBuilder::noise_perlin(SceneRng& rng) {
    auto p = std::make_unique<Perlin>();
    p->vec.resize(256 * 3);
    for (int i = 0; i < 256; i++) {
        double x = rng.normal(), y = rng.normal(), z = rng.normal();  // Vec4::random_unit (vec4.rs:42-48)
        double len = std::sqrt(x * x + y * y + z * z);
        p->vec[3 * i + 0] = x / len; p->vec[3 * i + 1] = y / len; p->vec[3 * i + 2] = z / len;
    }
    p->perm.resize(3 * 256);
    for (int a = 0; a < 3; a++) {
        uint32_t* q = p->perm.data() + 256 * a;
        for (uint32_t i = 0; i < 256; i++) q[i] = i;
        for (uint32_t i = 255; i >= 1; i--) std::swap(q[i], q[rng.below(i + 1)]);
    }
    perlins_.push_back(std::move(p));
    return int(perlins_.size()) - 1;
}
int SceneBuilder::tex_noise_solid(int generator, double scale, uint32_t samples) {
    const Perlin& g = *perlins_[size_t(generator)];
    RtTexture t{}; t.type = RT_TEX_NOISE_SOLID; t.a = t.b = t.c = -1; t.scale = 1.0;
    t.v[0] = t.v[1] = t.v[2] = scale; t.samples = samples;
    t.perlin_vec = g.vec.data(); t.perlin_perm = g.perm.data();
    textures_.push_back(t);
    return int(textures_.size()) - 1;
}
bool SceneBuilder::tex_is_color(int tex) const {
    const RtTexture& t = textures_[tex];
    switch (t.type) {
        case RT_TEX_CONST_COLOR: case RT_TEX_UV_DEBUG: case RT_TEX_IMAGE: return true;
        case RT_TEX_CONST_FLOAT: case RT_TEX_CHANNEL: case RT_TEX_NOISE_SOLID: return false;
        default: return tex_is_color(t.a);  // checker / lerp inherit their inputs' type
    }
}

// ---- materials ----
static RtMaterial mk_mat(uint32_t type, int a, int b, int c, double ior) {
    RtMaterial m{}; m.type = type; m.tex_a = a; m.tex_b = b; m.tex_c = c; m.ior = ior;
    return m;
}
int SceneBuilder::mat_lambertian(int albedo) { materials_.push_back(mk_mat(RT_MAT_LAMBERTIAN, albedo, -1, -1, 0)); return int(materials_.size()) - 1; }
int SceneBuilder::mat_metal(int albedo, int rough) { materials_.push_back(mk_mat(RT_MAT_METAL, albedo, rough, -1, 0)); return int(materials_.size()) - 1; }
int SceneBuilder::mat_dielectric(double ior) { materials_.push_back(mk_mat(RT_MAT_DIELECTRIC, -1, -1, -1, ior)); return int(materials_.size()) - 1; }
int SceneBuilder::mat_glossy(int albedo, int rough, double ior, int nm) { materials_.push_back(mk_mat(RT_MAT_GLOSSY, albedo, rough, nm, ior)); return int(materials_.size()) - 1; }
int SceneBuilder::mat_emissive(int emission) { materials_.push_back(mk_mat(RT_MAT_EMISSIVE, emission, -1, -1, 0)); return int(materials_.size()) - 1; }
int SceneBuilder::mat_isotropic(int albedo) { materials_.push_back(mk_mat(RT_MAT_ISOTROPIC, albedo, -1, -1, 0)); return int(materials_.size()) - 1; }
int SceneBuilder::mat_normal_debug(int nm) { materials_.push_back(mk_mat(RT_MAT_NORMAL_DEBUG, -1, -1, nm, 0)); return int(materials_.size()) - 1; }

// ---- objects ----
int SceneBuilder::sphere(V4 c, double r, int material) {
    RtNode n = blank_node(RT_NODE_SPHERE);
    n.material = material;
    n.p[0] = c.x; n.p[1] = c.y; n.p[2] = c.z; n.p[3] = r;
    int id = push_node(n);
    V4 rv = vec(r, r, r);
    set_bounds(id, {c - rv, c + rv});  // sphere.rs:28-29 (inverted for r < 0, SURVEY B-8)
    return id;
}

int SceneBuilder::plane(V4 center, V4 u, V4 v, int material, bool backface, std::string* err) {
    if (dot(u, v) != 0.0) {  // plane.rs:30-32 panics here
        if (err) *err = "The UV vectors must be orthogonal!";
        return -1;
    }
    RtNode n = blank_node(RT_NODE_PLANE);
    n.material = material;
    n.flags = backface ? RT_PLANE_RENDER_BACKFACE : 0u;
    n.p[0] = center.x; n.p[1] = center.y; n.p[2] = center.z;
    n.p[3] = u.x; n.p[4] = u.y; n.p[5] = u.z;
    n.p[6] = v.x; n.p[7] = v.y; n.p[8] = v.z;
    int id = push_node(n);
    V4 corners[4] = {center + u + v, center + u - v, center - u + v, center - u - v};  // plane.rs:41-46
    set_bounds(id, get_bounding_box(corners, 4));
    return id;
}

int SceneBuilder::box(V4 center, V4 size, int material) {
    int sides = list_new();
    V4 half = size / 2.0;
    V4 dx = vec(half.x, 0, 0), dy = vec(0, half.y, 0), dz = vec(0, 0, half.z);
    // obj_box.rs:16-45, same order and same (u, v) pairs
    list_add(sides, plane(center + dy, dx, -dz, material, false, nullptr));
    list_add(sides, plane(center - dy, -dx, -dz, material, false, nullptr));
    list_add(sides, plane(center - dx, dz, dy, material, false, nullptr));
    list_add(sides, plane(center + dx, -dz, dy, material, false, nullptr));
    list_add(sides, plane(center - dz, -dx, dy, material, false, nullptr));
    list_add(sides, plane(center + dz, dx, dy, material, false, nullptr));
    return sides;
}

int SceneBuilder::mesh(std::unique_ptr<MeshData> data, int material) {
    RtNode n = blank_node(RT_NODE_MESH);
    n.material = material;
    n.mesh = int(mesh_data_.size());
    std::vector<V4> pts(data->positions.size() / 3);
    for (size_t i = 0; i < pts.size(); i++)
        pts[i] = point(data->positions[3 * i], data->positions[3 * i + 1], data->positions[3 * i + 2]);
    Aabb b = get_bounding_box(pts.data(), pts.size());  // mesh.rs:45
    mesh_data_.push_back(std::move(data));
    int id = push_node(n);
    set_bounds(id, b);
    return id;
}

int SceneBuilder::list_new() {
    int id = push_node(blank_node(RT_NODE_LIST));
    set_bounds(id, {{kInf, kInf, kInf, 1.0}, {-kInf, -kInf, -kInf, -1.0}});  // list.rs:30
    return id;
}

void SceneBuilder::list_add(int list, int object) {
    Aabb two[2] = {bounds_of(list), bounds_of(object)};
    set_bounds(list, combine_bounds(two, 2));  // list.rs:52
    children_[list].push_back(uint32_t(object));
}

int SceneBuilder::transform_new(int object) {
    RtNode n = blank_node(RT_NODE_TRANSFORM);
    n.transform = int(transforms_.size());
    RtTransform t;
    M4 id4 = m4_identity();
    std::memcpy(t.m, id4.m, sizeof t.m);
    std::memcpy(t.inv, id4.m, sizeof t.inv);
    transforms_.push_back(t);
    int id = push_node(n);
    children_[id].push_back(uint32_t(object));
    set_bounds(id, bounds_of(object));  // transform.rs:32
    return id;
}

void SceneBuilder::transform_update_bounds(int t) {
    Aabb ob = bounds_of(int(children_[t][0]));
    V4 d = ob.hi - ob.lo;
    V4 corners[8] = {ob.lo,
                     ob.lo + vec(0, 0, d.z),
                     ob.lo + vec(0, d.y, 0),
                     ob.lo + vec(0, d.y, d.z),
                     ob.lo + vec(d.x, 0, 0),
                     ob.lo + vec(d.x, 0, d.z),
                     ob.lo + vec(d.x, d.y, 0),
                     ob.hi};
    M4 m;
    std::memcpy(m.m, transforms_[nodes_[t].transform].m, sizeof m.m);
    for (auto& c : corners) c = m * c;  // uses each corner's own w (see file comment)
    set_bounds(t, get_bounding_box(corners, 8));
}

static void apply_op(RtTransform& tr, const M4& op, const M4& inv_op) {
    M4 m, inv;
    std::memcpy(m.m, tr.m, sizeof m.m);
    std::memcpy(inv.m, tr.inv, sizeof inv.m);
    m = op * m;         // transform.rs:53 `self.transform = translation * self.transform`
    inv = inv * inv_op; // transform.rs:54 `self.inv_transform *= inv_translation`
    std::memcpy(tr.m, m.m, sizeof m.m);
    std::memcpy(tr.inv, inv.m, sizeof inv.m);
}

void SceneBuilder::transform_translate(int t, double x, double y, double z) {
    apply_op(transforms_[nodes_[t].transform], m4_translation(x, y, z), m4_translation(-x, -y, -z));
    transform_update_bounds(t);
}

void SceneBuilder::transform_rotate(int t, int axis, double theta) {
    M4 r = axis == 0 ? m4_rotate_x(theta) : axis == 1 ? m4_rotate_y(theta) : m4_rotate_z(theta);
    M4 ri = axis == 0 ? m4_rotate_x(-theta) : axis == 1 ? m4_rotate_y(-theta) : m4_rotate_z(-theta);
    apply_op(transforms_[nodes_[t].transform], r, ri);
    transform_update_bounds(t);
}

void SceneBuilder::transform_scale(int t, double x, double y, double z) {
    apply_op(transforms_[nodes_[t].transform], m4_scale(x, y, z), m4_scale(1.0 / x, 1.0 / y, 1.0 / z));
    transform_update_bounds(t);
}

int SceneBuilder::null_object() {
    int id = push_node(blank_node(RT_NODE_NULL));
    set_bounds(id, {{kInf, kInf, kInf, 1.0}, {-kInf, -kInf, -kInf, -1.0}});  // null_obj.rs:21
    return id;
}

// bvh.rs:32-80.  `objects` is consumed exactly like the Vec there (pop from the back).
int SceneBuilder::bvh(std::vector<int> objects, const bool axes[3], SceneRng& rng) {
    uint32_t axis = rng.below(3);
    while (!axes[axis]) axis = rng.below(3);

    int c0, c1;
    Aabb b;
    size_t count = objects.size();
    if (count == 1) {
        c0 = objects.back();
        c1 = null_object();
        b = bounds_of(c0);
    } else if (count == 2) {
        c0 = objects[1];  // (objects.pop(), objects.pop()) = (last, first)
        c1 = objects[0];
        Aabb two[2] = {bounds_of(c0), bounds_of(c1)};
        b = combine_bounds(two, 2);
    } else {
        // sort_unstable_by(total_cmp of bounds min along axis); tie order is an
        // implementation detail of Rust's pdqsort, ties are not expected in practice.
        std::stable_sort(objects.begin(), objects.end(), [&](int a, int c) {
            return nodes_[a].bounds[axis] < nodes_[c].bounds[axis];
        });
        size_t mid = count / 2;
        std::vector<int> first(objects.begin(), objects.begin() + mid);
        std::vector<int> second(objects.begin() + mid, objects.end());
        c0 = bvh(std::move(first), axes, rng);
        c1 = bvh(std::move(second), axes, rng);
        Aabb two[2] = {bounds_of(c0), bounds_of(c1)};
        b = combine_bounds(two, 2);
    }
    int id = push_node(blank_node(RT_NODE_BVH));
    children_[id].push_back(uint32_t(c0));
    children_[id].push_back(uint32_t(c1));
    set_bounds(id, b);
    return id;
}

static const double kF64Max = std::numeric_limits<double>::max();

int SceneBuilder::sky(int emission_tex) {
    RtNode n = blank_node(RT_NODE_SKY);
    n.material = mat_emissive(emission_tex);  // sky.rs:22
    int id = push_node(n);
    set_bounds(id, {point(-kF64Max, -kF64Max, -kF64Max), point(kF64Max, kF64Max, kF64Max)});  // sky.rs:54-59
    return id;
}

int SceneBuilder::sun(int emission_tex, V4 direction) {
    RtNode n = blank_node(RT_NODE_SUN);
    n.material = mat_emissive(emission_tex);
    V4 d = to_unit(direction);  // sun.rs:27
    n.p[0] = d.x; n.p[1] = d.y; n.p[2] = d.z;
    int id = push_node(n);
    set_bounds(id, {point(-kF64Max, -kF64Max, -kF64Max), point(kF64Max, kF64Max, kF64Max)});
    return id;
}

int SceneBuilder::volume(int boundary, int material, double density) {
    RtNode n = blank_node(RT_NODE_VOLUME);
    n.material = material;
    n.p[0] = density;
    int id = push_node(n);
    children_[id].push_back(uint32_t(boundary));
    set_bounds(id, bounds_of(boundary));  // volume.rs:73-75
    return id;
}

const RtSceneDesc* SceneBuilder::finish(int world, int lights, uint32_t flags) {
    child_indices_.clear();
    for (size_t i = 0; i < nodes_.size(); i++) {
        nodes_[i].first_child = uint32_t(child_indices_.size());
        nodes_[i].n_children = uint32_t(children_[i].size());
        child_indices_.insert(child_indices_.end(), children_[i].begin(), children_[i].end());
    }
    meshes_.clear();
    for (auto& md : mesh_data_) {
        RtMesh m{};
        m.positions = md->positions.data();
        m.normals = md->normals.data();
        m.uvs = md->uvs.empty() ? nullptr : md->uvs.data();
        m.tri_pos = md->tri_pos.data();
        m.tri_nrm = md->tri_nrm.data();
        m.tri_uv = md->any_uv ? md->tri_uv.data() : nullptr;
        m.n_positions = uint32_t(md->positions.size() / 3);
        m.n_normals = uint32_t(md->normals.size() / 3);
        m.n_uvs = uint32_t(md->uvs.size() / 3);
        m.n_triangles = uint32_t(md->tri_pos.size() / 3);
        m.flags = 0;
        meshes_.push_back(m);
    }
    desc_ = RtSceneDesc{};
    desc_.abi_version = RT_MI355_ABI_VERSION;
    desc_.n_nodes = uint32_t(nodes_.size());
    desc_.nodes = nodes_.data();
    desc_.n_child_indices = uint32_t(child_indices_.size());
    desc_.child_indices = child_indices_.data();
    desc_.n_transforms = uint32_t(transforms_.size());
    desc_.transforms = transforms_.data();
    desc_.n_meshes = uint32_t(meshes_.size());
    desc_.meshes = meshes_.data();
    desc_.n_materials = uint32_t(materials_.size());
    desc_.materials = materials_.data();
    desc_.n_textures = uint32_t(textures_.size());
    desc_.textures = textures_.data();
    desc_.world_root = uint32_t(world);
    desc_.lights_root = uint32_t(lights);
    desc_.flags = flags;
    return &desc_;
}

}  // namespace rth
