// Scene compiler: walks the Hit tree of an RtSceneDesc depth-first — the order in which the
// reference's recursive `test()` calls visit it — and emits the linear scene program plus the
// primitive / mesh / material / light tables of rt_scene.h.  Host, one-shot, f64.
#include "rt_compile.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>

namespace rt {
namespace {

struct V3 {
    double x, y, z;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline void put(double d[3], V3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

struct Compiler {
    const RtSceneDesc& d;
    CompiledScene& out;
    std::string* err;
    CompileOptions opt;
    int status = RT_OK;
    std::vector<int32_t> chain;                      // transforms enclosing the current position
    std::map<std::vector<int32_t>, int32_t> chain_ids;
    std::vector<std::vector<int32_t>> chains;
    std::map<uint32_t, int32_t> sphere_of_node, plane_of_node, sun_of_node;
    std::map<int32_t, std::pair<uint32_t, uint32_t>> mesh_geometry;  // RtMesh index -> (node_base, tri_base)
    std::map<int32_t, uint32_t> mesh_depth;
    std::map<int32_t, uint32_t> mesh_node4_base;
    std::map<int32_t, Bounds<double>> mesh_box;
    std::map<int32_t, int32_t> xform_of;             // RtTransform index -> xforms index

    bool fail(int st, const std::string& msg) {
        if (status == RT_OK) {
            status = st;
            *err = msg;
        }
        return false;
    }

    int32_t chain_id() {
        auto it = chain_ids.find(chain);
        if (it != chain_ids.end()) return it->second;
        int32_t id = int32_t(chains.size());
        chains.push_back(chain);
        chain_ids[chain] = id;
        return id;
    }

    int32_t next_rank = 1;  // reference visiting order of the hit-producing ops (Op::skip)
    void emit(int32_t type, int32_t arg) {
        int32_t rank = 0;
        switch (type) {
            case OP_SPHERE: case OP_PLANE: case OP_MESH: case OP_SKY: case OP_SUN: case OP_VOL_END: rank = next_rank++; break;
            default: break;
        }
        out.ops.push_back({type, arg, rank, chain_id()});
    }

    // ---- primitive groups: an object-BVH / list subtree made of spheres and quads only is re-built ----
    // The reference's own object BVH (bvh.rs:32-82: random axis, median split by the boxes' minima, one object + a
    // NullObject per leaf) is walked depth-first in a fixed order; as skip-pointer ops it costs every lane ~100 serial,
    // divergent steps on the default scene's 440-sphere field.  Only WHICH primitives a ray can reach and which of them
    // is hit first matters for the result, so the subtree is replaced by a binned-SAH tree over the same primitives,
    // emitted as the same kind of ops (OP_BOUNDS with skip pointers; no kernel change):
    //   * boxes of the new tree = unions of the primitives' GEOMETRIC boxes (sphere: centre -+ |r|; quad: the
    //     reference's own box), min / max of the reference's numbers only;
    //   * a reference ancestor box that does not contain a primitive's geometric box can cull real hits (SURVEY B-8:
    //     a negative-radius sphere has an inverted box, and the union with it is only partial): such boxes stay in
    //     front of that primitive as GUARD ops (a box test only asks whether the ray meets the box ahead of t_lo, so it
    //     gives the same verdict wherever in the order it is asked); a primitive behind a box no ray can pass
    //     (inverted / NaN) is unreachable in the reference and is dropped;
    //   * ties at exactly equal t go to the primitive the reference visits first: Op::skip = rank (hit_takes_over).
    struct GroupPrim {
        uint32_t node;
        int32_t rank_order;            // position in the reference's depth-first order inside the group
        double lo[3], hi[3];           // geometric box
        std::vector<int32_t> guards;   // indices into out.bounds, outermost first
    };
    struct BoxD { double lo[3], hi[3]; };
    static bool box_passable(const BoxD& b) {
        for (int a = 0; a < 3; a++) if (!(b.lo[a] <= b.hi[a])) return false;  // inverted or NaN: the slab test never passes
        return true;
    }
    static bool box_contains(const BoxD& outer, const double* lo, const double* hi) {
        for (int a = 0; a < 3; a++) if (!(outer.lo[a] <= lo[a] && hi[a] <= outer.hi[a])) return false;
        return true;
    }
    // Collects the primitives under `node` in the reference's order; false if the subtree holds anything else.
    bool collect_group(uint32_t node, int depth, std::vector<BoxD>& ancestors, std::vector<GroupPrim>& prims, int* order) {
        if (node >= d.n_nodes || depth > 512) return false;
        const RtNode& n = d.nodes[node];
        if (uint64_t(n.first_child) + n.n_children > d.n_child_indices) return false;
        switch (n.type) {
            case RT_NODE_NULL:
                return true;
            case RT_NODE_LIST:
            case RT_NODE_BVH: {
                if (n.type == RT_NODE_BVH && n.n_children != 2) return false;
                const bool check = n.type == RT_NODE_BVH || !(n.flags & RT_LIST_DISABLE_BOUNDS_CHECK);
                if (check) {
                    BoxD b;
                    for (int a = 0; a < 3; a++) { b.lo[a] = n.bounds[a]; b.hi[a] = n.bounds[3 + a]; }
                    ancestors.push_back(b);
                }
                bool ok = true;
                for (uint32_t k = 0; ok && k < n.n_children; k++) ok = collect_group(d.child_indices[n.first_child + k], depth + 1, ancestors, prims, order);
                if (check) ancestors.pop_back();
                return ok;
            }
            case RT_NODE_SPHERE:
            case RT_NODE_PLANE: {
                GroupPrim p;
                p.node = node;
                if (n.type == RT_NODE_SPHERE) {
                    const double r = std::fabs(n.p[3]);
                    for (int a = 0; a < 3; a++) { p.lo[a] = n.p[a] - r; p.hi[a] = n.p[a] + r; }
                } else {
                    for (int a = 0; a < 3; a++) { p.lo[a] = n.bounds[a]; p.hi[a] = n.bounds[3 + a]; }
                }
                for (int a = 0; a < 3; a++) if (!(p.lo[a] <= p.hi[a]) || !std::isfinite(p.lo[a]) || !std::isfinite(p.hi[a])) return false;
                bool reachable = true;
                for (const BoxD& b : ancestors) {
                    if (!box_passable(b)) { reachable = false; break; }
                    if (!box_contains(b, p.lo, p.hi)) {
                        Bounds<double> g;
                        for (int a = 0; a < 3; a++) { g.lo[a] = b.lo[a]; g.hi[a] = b.hi[a]; }
                        out.bounds.push_back(g);
                        p.guards.push_back(int32_t(out.bounds.size()) - 1);
                    }
                }
                if (reachable) {  // (otherwise the reference never reaches it: it takes part in nothing)
                    p.rank_order = (*order)++;
                    prims.push_back(std::move(p));
                }
                return true;
            }
            default:
                return false;
        }
    }

    // The SAH tree over prims[begin, end) (indices in `idx`, permuted in place); leaves hold one or two primitives.
    struct GNode {
        size_t begin, end;
        int left = -1, right = -1;
        BoxD box;
    };
    int build_group_tree(std::vector<GroupPrim>& prims, std::vector<uint32_t>& idx, size_t begin, size_t end, std::vector<GNode>& nodes) {
        GNode g;
        g.begin = begin;
        g.end = end;
        for (int a = 0; a < 3; a++) { g.box.lo[a] = HUGE_VAL; g.box.hi[a] = -HUGE_VAL; }
        for (size_t i = begin; i < end; i++)
            for (int a = 0; a < 3; a++) {
                g.box.lo[a] = std::fmin(g.box.lo[a], prims[idx[i]].lo[a]);
                g.box.hi[a] = std::fmax(g.box.hi[a], prims[idx[i]].hi[a]);
            }
        const int self = int(nodes.size());
        nodes.push_back(g);
        const size_t n = end - begin;
        if (n > 2) {
            // best split of the three sorted sweeps (surface area heuristic on the geometric boxes)
            auto area = [](const BoxD& x) { double dx = x.hi[0] - x.lo[0], dy = x.hi[1] - x.lo[1], dz = x.hi[2] - x.lo[2]; return dx * dy + dy * dz + dz * dx; };
            double best_cost = HUGE_VAL;
            int best_axis = 0;
            size_t best_k = n / 2;
            std::vector<double> right_area(n);
            for (int axis = 0; axis < 3; axis++) {
                std::sort(idx.begin() + begin, idx.begin() + end, [&](uint32_t x, uint32_t y) {
                    const double cx = prims[x].lo[axis] + prims[x].hi[axis], cy = prims[y].lo[axis] + prims[y].hi[axis];
                    return cx < cy || (cx == cy && x < y);
                });
                BoxD acc;
                for (int a = 0; a < 3; a++) { acc.lo[a] = HUGE_VAL; acc.hi[a] = -HUGE_VAL; }
                for (size_t k = n; k-- > 1;) {
                    const GroupPrim& p = prims[idx[begin + k]];
                    for (int a = 0; a < 3; a++) { acc.lo[a] = std::fmin(acc.lo[a], p.lo[a]); acc.hi[a] = std::fmax(acc.hi[a], p.hi[a]); }
                    right_area[k] = area(acc);
                }
                for (int a = 0; a < 3; a++) { acc.lo[a] = HUGE_VAL; acc.hi[a] = -HUGE_VAL; }
                for (size_t k = 1; k < n; k++) {
                    const GroupPrim& p = prims[idx[begin + k - 1]];
                    for (int a = 0; a < 3; a++) { acc.lo[a] = std::fmin(acc.lo[a], p.lo[a]); acc.hi[a] = std::fmax(acc.hi[a], p.hi[a]); }
                    const double cost = area(acc) * double(k) + right_area[k] * double(n - k);
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_k = k; }
                }
            }
            std::sort(idx.begin() + begin, idx.begin() + end, [&](uint32_t x, uint32_t y) {
                const double cx = prims[x].lo[best_axis] + prims[x].hi[best_axis], cy = prims[y].lo[best_axis] + prims[y].hi[best_axis];
                return cx < cy || (cx == cy && x < y);
            });
            const int l = build_group_tree(prims, idx, begin, begin + best_k, nodes);
            const int r = build_group_tree(prims, idx, begin + best_k, end, nodes);
            nodes[size_t(self)].left = l;
            nodes[size_t(self)].right = r;
        }
        return self;
    }

    // Op form of the tree: OP_BOUNDS with skip pointers, the primitives (behind their guard boxes) at the leaves.
    // prim_pc[i] = pc of the op of prims[i].
    void emit_group_ops(const std::vector<GroupPrim>& prims, const std::vector<uint32_t>& idx, const std::vector<GNode>& nodes, int ni,
                        int32_t rank_base, std::vector<int32_t>& prim_pc) {
        const GNode& g = nodes[size_t(ni)];
        Bounds<double> b;
        for (int a = 0; a < 3; a++) { b.lo[a] = g.box.lo[a]; b.hi[a] = g.box.hi[a]; }
        out.bounds.push_back(b);
        const size_t bounds_op = out.ops.size();
        emit(OP_BOUNDS, int32_t(out.bounds.size()) - 1);
        if (g.left < 0) {
            for (size_t i = g.begin; i < g.end; i++) {
                const GroupPrim& p = prims[idx[i]];
                std::vector<size_t> guard_ops;
                for (int32_t gb : p.guards) { guard_ops.push_back(out.ops.size()); emit(OP_BOUNDS, gb); }
                const RtNode& nd = d.nodes[p.node];
                if (nd.type == RT_NODE_SPHERE) emit(OP_SPHERE, sphere_index(p.node));
                else emit(OP_PLANE, plane_index(p.node));
                out.ops.back().skip = rank_base + p.rank_order;  // the reference's visiting order, not the emission order
                prim_pc[idx[i]] = int32_t(out.ops.size()) - 1;
                for (size_t go : guard_ops) out.ops[go].skip = int32_t(out.ops.size());
            }
        } else {
            emit_group_ops(prims, idx, nodes, g.left, rank_base, prim_pc);
            emit_group_ops(prims, idx, nodes, g.right, rank_base, prim_pc);
        }
        out.ops[bounds_op].skip = int32_t(out.ops.size());
    }

    // BVH2 form of the tree in the mesh builder's encoding (both child boxes in the parent; a leaf = positions
    // [begin, end) of `idx`), ready for the 4-wide collapse.
    int32_t group_bvh2(const std::vector<GNode>& nodes, int ni, BvhBuild& b2, uint32_t depth) {
        const GNode& g = nodes[size_t(ni)];
        if (g.left < 0) return ~int32_t((uint32_t(g.begin) << 3) | uint32_t(g.end - g.begin - 1));
        const size_t self = b2.nodes.size();
        b2.nodes.emplace_back();
        if (depth + 1 > b2.max_depth) b2.max_depth = depth + 1;
        const int32_t c0 = group_bvh2(nodes, g.left, b2, depth + 1), c1 = group_bvh2(nodes, g.right, b2, depth + 1);
        BuildNode& n = b2.nodes[self];
        for (int a = 0; a < 3; a++) {
            n.lo0[a] = nodes[size_t(g.left)].box.lo[a]; n.hi0[a] = nodes[size_t(g.left)].box.hi[a];
            n.lo1[a] = nodes[size_t(g.right)].box.lo[a]; n.hi1[a] = nodes[size_t(g.right)].box.hi[a];
        }
        n.c0 = c0;
        n.c1 = c1;
        return int32_t(self);
    }

    // Tries to compile the subtree under `node` as a re-built primitive group; false = compile it op by op as usual.
    bool try_emit_group(uint32_t node, int depth) {
        if (!opt.rebuild_prim_groups) return false;
        const size_t bounds_mark = out.bounds.size();
        std::vector<BoxD> ancestors;
        std::vector<GroupPrim> prims;
        int order = 0;
        if (!collect_group(node, depth, ancestors, prims, &order) || prims.size() < 12) {
            out.bounds.resize(bounds_mark);  // guard boxes of an abandoned attempt
            return false;
        }
        // materialise the primitive tables in the reference's order (sphere / plane indices are handed out on first use)
        for (const GroupPrim& p : prims) {
            int32_t i = d.nodes[p.node].type == RT_NODE_SPHERE ? sphere_index(p.node) : plane_index(p.node);
            if (i < 0) return true;  // status is set: compile fails
        }
        std::vector<uint32_t> idx(prims.size());
        for (size_t i = 0; i < idx.size(); i++) idx[i] = uint32_t(i);
        const int32_t rank_base = next_rank;
        next_rank += int32_t(prims.size());
        std::vector<GNode> nodes;
        build_group_tree(prims, idx, 0, prims.size(), nodes);
        // the 4-wide BVH form first (OP_GROUP in front of the op form), unless the scene already holds too many
        BvhBuild b2;
        const int32_t root_ref = group_bvh2(nodes, 0, b2, 0);
        const bool wide = root_ref >= 0 && out.group_prims.size() + prims.size() < (1u << 27);
        size_t group_op = SIZE_MAX;
        if (wide) {
            group_op = out.ops.size();
            emit(OP_GROUP, int32_t(out.groups.size()));
        }
        std::vector<int32_t> prim_pc(prims.size(), -1);
        emit_group_ops(prims, idx, nodes, 0, rank_base, prim_pc);
        if (wide) {
            out.ops[group_op].skip = int32_t(out.ops.size());
            Bvh4Build b4 = collapse_bvh4(b2);
            GroupRec<double> gr{};
            gr.root = uint32_t(out.group_nodes4.size());
            for (int a = 0; a < 3; a++) { gr.lo[a] = b4.root_lo[a]; gr.hi[a] = b4.root_hi[a]; }
            const uint32_t prim_base = uint32_t(out.group_prims.size());
            for (BuildNode4 nd : b4.nodes) {
                for (int k = 0; k < 4; k++) {
                    if (nd.child[k] == kEmptyChild) continue;
                    if (nd.child[k] >= 0) nd.child[k] += int32_t(gr.root);
                    else {
                        const uint32_t code = uint32_t(~nd.child[k]);
                        nd.child[k] = ~int32_t((((code >> 3) + prim_base) << 3) | (code & 7u));
                    }
                }
                out.group_nodes4.push_back(nd);
            }
            for (size_t j = 0; j < idx.size(); j++) {
                const GroupPrim& p = prims[idx[j]];
                GroupPrimRef r{};
                r.pc = prim_pc[idx[j]];
                r.guard_first = int32_t(out.group_guards.size());
                r.guard_count = int32_t(p.guards.size());
                for (int32_t gb : p.guards) out.group_guards.push_back(gb);
                out.group_prims.push_back(r);
            }
            out.groups.push_back(gr);
            if (b4.max_stack > out.max_group_stack) out.max_group_stack = b4.max_stack;
            if (std::getenv("RT_COMPILE_DEBUG"))
                std::fprintf(stderr, "[rt_compile] primitive group %zu: %zu primitives, BVH2 %zu nodes (depth %u) -> %zu 4-wide nodes, stack %u\n",
                             out.groups.size() - 1, prims.size(), b2.nodes.size(), b2.max_depth, b4.nodes.size(), b4.max_stack);
        }
        out.n_rebuilt_groups++;
        out.n_rebuilt_prims += uint32_t(prims.size());
        return true;
    }

    bool check_material(int32_t m) {
        if (m < 0 || uint32_t(m) >= d.n_materials) return fail(RT_E_INVALID, "material index out of range");
        return true;
    }

    int32_t sphere_index(uint32_t node) {
        auto it = sphere_of_node.find(node);
        if (it != sphere_of_node.end()) return it->second;
        const RtNode& n = d.nodes[node];
        if (!check_material(n.material)) return -1;
        SpherePrim<double> s{};
        s.center[0] = n.p[0]; s.center[1] = n.p[1]; s.center[2] = n.p[2];
        s.radius = n.p[3];
        s.material = n.material;
        out.spheres.push_back(s);
        return sphere_of_node[node] = int32_t(out.spheres.size()) - 1;
    }

    // Plane::new, reference src/object/plane.rs:29-63, same operation order.
    int32_t plane_index(uint32_t node) {
        auto it = plane_of_node.find(node);
        if (it != plane_of_node.end()) return it->second;
        const RtNode& n = d.nodes[node];
        if (!check_material(n.material)) return -1;
        V3 center{n.p[0], n.p[1], n.p[2]}, u{n.p[3], n.p[4], n.p[5]}, v{n.p[6], n.p[7], n.p[8]};
        if (dot(u, v) != 0.0) { fail(RT_E_INVALID, "plane: the UV vectors must be orthogonal (plane.rs:30)"); return -1; }
        V3 u_unit = u / length(u);
        V3 v_unit = v / length(v);
        V3 nn = cross(u, v);
        PlanePrim<double> p{};
        p.area = length(nn) * 4.0;
        put(p.normal, nn / length(nn));
        put(p.corner, (center - u) - v);
        put(p.u, u);
        put(p.v, v);
        put(p.inv_u, (u_unit * 0.5) / length(u));
        put(p.inv_v, (v_unit * 0.5) / length(v));
        p.material = n.material;
        p.backface = (n.flags & RT_PLANE_RENDER_BACKFACE) ? 1 : 0;
        out.planes.push_back(p);
        return plane_of_node[node] = int32_t(out.planes.size()) - 1;
    }

    int32_t sun_index(uint32_t node) {
        auto it = sun_of_node.find(node);
        if (it != sun_of_node.end()) return it->second;
        const RtNode& n = d.nodes[node];
        if (!check_material(n.material)) return -1;
        SunPrim<double> s{};
        s.direction[0] = n.p[0]; s.direction[1] = n.p[1]; s.direction[2] = n.p[2];
        s.material = n.material;
        out.suns.push_back(s);
        return sun_of_node[node] = int32_t(out.suns.size()) - 1;
    }

    int32_t xform_index(int32_t t) {
        if (t < 0 || uint32_t(t) >= d.n_transforms) { fail(RT_E_INVALID, "transform index out of range"); return -1; }
        auto it = xform_of.find(t);
        if (it != xform_of.end()) return it->second;
        const RtTransform& tr = d.transforms[t];
        const double last[4] = {0, 0, 0, 1};
        if (std::memcmp(tr.m + 12, last, sizeof last) != 0 || std::memcmp(tr.inv + 12, last, sizeof last) != 0) {
            fail(RT_E_UNSUPPORTED, "projective transform (last row != 0,0,0,1)");
            return -1;
        }
        Xform<double> x;
        std::memcpy(x.m, tr.m, sizeof x.m);
        std::memcpy(x.inv, tr.inv, sizeof x.inv);
        out.xforms.push_back(x);
        return xform_of[t] = int32_t(out.xforms.size()) - 1;
    }

    bool mesh_geometry_for(int32_t mi, uint32_t* node_base, uint32_t* tri_base, uint32_t* depth, uint32_t* node4_base, Bounds<double>* box) {
        auto it = mesh_geometry.find(mi);
        if (it != mesh_geometry.end()) {
            *node_base = it->second.first;
            *tri_base = it->second.second;
            *depth = mesh_depth[mi];
            *node4_base = mesh_node4_base[mi];
            *box = mesh_box[mi];
            return true;
        }
        const RtMesh& m = d.meshes[mi];
        if ((m.n_triangles && (!m.positions || !m.tri_pos || !m.tri_nrm || !m.normals)))
            return fail(RT_E_INVALID, "mesh arrays missing");
        if (m.n_triangles >= (1u << 27)) return fail(RT_E_UNSUPPORTED, "mesh too large for leaf encoding");
        for (uint32_t i = 0; i < m.n_triangles * 3; i++) {
            if (m.tri_pos[i] >= m.n_positions || m.tri_nrm[i] >= m.n_normals) return fail(RT_E_INVALID, "triangle index out of range");
            if (m.tri_uv && m.tri_uv[i] >= 0 && (uint32_t(m.tri_uv[i]) >= m.n_uvs || !m.uvs)) return fail(RT_E_INVALID, "uv index out of range");
        }
        uint32_t max_leaf = 4;  // triangles per leaf (1..8); RT_BVH_MAX_LEAF overrides for experiments
        if (const char* e = std::getenv("RT_BVH_MAX_LEAF")) { int v = std::atoi(e); if (v >= 1 && v <= 8) max_leaf = uint32_t(v); }
        const bool timing = std::getenv("RT_COMPILE_DEBUG") != nullptr;
        auto t0 = std::chrono::steady_clock::now();
        BvhBuild bvh;
        if (opt.bvh_on_device) {
            std::string berr;
            if (!build_bvh_device(m.positions, m.n_positions, m.tri_pos, m.n_triangles, max_leaf, &bvh, &berr)) return fail(RT_E_DEVICE, berr);
        } else {
            bvh = build_bvh(m.positions, m.tri_pos, m.n_triangles, max_leaf);
        }
        auto t1 = std::chrono::steady_clock::now();
        Bvh4Build bvh4 = collapse_bvh4(bvh);
        auto t2 = std::chrono::steady_clock::now();
        if (timing)
            std::fprintf(stderr, "[rt_compile] mesh %u triangles: %s BVH2 %.0f ms (%zu nodes, depth %u), 4-wide collapse %.0f ms (%zu nodes)\n",
                         m.n_triangles, opt.bvh_on_device ? "device LBVH" : "host binned-SAH", std::chrono::duration<double, std::milli>(t1 - t0).count(), bvh.nodes.size(), bvh.max_depth,
                         std::chrono::duration<double, std::milli>(t2 - t1).count(), bvh4.nodes.size());
        *node4_base = uint32_t(out.nodes4.size());
        // The 4-wide nodes carry ABSOLUTE references (node index into nodes4, first triangle slot into tris / attrs): k_wf_mesh
        // serves every mesh instance of a scene, lane by lane, without per-mesh base registers.
        if (out.tris.size() + m.n_triangles >= (1ull << 28)) return fail(RT_E_UNSUPPORTED, "more than 2^28 triangles in one scene");
        for (BuildNode4 nd : bvh4.nodes) {
            for (int k = 0; k < 4; k++) {
                if (nd.child[k] == kEmptyChild) continue;
                if (nd.child[k] >= 0) nd.child[k] += int32_t(*node4_base);
                else {
                    const uint32_t code = uint32_t(~nd.child[k]);
                    nd.child[k] = ~int32_t((((code >> 3) + uint32_t(out.tris.size())) << 3) | (code & 7u));
                }
            }
            out.nodes4.push_back(nd);
        }
        if (bvh4.max_stack > out.max_bvh4_stack) out.max_bvh4_stack = bvh4.max_stack;
        for (int a = 0; a < 3; a++) { box->lo[a] = bvh4.root_lo[a]; box->hi[a] = bvh4.root_hi[a]; }
        mesh_node4_base[mi] = *node4_base;
        mesh_box[mi] = *box;
        *node_base = uint32_t(out.nodes.size());
        *tri_base = uint32_t(out.tris.size());
        *depth = bvh.max_depth;
        out.nodes.insert(out.nodes.end(), bvh.nodes.begin(), bvh.nodes.end());
        for (uint32_t slot = 0; slot < m.n_triangles; slot++) {
            uint32_t t = bvh.tri_order[slot];
            const double* p0 = m.positions + 3 * size_t(m.tri_pos[3 * size_t(t)]);
            const double* p1 = m.positions + 3 * size_t(m.tri_pos[3 * size_t(t) + 1]);
            const double* p2 = m.positions + 3 * size_t(m.tri_pos[3 * size_t(t) + 2]);
            TriRec<double> r{};
            for (int a = 0; a < 3; a++) {
                r.v0[a] = p0[a];
                r.e1[a] = p1[a] - p0[a];  // mesh.rs:69
                r.e2[a] = p2[a] - p0[a];  // mesh.rs:70
            }
            out.tris.push_back(r);
            TriAttr<double> at{};
            const double* n0 = m.normals + 3 * size_t(m.tri_nrm[3 * size_t(t)]);
            const double* n1 = m.normals + 3 * size_t(m.tri_nrm[3 * size_t(t) + 1]);
            const double* n2 = m.normals + 3 * size_t(m.tri_nrm[3 * size_t(t) + 2]);
            for (int a = 0; a < 3; a++) { at.n0[a] = n0[a]; at.n1[a] = n1[a]; at.n2[a] = n2[a]; }
            bool has_uv = m.tri_uv && m.tri_uv[3 * size_t(t)] >= 0 && m.tri_uv[3 * size_t(t) + 1] >= 0 && m.tri_uv[3 * size_t(t) + 2] >= 0;
            at.has_uv = has_uv ? 1 : 0;
            if (has_uv) {
                const double* a0 = m.uvs + 3 * size_t(m.tri_uv[3 * size_t(t)]);
                const double* a1 = m.uvs + 3 * size_t(m.tri_uv[3 * size_t(t) + 1]);
                const double* a2 = m.uvs + 3 * size_t(m.tri_uv[3 * size_t(t) + 2]);
                at.uv0[0] = a0[0]; at.uv0[1] = a0[1];
                at.uv1[0] = a1[0]; at.uv1[1] = a1[1];
                at.uv2[0] = a2[0]; at.uv2[1] = a2[1];
            }
            out.attrs.push_back(at);
        }
        mesh_geometry[mi] = {*node_base, *tri_base};
        mesh_depth[mi] = bvh.max_depth;
        if (bvh.max_depth > out.max_bvh_depth) out.max_bvh_depth = bvh.max_depth;
        return true;
    }

    int vol_depth = 0;  // volumes whose boundary is being compiled

    bool compile_node(uint32_t node, int depth) {
        if (status != RT_OK) return false;
        if (node >= d.n_nodes) return fail(RT_E_INVALID, "node index out of range");
        if (depth > 512) return fail(RT_E_INVALID, "Hit tree too deep (cycle?)");
        const RtNode& n = d.nodes[node];
        if (uint64_t(n.first_child) + n.n_children > d.n_child_indices) return fail(RT_E_INVALID, "child range out of bounds");
        const uint32_t* kids = d.child_indices + n.first_child;
        switch (n.type) {
            case RT_NODE_SPHERE: {
                int32_t i = sphere_index(node);
                if (i < 0) return false;
                emit(OP_SPHERE, i);
                return true;
            }
            case RT_NODE_PLANE: {
                int32_t i = plane_index(node);
                if (i < 0) return false;
                emit(OP_PLANE, i);
                return true;
            }
            case RT_NODE_SKY:
                if (!check_material(n.material)) return false;
                emit(OP_SKY, n.material);
                return true;
            case RT_NODE_SUN: {
                int32_t i = sun_index(node);
                if (i < 0) return false;
                emit(OP_SUN, i);
                return true;
            }
            case RT_NODE_MESH: {
                if (n.mesh < 0 || uint32_t(n.mesh) >= d.n_meshes) return fail(RT_E_INVALID, "mesh index out of range");
                if (!check_material(n.material)) return false;
                MeshInst mi{};
                Bounds<double> mbox;
                if (!mesh_geometry_for(n.mesh, &mi.node_base, &mi.tri_base, &mi.max_depth, &mi.node4_base, &mbox)) return false;
                out.mesh_bounds.push_back(mbox);
                mi.material = n.material;
                mi.flags = d.meshes[n.mesh].flags & (RT_MESH_FLAT_SHADING | RT_MESH_HIT_BACK_FACES);
                if (d.meshes[n.mesh].tri_uv) mi.flags |= MESH_HAS_UV;
                mi.n_tris = d.meshes[n.mesh].n_triangles;
                out.meshes.push_back(mi);
                emit(OP_MESH, int32_t(out.meshes.size()) - 1);
                return true;
            }
            case RT_NODE_LIST:
            case RT_NODE_BVH: {
                if (n.type == RT_NODE_BVH && n.n_children != 2) return fail(RT_E_INVALID, "bvh node needs two children");
                if (vol_depth == 0 && try_emit_group(node, depth)) return status == RT_OK;
                size_t bounds_op = SIZE_MAX;
                bool check = n.type == RT_NODE_BVH || !(n.flags & RT_LIST_DISABLE_BOUNDS_CHECK);
                if (check) {
                    Bounds<double> b;
                    for (int a = 0; a < 3; a++) { b.lo[a] = n.bounds[a]; b.hi[a] = n.bounds[3 + a]; }
                    out.bounds.push_back(b);
                    bounds_op = out.ops.size();
                    emit(OP_BOUNDS, int32_t(out.bounds.size()) - 1);
                }
                for (uint32_t k = 0; k < n.n_children; k++)
                    if (!compile_node(kids[k], depth + 1)) return false;
                if (check) out.ops[bounds_op].skip = int32_t(out.ops.size());
                return true;
            }
            case RT_NODE_TRANSFORM: {
                if (n.n_children != 1) return fail(RT_E_INVALID, "transform node needs one child");
                int32_t x = xform_index(n.transform);
                if (x < 0) return false;
                emit(OP_XFORM_PUSH, x);
                chain.push_back(x);
                bool ok = compile_node(kids[0], depth + 1);
                chain.pop_back();
                if (!ok) return false;
                emit(OP_XFORM_POP, x);  // chain of the POP = the parent's chain
                return true;
            }
            case RT_NODE_NULL:
                return true;  // NullObject::test never hits (null_obj.rs:17)
            case RT_NODE_VOLUME: {  // volume.rs:33-71: the boundary's sub-program is emitted twice (entry search, exit search)
                if (n.n_children != 1) return fail(RT_E_INVALID, "volume node needs one child (its boundary)");
                if (vol_depth >= kMaxVolDepth) return fail(RT_E_UNSUPPORTED, "volumes nested more than two levels deep (a volume inside the boundary of a volume inside a boundary)");
                if (!check_material(n.material)) return false;
                VolumeRec<double> v{};
                v.neg_inv_density = -1.0 / n.p[0];  // Volume::new (volume.rs:23)
                v.material = n.material;
                out.volumes.push_back(v);
                const int32_t vi = int32_t(out.volumes.size()) - 1;
                vol_depth++;
                emit(OP_VOL_BEGIN, vi);
                bool ok = compile_node(kids[0], depth + 1);
                const size_t mid = out.ops.size();
                emit(OP_VOL_MID, vi);
                ok = ok && compile_node(kids[0], depth + 1);
                emit(OP_VOL_END, vi);
                vol_depth--;
                if (!ok) return false;
                out.ops[mid].skip = int32_t(out.ops.size());
                return true;
            }
            default:
                return fail(RT_E_INVALID, "unknown node type");
        }
    }

    // ---- texture expressions -> postfix programs ----
    std::map<const void*, uint32_t> image_base, perlin_index;
    std::map<int32_t, int32_t> program_of;  // texture id -> packed program id

    // Emits the program of texture `t`; returns its kind (1 colour, 2 float) or 0 on error.
    // `live` = values already on the stack when this sub-expression starts.
    int emit_texture(int32_t t, int depth, int live, int* max_live, bool* needs_uv) {
        if (t < 0 || uint32_t(t) >= d.n_textures) { fail(RT_E_INVALID, "texture index out of range"); return 0; }
        if (depth > 64) { fail(RT_E_INVALID, "texture graph too deep (cycle?)"); return 0; }
        const RtTexture& tx = d.textures[t];
        TextureRec<double> r{};
        r.type = int32_t(tx.type);
        r.v[0] = tx.v[0]; r.v[1] = tx.v[1]; r.v[2] = tx.v[2];
        r.scale = tx.scale;
        int kind = 0;
        switch (tx.type) {
            case RT_TEX_CONST_COLOR: kind = 1; break;
            case RT_TEX_CONST_FLOAT: kind = 2; break;
            case RT_TEX_UV_DEBUG: kind = 1; *needs_uv = true; break;
            case RT_TEX_IMAGE: {
                kind = 1;
                out.needs_tex_interpreter = true;
                *needs_uv = true;
                if (!tx.texels || tx.width == 0 || tx.height == 0) { fail(RT_E_INVALID, "image texture without texels"); return 0; }
                auto it = image_base.find(tx.texels);
                if (it == image_base.end()) {
                    size_t n = size_t(tx.width) * tx.height * 3;
                    if ((out.texels.size() + n) / 3 > 0xFFFFFFFFull) { fail(RT_E_UNSUPPORTED, "image textures too large"); return 0; }
                    uint32_t base = uint32_t(out.texels.size() / 3);
                    out.texels.insert(out.texels.end(), tx.texels, tx.texels + n);
                    it = image_base.emplace(tx.texels, base).first;
                }
                r.data = it->second;
                r.width = tx.width;
                r.height = tx.height;
                break;
            }
            case RT_TEX_NOISE_SOLID: {
                kind = 2;
                out.needs_tex_interpreter = true;
                if (!tx.perlin_vec || !tx.perlin_perm) { fail(RT_E_INVALID, "noise texture without generator tables"); return 0; }
                auto it = perlin_index.find(tx.perlin_vec);
                if (it == perlin_index.end()) {
                    uint32_t idx = uint32_t(out.perlin_vec.size() / 768);
                    out.perlin_vec.insert(out.perlin_vec.end(), tx.perlin_vec, tx.perlin_vec + 768);
                    for (int k = 0; k < 768; k++) out.perlin_perm.push_back(tx.perlin_perm[k] & 255u);
                    it = perlin_index.emplace(tx.perlin_vec, idx).first;
                }
                r.data = it->second;
                r.aux = int32_t(tx.samples);
                break;
            }
            case RT_TEX_CHECKER:
            case RT_TEX_CHECKER_SOLID: {
                if (tx.type == RT_TEX_CHECKER) *needs_uv = true;
                int ka = emit_texture(tx.a, depth + 1, live, max_live, needs_uv);
                if (!ka) return 0;
                r.aux = int32_t(out.textures.size()) - 1;  // root op of the even input (eval_texture_simple)
                int kb = emit_texture(tx.b, depth + 1, live + 1, max_live, needs_uv);
                if (!kb) return 0;
                r.data = uint32_t(out.textures.size()) - 1u;  // root op of the odd input
                if (ka != kb) { fail(RT_E_INVALID, "checker inputs differ in type"); return 0; }
                kind = ka;
                break;
            }
            case RT_TEX_LERP: {
                out.needs_tex_interpreter = true;
                int ka = emit_texture(tx.a, depth + 1, live, max_live, needs_uv);
                if (!ka) return 0;
                int kb = emit_texture(tx.b, depth + 1, live + 1, max_live, needs_uv);
                if (!kb) return 0;
                int kc = emit_texture(tx.c, depth + 1, live + 2, max_live, needs_uv);
                if (!kc) return 0;
                if (ka != kb || kc != 2) { fail(RT_E_INVALID, "lerp inputs have wrong types"); return 0; }
                kind = ka;
                break;
            }
            case RT_TEX_CHANNEL: {
                out.needs_tex_interpreter = true;
                int ka = emit_texture(tx.a, depth + 1, live, max_live, needs_uv);
                if (!ka) return 0;
                if (ka != 1) { fail(RT_E_INVALID, "channel input must be a colour texture"); return 0; }
                if (tx.channel > 3) { fail(RT_E_INVALID, "channel index out of range"); return 0; }
                r.aux = int32_t(tx.channel);
                kind = 2;
                break;
            }
            default:
                fail(RT_E_INVALID, "unknown texture type");
                return 0;
        }
        if (live + 1 > *max_live) *max_live = live + 1;
        out.textures.push_back(r);
        return kind;
    }

    // Compiles (once) the program of the texture in a material slot; `want` = 1 colour, 2 float.
    bool texture_program(int32_t t, int want, int32_t* packed, bool* needs_uv) {
        auto it = program_of.find(t);
        if (it == program_of.end()) {
            size_t first = out.textures.size();
            int max_live = 0;
            bool uv = false;
            int kind = emit_texture(t, 0, 0, &max_live, &uv);
            if (!kind) return false;
            size_t count = out.textures.size() - first;
            if (max_live > kTexStackMax)
                return fail(RT_E_UNSUPPORTED, "texture expression needs more than 16 live values");
            if (first >= (1u << kTexProgShift) || count >= (1u << 11)) return fail(RT_E_UNSUPPORTED, "texture programs too large");
            // kind and uv use are kept in the two top bits of the map entry's companion tables
            program_kind[t] = kind;
            program_uv[t] = uv;
            it = program_of.emplace(t, int32_t(first | (count << kTexProgShift))).first;
        }
        if (program_kind[t] != want) return fail(RT_E_INVALID, want == 1 ? "expected a colour texture" : "expected a float texture");
        if (program_uv[t]) *needs_uv = true;
        *packed = it->second;
        return true;
    }
    std::map<int32_t, int> program_kind;
    std::map<int32_t, bool> program_uv;

    bool compile_tables() {
        out.materials.resize(d.n_materials);
        out.material_params.resize(d.n_materials);
        for (uint32_t i = 0; i < d.n_materials; i++) {
            const RtMaterial& m = d.materials[i];
            MaterialRec r{};
            r.type = int32_t(m.type);
            r.tex_a = r.tex_b = r.tex_c = -1;
            bool needs_uv = false;
            switch (m.type) {
                case RT_MAT_LAMBERTIAN: case RT_MAT_EMISSIVE: case RT_MAT_ISOTROPIC:
                    if (!texture_program(m.tex_a, 1, &r.tex_a, &needs_uv)) return false;
                    break;
                case RT_MAT_METAL:
                    if (!texture_program(m.tex_a, 1, &r.tex_a, &needs_uv) || !texture_program(m.tex_b, 2, &r.tex_b, &needs_uv)) return false;
                    break;
                case RT_MAT_GLOSSY:
                    if (!texture_program(m.tex_a, 1, &r.tex_a, &needs_uv) || !texture_program(m.tex_b, 2, &r.tex_b, &needs_uv)) return false;
                    if (m.tex_c >= 0) {
                        if (!texture_program(m.tex_c, 1, &r.tex_c, &needs_uv)) return false;
                        r.has_normal_map = 1;
                        out.needs_tex_interpreter = true;
                    }
                    break;
                case RT_MAT_DIELECTRIC:
                    break;
                case RT_MAT_NORMAL_DEBUG:
                    if (m.tex_c >= 0) {
                        if (!texture_program(m.tex_c, 1, &r.tex_c, &needs_uv)) return false;
                        r.has_normal_map = 1;
                        out.needs_tex_interpreter = true;
                    }
                    break;
                default:
                    return fail(RT_E_INVALID, "unknown material type");
            }
            // a normal map needs the tangent frame, which the sphere only computes together with (u, v) (sphere.rs:78-88)
            r.needs_uv = (needs_uv || r.has_normal_map) ? 1 : 0;
            out.materials[i] = r;
            out.material_params[i].ior = m.ior;
            out.material_params[i].inv_ior = 1.0 / m.ior;  // glossy.rs:30
        }
        return true;
    }

    // One entry of the light table for `node`.  An ObjectList member becomes a LIGHT_LIST entry whose own members are
    // placed behind (compile_light_entries): a tree of LightRecs with the reference's recursion (list.rs:80-100).
    bool light_of(uint32_t node, LightRec* l) {
        if (node >= d.n_nodes) return fail(RT_E_INVALID, "light node index out of range");
        const RtNode& n = d.nodes[node];
        *l = LightRec{LIGHT_OTHER, 0};
        switch (n.type) {
            case RT_NODE_PLANE: l->kind = LIGHT_PLANE; l->index = plane_index(node); break;
            case RT_NODE_SPHERE: l->kind = LIGHT_SPHERE; l->index = sphere_index(node); break;
            case RT_NODE_SKY: l->kind = LIGHT_SKY; break;
            case RT_NODE_SUN: l->kind = LIGHT_SUN; l->index = sun_index(node); break;
            case RT_NODE_LIST: l->kind = LIGHT_LIST; break;
            default: break;  // Transform / mesh / bvh / volume / null: pdf_value 0, random (1,0,0)
        }
        return l->index >= 0;
    }

    // Places the entries of one ObjectList (its members, contiguous) and, behind them, recursively the members of every
    // member that is itself a list.  depth = nesting level of this list inside `lights` (the kernels evaluate nested
    // pdf_value sums on an explicit stack of kMaxLightDepth frames).
    bool compile_light_entries(const uint32_t* nodes, uint32_t count, int depth) {
        if (depth >= kMaxLightDepth) return fail(RT_E_UNSUPPORTED, "lists nested more than 8 levels inside `lights`");
        const size_t base = out.lights.size();
        for (uint32_t k = 0; k < count; k++) {
            LightRec l;
            if (!light_of(nodes[k], &l)) return false;
            out.lights.push_back(l);
        }
        for (uint32_t k = 0; k < count; k++) {
            if (out.lights[base + k].kind != LIGHT_LIST) continue;
            const RtNode& n = d.nodes[nodes[k]];
            if (uint64_t(n.first_child) + n.n_children > d.n_child_indices) return fail(RT_E_INVALID, "child range out of bounds");
            if (out.lights.size() >= (1u << kLightListShift) || n.n_children >= (1u << 11)) return fail(RT_E_UNSUPPORTED, "light list too large");
            out.lights[base + k].index = int32_t(out.lights.size() | (size_t(n.n_children) << kLightListShift));
            out.needs_tex_interpreter = true;  // the full-feature kernel variants evaluate nested light lists
            if (!compile_light_entries(d.child_indices + n.first_child, n.n_children, depth + 1)) return false;
        }
        return true;
    }

    bool compile_lights() {
        if (d.lights_root >= d.n_nodes) return fail(RT_E_INVALID, "lights root out of range");
        const RtNode& n = d.nodes[d.lights_root];
        if (n.type == RT_NODE_LIST) {
            out.lights_is_list = 1;
            if (uint64_t(n.first_child) + n.n_children > d.n_child_indices) return fail(RT_E_INVALID, "child range out of bounds");
            out.n_top_lights = int32_t(n.n_children);
            return compile_light_entries(d.child_indices + n.first_child, n.n_children, 0);
        }
        out.lights_is_list = 0;
        out.n_top_lights = 1;
        uint32_t root = d.lights_root;
        return compile_light_entries(&root, 1, 0);
    }

    // ---- CompiledScene::zero_weight_stop ----
    // A pdf-sampled vertex (Lambertian / Glossy diffuse lobe / Isotropic) gets an infinite or NaN weight only if
    // mix_pdf.value == 0 or the sampled direction is NaN.  The material's own pdf is > 0 for its own samples, so
    // this needs a LIGHT sample (mix.rs:23-36) whose direction no light's pdf_value covers, or a NaN from
    // Sphere::random (origin inside the sphere: sqrt of a negative number, sphere.rs:123-145).  Ruled out when
    //   * `lights` (and every list inside it) is non-empty                                       (list.rs:93-95),
    //   * every light is a two-sided quad (a one-sided quad's pdf_value is 0 from behind, plane.rs:74,107-118),
    //     a sphere, the sky or the sun (Transform / mesh / bvh / volume lights have pdf_value 0),
    //   * quads and spheres used as lights do not scatter with a pdf themselves (a point ON the light samples
    //     the light from its own surface: in-plane directions, |c - p| < r by rounding),
    //   * no pdf-scattering object of the world reaches inside a sphere light (conservative box test).
    // What remains are rounding-level events (a sampled point on the very edge of a quad or of a sphere's cone
    // failing the re-intersection): probability of the order of 2^-50 per light sample.
    bool scatters_with_pdf(int32_t material) const {
        if (material < 0 || uint32_t(material) >= d.n_materials) return true;
        const uint32_t t = d.materials[material].type;
        return t == RT_MAT_LAMBERTIAN || t == RT_MAT_GLOSSY || t == RT_MAT_ISOTROPIC;
    }
    bool subtree_scatters_with_pdf(uint32_t node, int depth) const {
        if (node >= d.n_nodes || depth > 512) return true;
        const RtNode& n = d.nodes[node];
        switch (n.type) {
            case RT_NODE_SPHERE: case RT_NODE_PLANE: case RT_NODE_MESH: case RT_NODE_SKY: case RT_NODE_SUN:
                return scatters_with_pdf(n.material);
            case RT_NODE_VOLUME:
                return true;  // Isotropic medium
            case RT_NODE_NULL:
                return false;
            default:
                for (uint32_t k = 0; k < n.n_children; k++)
                    if (subtree_scatters_with_pdf(d.child_indices[n.first_child + k], depth + 1)) return true;
                return false;
        }
    }
    // Can a pdf-scattering surface of the subtree `node` lie inside the ball (centre c, radius r, world space)?
    // `m` = object-to-world matrix of the space `node` lives in (row-major 3x4).  Boxes are the reference's
    // get_bounding_box() values, moved to world space corner by corner (a Transform node's own box is NOT used:
    // Transform::update_bounds has a quirk, transform.rs:98-118, SURVEY B-7; its child is entered instead).
    bool pdf_surface_inside_ball(uint32_t node, const double c[3], double r, const double m[12], int depth) const {
        if (node >= d.n_nodes || depth > 512) return true;
        const RtNode& n = d.nodes[node];
        if (n.type == RT_NODE_TRANSFORM) {
            if (n.n_children != 1 || n.transform < 0 || uint32_t(n.transform) >= d.n_transforms) return true;
            const double* t = d.transforms[n.transform].m;  // row-major 4x4, last row 0 0 0 1
            double mt[12];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 4; j++)
                    mt[4 * i + j] = m[4 * i] * t[j] + m[4 * i + 1] * t[4 + j] + m[4 * i + 2] * t[8 + j] + (j == 3 ? m[4 * i + 3] : 0.0);
            return pdf_surface_inside_ball(d.child_indices[n.first_child], c, r, mt, depth + 1);
        }
        if (n.type == RT_NODE_NULL) return false;
        if (n.type == RT_NODE_SKY || n.type == RT_NODE_SUN) return false;  // at infinity (their own materials are Emissive)
        double lo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, hi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
        for (int corner = 0; corner < 8; corner++) {
            const double p[3] = {n.bounds[(corner & 1) ? 3 : 0], n.bounds[(corner & 2) ? 4 : 1], n.bounds[(corner & 4) ? 5 : 2]};
            for (int a = 0; a < 3; a++) {
                const double w = m[4 * a] * p[0] + m[4 * a + 1] * p[1] + m[4 * a + 2] * p[2] + m[4 * a + 3];
                lo[a] = std::fmin(lo[a], w);
                hi[a] = std::fmax(hi[a], w);
            }
        }
        double d2 = 0.0;  // squared distance from the centre to the box
        for (int a = 0; a < 3; a++) {
            const double x = c[a] < lo[a] ? lo[a] - c[a] : (c[a] > hi[a] ? c[a] - hi[a] : 0.0);
            d2 += x * x;
        }
        if (d2 > r * r) return false;  // the box is outside the ball (NaN: undecided -> go on)
        if (n.type == RT_NODE_LIST || n.type == RT_NODE_BVH) {
            for (uint32_t k = 0; k < n.n_children; k++)
                if (pdf_surface_inside_ball(d.child_indices[n.first_child + k], c, r, m, depth + 1)) return true;
            return false;
        }
        return subtree_scatters_with_pdf(node, depth);
    }
    bool light_is_safe(uint32_t node, bool nested) const {
        const RtNode& n = d.nodes[node];
        switch (n.type) {
            case RT_NODE_PLANE:
                return (n.flags & RT_PLANE_RENDER_BACKFACE) != 0 && !scatters_with_pdf(n.material);
            case RT_NODE_SPHERE: {
                if (scatters_with_pdf(n.material)) return false;
                const double r = std::fabs(n.p[3]) * (1.0 + 1e-9);
                const double identity[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
                return !pdf_surface_inside_ball(d.world_root, n.p, r, identity, 0);
            }
            case RT_NODE_SKY: case RT_NODE_SUN:
                return true;
            case RT_NODE_LIST: {
                if (nested || n.n_children == 0) return false;
                for (uint32_t k = 0; k < n.n_children; k++)
                    if (!light_is_safe(d.child_indices[n.first_child + k], true)) return false;
                return true;
            }
            default:
                return false;
        }
    }
    void decide_zero_weight_stop() {
        const RtNode& n = d.nodes[d.lights_root];
        bool ok;
        if (n.type == RT_NODE_LIST) {
            ok = n.n_children > 0;
            for (uint32_t k = 0; ok && k < n.n_children; k++) ok = light_is_safe(d.child_indices[n.first_child + k], false);
        } else {
            ok = light_is_safe(d.lights_root, false);
        }
        out.zero_weight_stop = ok;
    }
};

}  // namespace

// The split intersect: k_wf_prims for the spheres / quads / sky / sun (and the volumes they bound) and, when the program has
// mesh ops (any number of instances), k_wf_mesh for those.  Scenes it does not cover use the combined kernel
// (k_wf_intersect): a mesh inside a volume's boundary, or in front of a volume in program order (the volume's free-flight
// draw depends on the closest hit so far, volume.rs:40-43, so that mesh cannot be deferred); more than 32 766 mesh ops
// (k_wf_mesh packs the mesh-op index in 15 bits).  Groups: their nodes (<= 24 KB) and a per-lane stack (<= 16 levels) must
// fit k_wf_prims' LDS; volumes and groups do not combine (k_wf_prims<VOL> walks the op form).
WavefrontPlan plan_wavefront(const CompiledScene& cs) {
    WavefrontPlan p{};
    const bool vol = !cs.volumes.empty();
    bool vol_ok = true;
    if (vol) {
        int32_t last_vol = -1;
        for (size_t i = 0; i < cs.ops.size(); i++)
            if (cs.ops[i].type == OP_VOL_END) last_vol = int32_t(i);
        vol_ok = cs.mesh_ops.empty() || cs.mesh_ops.front() > last_vol;
    }
    p.split = vol_ok && cs.mesh_ops.size() < 32767;
    p.vol_prims = p.split && vol;
    p.multi_mesh = cs.mesh_ops.size() > 1;
    p.groups = p.split && !vol && !cs.group_nodes4.empty() && cs.max_group_stack <= 16 && cs.group_nodes4.size() * 64 <= 24u * 1024u;
    return p;
}

int compile_scene(const RtSceneDesc* desc, CompiledScene* out, std::string* err, const CompileOptions& opt) {
    if (!desc || desc->abi_version != RT_MI355_ABI_VERSION) {
        *err = "scene description missing or ABI version mismatch";
        return RT_E_INVALID;
    }
    if ((desc->n_nodes && !desc->nodes) || (desc->n_child_indices && !desc->child_indices) ||
        (desc->n_transforms && !desc->transforms) || (desc->n_meshes && !desc->meshes) ||
        (desc->n_materials && !desc->materials) || (desc->n_textures && !desc->textures)) {
        *err = "scene description has NULL tables";
        return RT_E_INVALID;
    }
    *out = CompiledScene{};
    Compiler c{*desc, *out, err, opt};
    if (!c.compile_tables()) return c.status;
    if (!c.compile_node(desc->world_root, 0)) return c.status;
    c.chain.clear();
    out->ops.push_back({OP_END, 0, 0, c.chain_id()});
    for (size_t pc = 0; pc < out->ops.size(); pc++)
        if (out->ops[pc].type == OP_MESH) out->mesh_ops.push_back(int32_t(pc));
    if (!c.compile_lights()) return c.status;
    c.decide_zero_weight_stop();
    out->chain_offsets.push_back(0);
    for (auto& ch : c.chains) {
        out->chain_items.insert(out->chain_items.end(), ch.begin(), ch.end());
        out->chain_offsets.push_back(int32_t(out->chain_items.size()));
    }
    return RT_OK;
}

}  // namespace rt
