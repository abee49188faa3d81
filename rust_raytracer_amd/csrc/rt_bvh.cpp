// Binned-SAH BVH2 builder (host, f64).  See rt_bvh.h.
#include "rt_bvh.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <unordered_map>

#include "rt_scene.h"

namespace rt {
namespace {

constexpr int kMaxBins = 64;
static int g_bins = 16;  // RT_BVH_BINS (experiments)
constexpr double kInf = std::numeric_limits<double>::infinity();

struct Box {
    double lo[3] = {kInf, kInf, kInf};
    double hi[3] = {-kInf, -kInf, -kInf};
    void grow(const double* p) {
        for (int a = 0; a < 3; a++) {
            if (p[a] < lo[a]) lo[a] = p[a];
            if (p[a] > hi[a]) hi[a] = p[a];
        }
    }
    void grow(const Box& b) {
        for (int a = 0; a < 3; a++) {
            if (b.lo[a] < lo[a]) lo[a] = b.lo[a];
            if (b.hi[a] > hi[a]) hi[a] = b.hi[a];
        }
    }
    double half_area() const {
        double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0 && dy >= 0 && dz >= 0)) return 0.0;
        return dx * dy + dy * dz + dz * dx;
    }
};

// Primitives of the SAH build are UNITS: a fan pair of triangles (kept together so that they can share a record)
// or a single triangle.  Costs and leaf sizes count triangles.
struct Builder {
    std::vector<Box> prim_box;
    std::vector<double> centroid;    // n_prims*3
    std::vector<uint32_t> weight;    // triangles of the unit (1 or 2)
    std::vector<uint32_t> order;     // permutation of the units
    std::vector<BuildNode> nodes;
    uint32_t max_leaf;
    uint32_t max_depth = 0;

    static int32_t raw_leaf_ref(uint32_t first, uint32_t count) { return ~int32_t((first << 3) | (count - 1)); }

    // Builds the subtree over order[begin, end) whose triangles occupy [tri_begin, ...) of the expanded triangle
    // order; returns the child reference and its box.
    int32_t build(uint32_t begin, uint32_t end, uint32_t tri_begin, Box* out_box, uint32_t depth) {
        Box box, cbox;
        uint32_t n_tris = 0;
        for (uint32_t i = begin; i < end; i++) {
            box.grow(prim_box[order[i]]);
            cbox.grow(&centroid[3 * order[i]]);
            n_tris += weight[order[i]];
        }
        *out_box = box;
        uint32_t n = end - begin;
        if (n_tris <= max_leaf || n == 1) return raw_leaf_ref(tri_begin, n_tris);

        // pick axis/plane by binned SAH over centroid bounds
        int best_axis = -1, best_bin = -1;
        double best_cost = kInf;
        for (int a = 0; a < 3; a++) {
            double ext = cbox.hi[a] - cbox.lo[a];
            if (!(ext > 0.0) || !std::isfinite(ext)) continue;
            const int kBins = g_bins;
            Box bins[kMaxBins];
            uint32_t counts[kMaxBins] = {0};
            double scale = double(kBins) / ext;
            for (uint32_t i = begin; i < end; i++) {
                int b = int((centroid[3 * order[i] + a] - cbox.lo[a]) * scale);
                if (b < 0) b = 0;
                if (b >= kBins) b = kBins - 1;
                bins[b].grow(prim_box[order[i]]);
                counts[b] += weight[order[i]];
            }
            double right_area[kMaxBins];
            uint32_t right_count[kMaxBins];
            Box acc;
            uint32_t cnt = 0;
            for (int b = kBins - 1; b > 0; b--) {
                acc.grow(bins[b]);
                cnt += counts[b];
                right_area[b] = acc.half_area();
                right_count[b] = cnt;
            }
            Box lacc;
            uint32_t lcnt = 0;
            for (int b = 0; b < kBins - 1; b++) {
                lacc.grow(bins[b]);
                lcnt += counts[b];
                if (lcnt == 0 || right_count[b + 1] == 0) continue;
                double cost = lacc.half_area() * double(lcnt) + right_area[b + 1] * double(right_count[b + 1]);
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = a;
                    best_bin = b;
                }
            }
        }
        uint32_t mid;
        if (best_axis < 0) {
            mid = begin + n / 2;  // all centroids coincide (or non-finite): split by index
        } else {
            const int kBins = g_bins;
            double ext = cbox.hi[best_axis] - cbox.lo[best_axis];
            double scale = double(kBins) / ext;
            double lo = cbox.lo[best_axis];
            auto it = std::partition(order.begin() + begin, order.begin() + end, [&](uint32_t t) {
                int b = int((centroid[3 * t + best_axis] - lo) * scale);
                if (b < 0) b = 0;
                if (b >= kBins) b = kBins - 1;
                return b <= best_bin;
            });
            mid = uint32_t(it - order.begin());
            if (mid == begin || mid == end) mid = begin + n / 2;
        }
        uint32_t left_tris = 0;
        for (uint32_t i = begin; i < mid; i++) left_tris += weight[order[i]];
        uint32_t idx = uint32_t(nodes.size());
        nodes.emplace_back();
        if (depth + 1 > max_depth) max_depth = depth + 1;
        Box b0, b1;
        int32_t c0 = build(begin, mid, tri_begin, &b0, depth + 1);
        int32_t c1 = build(mid, end, tri_begin + left_tris, &b1, depth + 1);
        BuildNode& nd = nodes[idx];
        for (int a = 0; a < 3; a++) {
            nd.lo0[a] = b0.lo[a]; nd.hi0[a] = b0.hi[a];
            nd.lo1[a] = b1.lo[a]; nd.hi1[a] = b1.hi[a];
        }
        nd.c0 = c0;
        nd.c1 = c1;
        return int32_t(idx);
    }
};

// Is (a, b) a fan pair in the triangles' own vertex order: a = (p, q, r), b = (p, r, s)?
inline bool fan_pair(const uint32_t* tri_pos, uint32_t a, uint32_t b) {
    const uint32_t* A = tri_pos + 3 * size_t(a);
    const uint32_t* B = tri_pos + 3 * size_t(b);
    return a != b && A[0] == B[0] && A[2] == B[1];
}

}  // namespace

BvhBuild pack_leaf_records(BvhBuild raw, const uint32_t* tri_pos) {
    BvhBuild out;
    out.max_depth = raw.max_depth;
    out.tri_order.reserve(raw.tri_order.size() + raw.tri_order.size() / 8);
    auto pack = [&](int32_t ref) -> int32_t {
        if (ref >= 0 || ref == kEmptyChild) return ref;
        const uint32_t code = uint32_t(~ref);
        const uint32_t first = code >> 3, count = (code & 7u) + 1u;
        const uint32_t* t = raw.tri_order.data() + first;
        bool used[8] = {false, false, false, false, false, false, false, false};
        const uint32_t first_slot = uint32_t(out.tri_order.size());
        for (uint32_t i = 0; i < count; i++) {  // pair records first
            if (used[i]) continue;
            for (uint32_t j = 0; j < count; j++) {
                if (used[j] || j == i) continue;
                uint32_t a = kHoleSlot, b = kHoleSlot;
                if (fan_pair(tri_pos, t[i], t[j])) { a = t[i]; b = t[j]; }
                else if (fan_pair(tri_pos, t[j], t[i])) { a = t[j]; b = t[i]; }
                if (a == kHoleSlot) continue;
                used[i] = used[j] = true;
                out.tri_order.push_back(a);
                out.tri_order.push_back(b);
                out.n_pair_records++;
                break;
            }
        }
        for (uint32_t i = 0; i < count; i++)
            if (!used[i]) {
                out.tri_order.push_back(t[i]);
                out.tri_order.push_back(kHoleSlot);
            }
        return leaf_ref_of(first_slot, uint32_t(out.tri_order.size()) - first_slot);
    };
    out.nodes = std::move(raw.nodes);
    for (BuildNode& n : out.nodes) {
        n.c0 = pack(n.c0);
        n.c1 = pack(n.c1);
    }
    return out;
}

BvhBuild build_bvh(const double* positions, const uint32_t* tri_pos, uint32_t n_tris, uint32_t max_leaf) {
    if (const char* e = std::getenv("RT_BVH_BINS")) { int v = std::atoi(e); if (v >= 2 && v <= kMaxBins) g_bins = v; }
    Builder b;
    b.max_leaf = std::min<uint32_t>(std::max<uint32_t>(max_leaf, 1), 8);
    // units: greedy fan pairs over the whole mesh (partner looked up by its first edge), the rest single
    std::vector<uint32_t> unit_first, unit_second;  // triangle indices; unit_second = kHoleSlot for a single
    {
        const bool pairs = !(std::getenv("RT_BVH_PAIRS") && std::atoi(std::getenv("RT_BVH_PAIRS")) == 0) && b.max_leaf >= 2;
        std::vector<uint8_t> used(n_tris, 0);
        // (v0, v1) -> triangles that can be the SECOND of a fan, (v0, v2) -> triangles that can be the FIRST
        std::unordered_map<uint64_t, std::vector<uint32_t>> by_first_edge, by_second_edge;
        auto key = [&](uint32_t t, int k) { return (uint64_t(tri_pos[3 * size_t(t)]) << 32) | tri_pos[3 * size_t(t) + k]; };
        if (pairs) {
            by_first_edge.reserve(size_t(n_tris) * 2);
            by_second_edge.reserve(size_t(n_tris) * 2);
            for (uint32_t t = 0; t < n_tris; t++) {
                by_first_edge[key(t, 1)].push_back(t);
                by_second_edge[key(t, 2)].push_back(t);
            }
        }
        for (uint32_t t = 0; t < n_tris; t++) {
            if (used[t]) continue;
            used[t] = 1;
            uint32_t first = t, second = kHoleSlot;
            if (pairs) {
                auto it = by_first_edge.find(key(t, 2));  // t = (a, b, c): a partner (a, c, d)
                if (it != by_first_edge.end())
                    for (uint32_t c : it->second)
                        if (!used[c] && fan_pair(tri_pos, t, c)) { second = c; break; }
                if (second == kHoleSlot) {  // t = (a, c, d): a partner (a, b, c) in front of it
                    auto jt = by_second_edge.find(key(t, 1));
                    if (jt != by_second_edge.end())
                        for (uint32_t c : jt->second)
                            if (!used[c] && fan_pair(tri_pos, c, t)) { first = c; second = t; break; }
                }
            }
            if (second != kHoleSlot) used[first] = used[second] = 1;
            unit_first.push_back(first);
            unit_second.push_back(second);
        }
    }
    const uint32_t n_units = uint32_t(unit_first.size());
    b.prim_box.resize(n_units);
    b.centroid.resize(size_t(n_units) * 3);
    b.weight.resize(n_units);
    b.order.resize(n_units);
    for (uint32_t u = 0; u < n_units; u++) {
        b.order[u] = u;
        Box bx;
        for (int k = 0; k < 3; k++) bx.grow(positions + 3 * size_t(tri_pos[3 * size_t(unit_first[u]) + k]));
        if (unit_second[u] != kHoleSlot)
            for (int k = 0; k < 3; k++) bx.grow(positions + 3 * size_t(tri_pos[3 * size_t(unit_second[u]) + k]));
        b.prim_box[u] = bx;
        b.weight[u] = unit_second[u] != kHoleSlot ? 2u : 1u;
        for (int a = 0; a < 3; a++) b.centroid[3 * size_t(u) + a] = 0.5 * (bx.lo[a] + bx.hi[a]);
    }
    b.nodes.reserve(n_tris / 2 + 4);
    BvhBuild raw;
    if (n_tris <= b.max_leaf) {
        // Tiny mesh: a root whose first child is the only leaf.
        BuildNode root{};
        Box bx;
        for (uint32_t u = 0; u < n_units; u++) bx.grow(b.prim_box[u]);
        for (int a = 0; a < 3; a++) {
            root.lo0[a] = bx.lo[a]; root.hi0[a] = bx.hi[a];
            root.lo1[a] = kInf; root.hi1[a] = -kInf;
        }
        root.c0 = n_tris ? Builder::raw_leaf_ref(0, n_tris) : kEmptyChild;
        root.c1 = kEmptyChild;
        raw.nodes.push_back(root);
        raw.max_depth = 1;
    } else {
        Box root_box;
        b.build(0, n_units, 0, &root_box, 0);  // more triangles than a leaf holds: the root is an inner node at index 0
        raw.nodes = std::move(b.nodes);
        raw.max_depth = b.max_depth;
    }
    raw.tri_order.reserve(n_tris);
    for (uint32_t u : b.order) {
        raw.tri_order.push_back(unit_first[u]);
        if (unit_second[u] != kHoleSlot) raw.tri_order.push_back(unit_second[u]);
    }
    return pack_leaf_records(std::move(raw), tri_pos);
}


namespace {
struct Cand {
    int32_t ref;
    double lo[3], hi[3];
    double area() const {
        double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0 && dy >= 0 && dz >= 0)) return -1.0;
        return dx * dy + dy * dz + dz * dx;
    }
};
struct Collapser {
    const BvhBuild& b2;
    Bvh4Build out;
    // BVH2 leaf reference (slots) -> BVH4 leaf reference (records and triangles)
    int32_t leaf4(int32_t ref2) const {
        const uint32_t code = uint32_t(~ref2);
        const uint32_t first_slot = code >> 4, n_slots = (code & 15u) + 1u;
        uint32_t n_tris = 0;
        for (uint32_t k = 0; k < n_slots; k++) n_tris += b2.tri_order[first_slot + k] != kHoleSlot;
        return ~int32_t(((first_slot / 2) << 6) | ((n_slots / 2 - 1) << 3) | (n_tris - 1));
    }
    uint32_t build(int32_t n2, uint32_t depth, uint32_t stack_above) {
        uint32_t idx = uint32_t(out.nodes.size());
        out.nodes.emplace_back();
        if (depth + 1 > out.max_depth) out.max_depth = depth + 1;
        std::vector<Cand> c;
        auto add_children = [&](int32_t node2) {
            const BuildNode& n = b2.nodes[size_t(node2)];
            Cand a{n.c0, {n.lo0[0], n.lo0[1], n.lo0[2]}, {n.hi0[0], n.hi0[1], n.hi0[2]}};
            Cand b{n.c1, {n.lo1[0], n.lo1[1], n.lo1[2]}, {n.hi1[0], n.hi1[1], n.hi1[2]}};
            if (a.ref != kEmptyChild) c.push_back(a);
            if (b.ref != kEmptyChild) c.push_back(b);
        };
        add_children(n2);
        for (;;) {
            if (c.size() >= 4) break;
            int best = -1;
            double best_area = -1.0;
            for (size_t i = 0; i < c.size(); i++)
                if (c[i].ref >= 0 && c[i].area() > best_area) { best_area = c[i].area(); best = int(i); }
            if (best < 0) break;  // only leaves left
            int32_t expand = c[size_t(best)].ref;
            c.erase(c.begin() + best);
            add_children(expand);
        }
        uint32_t n_children = uint32_t(c.size());
        uint32_t stack_here = stack_above + (n_children > 0 ? n_children - 1 : 0);
        if (stack_here + 1 > out.max_stack) out.max_stack = stack_here + 1;
        BuildNode4 node{};
        for (int k = 0; k < 4; k++) {
            node.child[k] = kEmptyChild;
            for (int a = 0; a < 3; a++) { node.lo[k][a] = kInf; node.hi[k][a] = -kInf; }
        }
        for (uint32_t k = 0; k < n_children; k++) {
            for (int a = 0; a < 3; a++) { node.lo[k][a] = c[k].lo[a]; node.hi[k][a] = c[k].hi[a]; }
            node.child[k] = c[k].ref >= 0 ? c[k].ref : leaf4(c[k].ref);  // inner refs are patched below
        }
        out.nodes[idx] = node;
        for (uint32_t k = 0; k < n_children; k++)
            if (c[k].ref >= 0) {
                uint32_t child_idx = build(c[k].ref, depth + 1, stack_here);
                out.nodes[idx].child[k] = int32_t(child_idx);
            }
        return idx;
    }
};
}  // namespace

Bvh4Build collapse_bvh4(const BvhBuild& b2) {
    Collapser c{b2, {}};
    c.out.nodes.reserve(b2.nodes.size() / 2 + 4);
    c.build(0, 0, 0);
    for (int a = 0; a < 3; a++) { c.out.root_lo[a] = kInf; c.out.root_hi[a] = -kInf; }
    const BuildNode4& r = c.out.nodes[0];
    for (int k = 0; k < 4; k++)
        if (r.child[k] != kEmptyChild)
            for (int a = 0; a < 3; a++) {
                c.out.root_lo[a] = std::min(c.out.root_lo[a], r.lo[k][a]);
                c.out.root_hi[a] = std::max(c.out.root_hi[a], r.hi[k][a]);
            }
    return std::move(c.out);
}

}  // namespace rt
