// Binned-SAH BVH2 builder (host, f64).  See rt_bvh.h.
#include "rt_bvh.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>

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

struct Builder {
    std::vector<Box> tri_box;
    std::vector<double> centroid;  // n*3
    std::vector<uint32_t> order;
    std::vector<BuildNode> nodes;
    uint32_t max_leaf;
    uint32_t max_depth = 0;

    static int32_t leaf_ref(uint32_t first, uint32_t count) { return ~int32_t((first << 3) | (count - 1)); }

    // Builds the subtree over order[begin, end); returns the child reference and its box.
    int32_t build(uint32_t begin, uint32_t end, Box* out_box, uint32_t depth) {
        Box box, cbox;
        for (uint32_t i = begin; i < end; i++) {
            box.grow(tri_box[order[i]]);
            cbox.grow(&centroid[3 * order[i]]);
        }
        *out_box = box;
        uint32_t n = end - begin;
        if (n <= max_leaf) return leaf_ref(begin, n);

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
                bins[b].grow(tri_box[order[i]]);
                counts[b]++;
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
        uint32_t idx = uint32_t(nodes.size());
        nodes.emplace_back();
        if (depth + 1 > max_depth) max_depth = depth + 1;
        Box b0, b1;
        int32_t c0 = build(begin, mid, &b0, depth + 1);
        int32_t c1 = build(mid, end, &b1, depth + 1);
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

}  // namespace

BvhBuild build_bvh(const double* positions, const uint32_t* tri_pos, uint32_t n_tris, uint32_t max_leaf) {
    if (const char* e = std::getenv("RT_BVH_BINS")) { int v = std::atoi(e); if (v >= 2 && v <= kMaxBins) g_bins = v; }
    Builder b;
    b.max_leaf = std::min<uint32_t>(std::max<uint32_t>(max_leaf, 1), 8);
    b.tri_box.resize(n_tris);
    b.centroid.resize(size_t(n_tris) * 3);
    b.order.resize(n_tris);
    for (uint32_t t = 0; t < n_tris; t++) {
        b.order[t] = t;
        Box bx;
        for (int k = 0; k < 3; k++) bx.grow(positions + 3 * size_t(tri_pos[3 * size_t(t) + k]));
        b.tri_box[t] = bx;
        for (int a = 0; a < 3; a++) b.centroid[3 * size_t(t) + a] = 0.5 * (bx.lo[a] + bx.hi[a]);
    }
    b.nodes.reserve(n_tris / 2 + 4);
    BvhBuild out;
    if (n_tris <= b.max_leaf) {
        // Tiny mesh: a root whose first child is the only leaf.
        BuildNode root{};
        Box bx;
        for (uint32_t t = 0; t < n_tris; t++) bx.grow(b.tri_box[t]);
        for (int a = 0; a < 3; a++) {
            root.lo0[a] = bx.lo[a]; root.hi0[a] = bx.hi[a];
            root.lo1[a] = kInf; root.hi1[a] = -kInf;
        }
        root.c0 = n_tris ? Builder::leaf_ref(0, n_tris) : kEmptyChild;
        root.c1 = kEmptyChild;
        out.nodes.push_back(root);
        out.max_depth = 1;
    } else {
        Box root_box;
        b.build(0, n_tris, &root_box, 0);  // n > max_leaf: the root is an inner node at index 0
        out.nodes = std::move(b.nodes);
        out.max_depth = b.max_depth;
    }
    out.tri_order = std::move(b.order);
    return out;
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
template <int W>
struct Collapser {
    const BvhBuild& b2;
    BvhNBuild<W> out;
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
            if (c.size() >= size_t(W)) break;
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
        BuildNodeN<W> node{};
        for (int k = 0; k < W; k++) {
            node.child[k] = kEmptyChild;
            for (int a = 0; a < 3; a++) { node.lo[k][a] = kInf; node.hi[k][a] = -kInf; }
        }
        for (uint32_t k = 0; k < n_children; k++) {
            for (int a = 0; a < 3; a++) { node.lo[k][a] = c[k].lo[a]; node.hi[k][a] = c[k].hi[a]; }
            node.child[k] = c[k].ref;  // inner refs are patched below
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

namespace {
template <int W>
BvhNBuild<W> collapse(const BvhBuild& b2) {
    Collapser<W> c{b2, {}};
    c.out.nodes.reserve(b2.nodes.size() / (W == 4 ? 2 : 3) + 4);
    c.build(0, 0, 0);
    for (int a = 0; a < 3; a++) { c.out.root_lo[a] = kInf; c.out.root_hi[a] = -kInf; }
    const BuildNodeN<W>& r = c.out.nodes[0];
    for (int k = 0; k < W; k++)
        if (r.child[k] != kEmptyChild)
            for (int a = 0; a < 3; a++) {
                c.out.root_lo[a] = std::min(c.out.root_lo[a], r.lo[k][a]);
                c.out.root_hi[a] = std::max(c.out.root_hi[a], r.hi[k][a]);
            }
    return std::move(c.out);
}
}  // namespace

Bvh4Build collapse_bvh4(const BvhBuild& b2) { return collapse<4>(b2); }

}  // namespace rt
