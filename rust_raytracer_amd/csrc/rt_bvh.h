// rt_bvh.h — host-side binned-SAH BVH2 builder for one triangle mesh.
//
// Replaces the reference's octree (src/object/mesh/octree.rs:31-210: midpoint split into 8
// octants, <= 50 triangles per leaf, triangles duplicated into every octant they touch,
// children visited in fixed order).  Any structure that returns the same closest hit is
// admissible: the octree only culls; ties between different triangles at exactly
// equal t may resolve to a different (adjacent) triangle.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace rt {

struct BuildNode {
    double lo0[3], hi0[3], lo1[3], hi1[3];
    int32_t c0, c1;  // encoding: see BvhNode in rt_scene.h
};

struct BvhBuild {
    std::vector<BuildNode> nodes;     // nodes[0] is the root
    std::vector<uint32_t> tri_order;  // leaf order -> original triangle index
    uint32_t max_depth = 0;           // number of inner-node levels (bounds the traversal stack)
};

// positions: n_positions*3 doubles; tri_pos: n_tris*3 indices.  max_leaf in 1..8.
BvhBuild build_bvh(const double* positions, const uint32_t* tri_pos, uint32_t n_tris, uint32_t max_leaf);

// The same contract, built on the current HIP device (rt_bvh_device.hip: Morton codes, radix sort, Karras radix
// tree, bottom-up box fit).  Returns false with `err` set on a device error.
bool build_bvh_device(const double* positions, uint32_t n_positions, const uint32_t* tri_pos, uint32_t n_tris, uint32_t max_leaf,
                      BvhBuild* out, std::string* err);

// W-wide collapse of a BVH2 (same leaves, same triangle order): every node holds up to W children,
// obtained by repeatedly replacing the inner child of largest surface area by its two children.  One node
// fetch then decides W boxes: about half the dependent fetches per ray for W = 4 (W = 8 was tried with quantised
// nodes and was slower: profiles/r02/ab/node_width_and_size.txt).
template <int W>
struct BuildNodeN {
    double lo[W][3], hi[W][3];
    int32_t child[W];  // same encoding as BuildNode::c0 (inner index into the wide node array / leaf code / kEmptyChild)
};
template <int W>
struct BvhNBuild {
    std::vector<BuildNodeN<W>> nodes;  // nodes[0] is the root
    uint32_t max_depth = 0;            // inner levels
    uint32_t max_stack = 0;            // worst-case traversal stack entries (sum over a root-leaf path of children-1) + 1
    double root_lo[3], root_hi[3];
};
using BuildNode4 = BuildNodeN<4>;
using Bvh4Build = BvhNBuild<4>;
Bvh4Build collapse_bvh4(const BvhBuild& bvh2);

}  // namespace rt
