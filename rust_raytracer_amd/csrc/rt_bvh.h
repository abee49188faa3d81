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

// Leaves hold RECORDS of one or two triangles (rt_scene.h, TriPair): two triangles share a record when they form a
// fan (a, b, c), (a, c, d) in their OWN vertex order, so that the second one's first edge is the first one's second
// edge and both keep exactly the arithmetic of mesh.rs:69-70.  Triangle SLOT 2 r + s is triangle s of record r;
// the second slot of a single-triangle record is a hole.  Leaf reference: ~((first_slot << 4) | (n_slots - 1)),
// n_slots = 2 x records <= 16; inside a leaf the pair records come first.
constexpr uint32_t kHoleSlot = 0xFFFFFFFFu;
struct BvhBuild {
    std::vector<BuildNode> nodes;     // nodes[0] is the root
    std::vector<uint32_t> tri_order;  // slot -> original triangle index, kHoleSlot for a hole (size = 2 x records)
    uint32_t max_depth = 0;           // number of inner-node levels (bounds the traversal stack)
    uint32_t n_pair_records = 0;      // records that hold two triangles
};
inline int32_t leaf_ref_of(uint32_t first_slot, uint32_t n_slots) { return ~int32_t((first_slot << 4) | (n_slots - 1)); }

// positions: n_positions*3 doubles; tri_pos: n_tris*3 indices.  max_leaf in 1..8 (triangles per leaf).
BvhBuild build_bvh(const double* positions, const uint32_t* tri_pos, uint32_t n_tris, uint32_t max_leaf);

// Shared last step of both builders: `raw` has leaves ~((first << 3) | (count - 1)) over raw.tri_order (no holes);
// groups every leaf's triangles into records (fan pairs first, then singles) and rewrites order and references.
BvhBuild pack_leaf_records(BvhBuild raw, const uint32_t* tri_pos);

// The same contract, built on the current HIP device (rt_bvh_device.hip: Morton codes, radix sort, Karras radix
// tree, bottom-up box fit).  Returns false with `err` set on a device error.
bool build_bvh_device(const double* positions, uint32_t n_positions, const uint32_t* tri_pos, uint32_t n_tris, uint32_t max_leaf,
                      BvhBuild* out, std::string* err);

// 4-wide collapse of a BVH2 (same leaves, same triangle order): every node holds up to four
// children, obtained by repeatedly replacing the inner child of largest surface area by its two
// children.  One node fetch then decides four boxes: about half the dependent fetches per ray.
struct BuildNode4 {
    double lo[4][3], hi[4][3];
    int32_t child[4];  // inner index into nodes4 / kEmptyChild / leaf: ~((first_record << 6) | ((records - 1) << 3) | (triangles - 1))
};
struct Bvh4Build {
    std::vector<BuildNode4> nodes;  // nodes[0] is the root
    uint32_t max_depth = 0;         // inner levels
    uint32_t max_stack = 0;         // worst-case traversal stack entries (sum over a root-leaf path of children-1) + 1
    double root_lo[3], root_hi[3];
};
Bvh4Build collapse_bvh4(const BvhBuild& bvh2);

}  // namespace rt
