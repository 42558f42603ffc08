// bvh.h -- host-side builder of the acceleration structure the HIP kernels traverse.
//
// Replaces Bvh::new (pyrite/src/spatial/bvh.rs:13-155). The reference builds a binary tree with one item per leaf,
// 6 SAH buckets on the widest centroid axis, flattened pre-order with skip counts and traversed without near/far
// ordering. None of that layout survives here: the kernels want few, wide, aligned fetches and an ordered
// traversal, so this builder emits 64-byte nodes that hold BOTH children's boxes (one visit = one 64 B fetch =
// two slab tests), leaves of up to 4 primitives stored contiguously in leaf order, 16-bin SAH over all three
// axes, and a hard depth bound so the per-lane traversal stack in LDS can be sized from the tree itself.
// Closest-hit results do not depend on the tree (only exact-distance ties do).
#pragma once
#include <cstdint>
#include <vector>

namespace pyr {

struct PrimBounds {
    float lo[3], hi[3];
    uint32_t shape; // (PyrShapeKind << 30) | index
};

// Two-child node, 64 bytes, read by the kernels as four float4.
struct alignas(64) Node64 {
    float lo0[3];
    int32_t child0; // >= 0: node index; < 0: leaf, -1 - ((first_prim << 3) | count), count 0..4 (0 = empty)
    float hi0[3];
    int32_t child1;
    float lo1[3];
    uint32_t pad0;
    float hi1[3];
    uint32_t pad1;
};
static_assert(sizeof(Node64) == 64, "node must be 64 bytes");

struct BuiltBvh {
    std::vector<Node64> nodes;        // nodes[0] is the root
    std::vector<uint32_t> prim_order; // shape codes in leaf order
    uint32_t max_depth = 0;           // edges on the longest root-to-leaf path = stack entries an ordered traversal can need
    uint32_t num_leaves = 0;
};

constexpr uint32_t kMaxLeafPrims = 4;
constexpr uint32_t kMaxBvhDepth = 40;

inline int32_t encode_leaf(uint32_t first, uint32_t count) { return -1 - (int32_t)((first << 3) | count); }

BuiltBvh build_bvh(const std::vector<PrimBounds>& prims);

} // namespace pyr
