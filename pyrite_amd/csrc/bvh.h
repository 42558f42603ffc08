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
#include <climits>
#include <vector>

namespace pyr {

struct PrimBounds {
    float lo[3], hi[3];
    uint32_t shape; // (PyrShapeKind << 30) | index
};

// Two-child node, 64 bytes, read by the kernels as four float4. The two children's bounds are interleaved so that each
// float4 holds two (child 0, child 1) pairs: the kernels test both boxes with packed-fp32 instructions (v_pk_fma_f32) and a
// pair must sit in an aligned register pair. Boxes are padded by a few ulps of the scene's extent at build time
// (build_bvh), which makes the kernels' `t = bound * inv - origin * inv` form of the slab test conservative.
struct alignas(64) Node64 {
    float lo_x[2], lo_y[2]; // [child]
    float lo_z[2], hi_x[2];
    float hi_y[2], hi_z[2];
    int32_t child[2]; // >= 0: node index; < 0: leaf, -1 - ((first_prim << 3) | count), count 0..4 (0 = empty)
    uint32_t pad[2];
};
static_assert(sizeof(Node64) == 64, "node must be 64 bytes");

struct BuiltBvh {
    std::vector<Node64> nodes;        // nodes[0] is the root
    std::vector<uint32_t> prim_order; // shape codes in leaf order
    uint32_t max_depth = 0;           // edges on the longest root-to-leaf path = stack entries an ordered traversal can need
    uint32_t num_leaves = 0;
};

#ifndef PYR_MAX_LEAF
#define PYR_MAX_LEAF 4
#endif
constexpr uint32_t kMaxLeafPrims = PYR_MAX_LEAF; // <= 7: the leaf code keeps the count in 3 bits
constexpr uint32_t kMaxBvhDepth = 40;
// The SAH counts a leaf's primitives in pairs (an odd one costs a whole test): the render kernels test the triangles of a
// leaf two per step (device_scene.h DevPrimPair). Measured against counting singly: C3 451 -> 457, C5 393 -> 397 Msamples/s
// with the same number of triangle tests (fewer steps); C2's 71-node tree does not change.
#ifndef PYR_SAH_PAIRS
#define PYR_SAH_PAIRS 1
#endif
#ifndef PYR_SAH_NODE_COST
#define PYR_SAH_NODE_COST 1.0f
#endif
constexpr float kSahNodeCost = PYR_SAH_NODE_COST; // cost of one node visit in units of one primitive test (SAH termination)

inline int32_t encode_leaf(uint32_t first, uint32_t count) { return -1 - (int32_t)((first << 3) | count); }

// `leaves_tested_in_pairs`: the tree's leaves will be tested two triangles per step (the four-child pair tree of a
// triangle-only scene that does not live in LDS), so the SAH counts a leaf of n primitives as n rounded up to even -- only
// when PYR_SAH_PAIRS is on; everything else (sphere scenes, LDS-resident scenes, the binary walk) counts singly.
BuiltBvh build_bvh(const std::vector<PrimBounds>& prims, bool leaves_tested_in_pairs = false);

// Four-child node, 128 bytes = one L2 line, read as eight float4: lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4] hi.z[4] child[4]
// pad[4]. Built by collapsing the binary tree (the child with the largest surface area is replaced by its own two children
// until the node has four): a ray then makes about half as many dependent node fetches. Used by the resumable traversal
// on scenes that do not live in LDS, where the walk is latency bound; unused slots hold kEmptyChild and are never entered.
struct alignas(128) Node128 {
    float lo_x[4], lo_y[4], lo_z[4], hi_x[4], hi_y[4], hi_z[4];
    int32_t child[4]; // >= 0: Node128 index; < 0: leaf code as in Node64; kEmptyChild: nothing
    uint32_t pad[4];
};
static_assert(sizeof(Node128) == 128, "wide node must be 128 bytes");
constexpr int32_t kEmptyChild = INT32_MIN;

struct WideBvh {
    std::vector<Node128> nodes; // nodes[0] is the root
    uint32_t max_depth = 0;     // node levels on the longest path
    uint32_t stack_need = 0;    // entries an ordered traversal can hold at once: max over paths of sum(children - 1)
};
WideBvh collapse_to_wide(const BuiltBvh& bvh);

} // namespace pyr
