// bvh.cpp -- binned-SAH builder for the 64-byte two-child node layout (see bvh.h).
#include "bvh.h"

#include <algorithm>
#include <cmath>
#include <limits>

namespace pyr {
namespace {

#ifndef PYR_SAH_BINS
#define PYR_SAH_BINS 16
#endif
constexpr int kBins = PYR_SAH_BINS;
constexpr float kInf = std::numeric_limits<float>::infinity();

struct Box {
    float lo[3] = {kInf, kInf, kInf};
    float hi[3] = {-kInf, -kInf, -kInf};
    void grow(const float* l, const float* h) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], l[a]);
            hi[a] = std::max(hi[a], h[a]);
        }
    }
    void grow(const Box& b) { grow(b.lo, b.hi); }
    void grow_point(const float* p) { grow(p, p); }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
        return dx * dy + dx * dz + dy * dz;
    }
};

struct Ref {
    float lo[3], hi[3], c[3];
    uint32_t shape;
};

struct Task {
    uint32_t begin, end, depth;
    int32_t parent; // node that receives this subtree, -1 for the root
    int slot;       // which child of `parent`
};

inline uint32_t ceil_log2(uint32_t n) {
    uint32_t l = 0;
    while ((1u << l) < n) ++l;
    return l;
}

} // namespace

BuiltBvh build_bvh(const std::vector<PrimBounds>& prims, bool leaves_tested_in_pairs) {
    BuiltBvh out;
    const uint32_t n = (uint32_t)prims.size();
    std::vector<Ref> refs(n);
    for (uint32_t i = 0; i < n; ++i) {
        for (int a = 0; a < 3; ++a) {
            refs[i].lo[a] = prims[i].lo[a];
            refs[i].hi[a] = prims[i].hi[a];
            refs[i].c[a] = 0.5f * prims[i].lo[a] + 0.5f * prims[i].hi[a];
        }
        refs[i].shape = prims[i].shape;
    }

    // Padding of every stored box: 16 ulps of the largest coordinate in the scene. The kernels compute slab distances as
    // fma(bound, 1/d, -(o * 1/d)), whose error is half an ulp of |o / d| -- in world units half an ulp of the ray origin,
    // which lies inside the scene -- so a padded box is never missed by a ray that hits something inside the exact box.
    float max_abs = 0.0f;
    for (uint32_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) max_abs = std::max(max_abs, std::max(std::fabs(prims[i].lo[a]), std::fabs(prims[i].hi[a])));
    const float pad = 16.0f * 1.1920929e-7f * max_abs;

    auto set_child = [&](int32_t parent, int slot, int32_t code, const Box& box) {
        Node64& p = out.nodes[parent];
        p.child[slot] = code;
        p.lo_x[slot] = box.lo[0] - pad, p.lo_y[slot] = box.lo[1] - pad, p.lo_z[slot] = box.lo[2] - pad;
        p.hi_x[slot] = box.hi[0] + pad, p.hi_y[slot] = box.hi[1] + pad, p.hi_z[slot] = box.hi[2] + pad;
    };
    auto empty_node = []() {
        Node64 nd{};
        for (int c = 0; c < 2; ++c) {
            nd.lo_x[c] = nd.lo_y[c] = nd.lo_z[c] = kInf; // an empty child leads to an empty leaf at worst
            nd.hi_x[c] = nd.hi_y[c] = nd.hi_z[c] = -kInf;
            nd.child[c] = encode_leaf(0, 0);
        }
        return nd;
    };

    // The root is always a node; a scene with <= kMaxLeafPrims primitives hangs one leaf under it.
    out.nodes.push_back(empty_node());
    if (n == 0) return out;

    auto make_leaf = [&](const Task& t, const Box& box) {
        uint32_t first = (uint32_t)out.prim_order.size();
        for (uint32_t i = t.begin; i < t.end; ++i) out.prim_order.push_back(refs[i].shape);
        set_child(t.parent, t.slot, encode_leaf(first, t.end - t.begin), box);
        out.num_leaves++;
        out.max_depth = std::max(out.max_depth, t.depth);
    };

    std::vector<Task> stack;
    // Root split: children of node 0. Handle it by treating the whole range as a task whose result is written
    // into a virtual parent; simpler: split the root range here, in the same code path as every other inner node.
    struct Pending {
        Task task;
    };
    // Task semantics: build the subtree for [begin,end) and store it as child `slot` of `parent`.
    // Depth counts edges from the root node; children of the root sit at depth 1.
    auto bounds_of = [&](uint32_t b, uint32_t e, Box& box, Box& cbox) {
        for (uint32_t i = b; i < e; ++i) {
            box.grow(refs[i].lo, refs[i].hi);
            cbox.grow_point(refs[i].c);
        }
    };

    // What testing n primitives of one leaf costs, in primitive tests. PYR_SAH_PAIRS=1: the render kernels test a leaf's
    // triangles two per step (DevPrimPair), so an odd triangle costs a whole step.
    const bool in_pairs = PYR_SAH_PAIRS && leaves_tested_in_pairs;
    auto leaf_tests = [in_pairs](uint32_t count) { return in_pairs ? (float)((count + 1u) & ~1u) : (float)count; };
    // Splits [begin,end) and returns mid; false when the range should become a leaf.
    auto split = [&](uint32_t begin, uint32_t end, uint32_t depth, const Box& box, const Box& cbox, uint32_t& mid) -> bool {
        uint32_t count = end - begin;
        if (count <= 1) return false;
        // Depth bound: once the remaining budget only just fits a balanced tree, split at the median.
        bool force_median = depth + ceil_log2((count + kMaxLeafPrims - 1) / kMaxLeafPrims) + 1 >= kMaxBvhDepth;
        float best_cost = kInf;
        int best_axis = -1, best_bin = -1;
        if (!force_median) {
            for (int a = 0; a < 3; ++a) {
                float extent = cbox.hi[a] - cbox.lo[a];
                if (!(extent > 0.0f)) continue;
                Box bin_box[kBins];
                uint32_t bin_count[kBins] = {0};
                float scale = (float)kBins / extent;
                for (uint32_t i = begin; i < end; ++i) {
                    int b = std::min(kBins - 1, std::max(0, (int)((refs[i].c[a] - cbox.lo[a]) * scale)));
                    bin_box[b].grow(refs[i].lo, refs[i].hi);
                    bin_count[b]++;
                }
                float right_area[kBins];
                uint32_t right_count[kBins];
                Box acc;
                uint32_t cnt = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    acc.grow(bin_box[b]);
                    cnt += bin_count[b];
                    right_area[b] = acc.half_area();
                    right_count[b] = cnt;
                }
                Box left;
                uint32_t lcnt = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    left.grow(bin_box[b]);
                    lcnt += bin_count[b];
                    if (lcnt == 0 || right_count[b + 1] == 0) continue;
                    float cost = left.half_area() * leaf_tests(lcnt) + right_area[b + 1] * leaf_tests(right_count[b + 1]);
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_axis = a;
                        best_bin = b;
                    }
                }
            }
        }
        if (best_axis >= 0) {
            // SAH termination: a leaf costs `count` primitive tests, a split costs one node visit plus the children.
            float parent_area = box.half_area();
            float split_cost = kSahNodeCost + (parent_area > 0.0f ? best_cost / parent_area : kInf);
            if (count <= kMaxLeafPrims && leaf_tests(count) <= split_cost) return false;
            float extent = cbox.hi[best_axis] - cbox.lo[best_axis];
            float scale = (float)kBins / extent;
            float lo = cbox.lo[best_axis];
            auto it = std::partition(refs.begin() + begin, refs.begin() + end, [&](const Ref& r) {
                int b = std::min(kBins - 1, std::max(0, (int)((r.c[best_axis] - lo) * scale)));
                return b <= best_bin;
            });
            mid = (uint32_t)(it - refs.begin());
            if (mid > begin && mid < end) return true;
        }
        if (count <= kMaxLeafPrims) return false;
        // Median split on the widest centroid axis (coincident centroids, or depth budget exhausted).
        int a = 0;
        float w = -1.0f;
        for (int k = 0; k < 3; ++k) {
            float e = cbox.hi[k] - cbox.lo[k];
            if (e > w) {
                w = e;
                a = k;
            }
        }
        mid = begin + count / 2;
        std::nth_element(refs.begin() + begin, refs.begin() + mid, refs.begin() + end, [a](const Ref& x, const Ref& y) { return x.c[a] < y.c[a]; });
        return true;
    };

    if (n <= kMaxLeafPrims) {
        Box box, cbox;
        bounds_of(0, n, box, cbox);
        make_leaf(Task{0, n, 1, 0, 0}, box);
        return out;
    }
    {
        Box box, cbox;
        bounds_of(0, n, box, cbox);
        uint32_t mid = 0;
        split(0, n, 0, box, cbox, mid); // n > kMaxLeafPrims: always splits
        stack.push_back(Task{mid, n, 1, 0, 1});
        stack.push_back(Task{0, mid, 1, 0, 0});
    }
    while (!stack.empty()) {
        Task t = stack.back();
        stack.pop_back();
        Box box, cbox;
        bounds_of(t.begin, t.end, box, cbox);
        uint32_t mid = 0;
        if (!split(t.begin, t.end, t.depth, box, cbox, mid)) {
            make_leaf(t, box);
            continue;
        }
        int32_t id = (int32_t)out.nodes.size();
        out.nodes.push_back(empty_node());
        set_child(t.parent, t.slot, id, box);
        stack.push_back(Task{mid, t.end, t.depth + 1, id, 1});
        stack.push_back(Task{t.begin, mid, t.depth + 1, id, 0});
    }
    return out;
}

WideBvh collapse_to_wide(const BuiltBvh& bvh) {
    WideBvh out;
    struct Child {
        float lo[3], hi[3];
        int32_t code; // Node64 child code
    };
    auto child_of = [&](const Node64& n, int k) {
        Child c;
        c.lo[0] = n.lo_x[k], c.lo[1] = n.lo_y[k], c.lo[2] = n.lo_z[k];
        c.hi[0] = n.hi_x[k], c.hi[1] = n.hi_y[k], c.hi[2] = n.hi_z[k];
        c.code = n.child[k];
        return c;
    };
    auto area = [](const Child& c) {
        float dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
        return (dx > 0 && dy >= 0 && dz >= 0) || (dy > 0 && dx >= 0 && dz >= 0) || (dz > 0 && dx >= 0 && dy >= 0) ? dx * dy + dy * dz + dz * dx : 0.0f;
    };
    struct Task {
        int32_t binary_node; // Node64 index this wide node stands for
        int32_t wide_index;
        uint32_t depth, stack_before;
    };
    out.nodes.emplace_back();
    std::vector<Task> todo{{0, 0, 1, 0}};
    while (!todo.empty()) {
        Task t = todo.back();
        todo.pop_back();
        std::vector<Child> kids{child_of(bvh.nodes[t.binary_node], 0), child_of(bvh.nodes[t.binary_node], 1)};
        for (;;) {
            if (kids.size() >= 4) break;
            int best = -1;
            float best_area = -1.0f;
            for (size_t i = 0; i < kids.size(); ++i)
                if (kids[i].code >= 0 && area(kids[i]) > best_area) best_area = area(kids[i]), best = (int)i;
            if (best < 0) break;
            const Node64& inner = bvh.nodes[kids[best].code];
            kids[best] = child_of(inner, 0);
            kids.push_back(child_of(inner, 1));
        }
        // drop empty leaves (a binary node with fewer than two real children)
        std::vector<Child> real;
        for (const Child& c : kids)
            if (c.code >= 0 || ((uint32_t)(-1 - c.code) & 7u) != 0) real.push_back(c);
        Node128 node{};
        for (int k = 0; k < 4; ++k) {
            // an unused slot's box is NaN: every comparison of the slab test fails on it, so the traversal needs no "is there a
            // child" test of its own (an inverted box would not do: the slab test orders each pair of planes itself)
            node.lo_x[k] = node.lo_y[k] = node.lo_z[k] = std::numeric_limits<float>::quiet_NaN();
            node.hi_x[k] = node.hi_y[k] = node.hi_z[k] = std::numeric_limits<float>::quiet_NaN();
            node.child[k] = kEmptyChild;
        }
        const uint32_t pushes = real.empty() ? 0u : (uint32_t)real.size() - 1u;
        out.max_depth = std::max(out.max_depth, t.depth);
        out.stack_need = std::max(out.stack_need, t.stack_before + pushes);
        for (size_t k = 0; k < real.size(); ++k) {
            node.lo_x[k] = real[k].lo[0], node.lo_y[k] = real[k].lo[1], node.lo_z[k] = real[k].lo[2];
            node.hi_x[k] = real[k].hi[0], node.hi_y[k] = real[k].hi[1], node.hi_z[k] = real[k].hi[2];
            if (real[k].code >= 0) {
                const int32_t index = (int32_t)out.nodes.size();
                out.nodes.emplace_back();
                node.child[k] = index;
                todo.push_back(Task{real[k].code, index, t.depth + 1, t.stack_before + pushes});
            } else {
                node.child[k] = real[k].code;
            }
        }
        out.nodes[t.wide_index] = node;
    }
    return out;
}

} // namespace pyr
