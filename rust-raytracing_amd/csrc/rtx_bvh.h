// rtx_bvh.h -- flat BVH over the scene's spheres: host builder + node layout.
//
// The reference walks every object per segment (scene.rs:243-251).  For spheres the same winner can be found
// by visiting only the boxes a ray touches: a sphere the reference reports (near root, is_normal, > 0) is a real
// intersection with t > 0 (sphere.rs:19-30; SURVEY H2 "spheres are BVH-safe"), so it lies inside the sphere's
// (inflated) bounding box, which lies inside every ancestor's box.  The leaves run the SAME exact f64 test as
// the brute-force kernels and the winner is chosen with the same (t, scene index) order, so the image has the
// same bits.  Triangles are NOT put in the BVH: the reference's Triangle::distance reports hits for rays that
// miss the triangle's box (phantom hits), so any spatial culling would change the image.
//
// The reference's own GPU path already attaches a box to each shape -- sphere: position -/+ radius
// (object/sphere.rs:82-86), planes unbounded (plane.rs:83-85) -- the builder uses that box formula.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <numeric>
#include <vector>

namespace rtx {

// 32-byte node of the flat array (depth-first order: the left child of node i is node i + 1).
//   interior: count == 0, link = index of the right child
//   leaf:     count  > 0, link = first entry in bvh_prims[], count entries
// Boxes are f32, rounded OUTWARD from the f64 sphere bounds and inflated, so that the f64 slab test of the
// traversal (which converts them back to f64) can never exclude a sphere the exact test would accept.
struct BvhNode {
    float lo[3];
    uint32_t link;
    float hi[3];
    uint32_t count;
};
static_assert(sizeof(BvhNode) == 32, "BvhNode must be 32 bytes");

constexpr int kBvhLeafSize = 4;
constexpr int kBvhMaxDepth = 30;          // traversal stack entries per ray

struct BvhBuild {
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> prims;          // local sphere indices, leaf-contiguous
    int depth = 0;
};

inline float round_down_f32(double x)
{
    float f = (float)x;
    if ((double)f > x) f = std::nextafterf(f, -INFINITY);
    return f;
}

inline float round_up_f32(double x)
{
    float f = (float)x;
    if ((double)f < x) f = std::nextafterf(f, INFINITY);
    return f;
}

// spheres: n x {cx, cy, cz, r} (f64).  Returns an empty build when any sphere is not finite (the caller then
// does not offer the BVH kernel for this scene).
inline BvhBuild build_sphere_bvh(const double *spheres4, uint32_t n)
{
    BvhBuild out;
    if (n == 0) return out;
    struct Box { double lo[3], hi[3]; };
    std::vector<Box> box(n);
    std::vector<double> cen(3 * (size_t)n);
    for (uint32_t i = 0; i < n; ++i) {
        const double *s = spheres4 + 4 * (size_t)i;
        const double r = std::fabs(s[3]);
        for (int a = 0; a < 3; ++a) {
            if (!std::isfinite(s[a]) || !std::isfinite(r)) return BvhBuild();
            // object/sphere.rs:82-86: position -/+ radius, inflated by 2^-20 relative + a denormal-safe absolute term
            const double pad = (std::fabs(s[a]) + r) * (1.0 / 1048576.0) + 1e-300;
            box[i].lo[a] = s[a] - r - pad;
            box[i].hi[a] = s[a] + r + pad;
            cen[3 * (size_t)i + a] = s[a];
        }
    }
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    out.nodes.reserve(2 * (size_t)n / kBvhLeafSize + 2);

    struct Task { uint32_t begin, end; int parent; int depth; };     // parent < 0: root; parent's link is set when the right child is emitted
    std::vector<Task> todo;
    todo.push_back({0u, n, -1, 1});
    while (!todo.empty()) {
        const Task t = todo.back();
        todo.pop_back();
        const uint32_t me = (uint32_t)out.nodes.size();
        if (t.parent >= 0) out.nodes[(size_t)t.parent].link = me;   // we are a RIGHT child (left children are emitted right after their parent)
        out.depth = std::max(out.depth, t.depth);
        Box b;
        double clo[3], chi[3];
        for (int a = 0; a < 3; ++a) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; clo[a] = INFINITY; chi[a] = -INFINITY; }
        for (uint32_t k = t.begin; k < t.end; ++k) {
            const uint32_t i = order[k];
            for (int a = 0; a < 3; ++a) {
                b.lo[a] = std::min(b.lo[a], box[i].lo[a]); b.hi[a] = std::max(b.hi[a], box[i].hi[a]);
                clo[a] = std::min(clo[a], cen[3 * (size_t)i + a]); chi[a] = std::max(chi[a], cen[3 * (size_t)i + a]);
            }
        }
        BvhNode node;
        for (int a = 0; a < 3; ++a) { node.lo[a] = round_down_f32(b.lo[a]); node.hi[a] = round_up_f32(b.hi[a]); }
        const uint32_t cnt = t.end - t.begin;
        if (cnt <= (uint32_t)kBvhLeafSize) {
            node.link = (uint32_t)out.prims.size();
            node.count = cnt;
            for (uint32_t k = t.begin; k < t.end; ++k) out.prims.push_back(order[k]);
            out.nodes.push_back(node);
            continue;
        }
        // median split of the centroids along their widest axis: balanced, depth <= ceil(log2(n / leaf)) + 1
        int axis = 0;
        if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
        if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
        const uint32_t mid = t.begin + cnt / 2;
        std::nth_element(order.begin() + t.begin, order.begin() + mid, order.begin() + t.end,
                         [&](uint32_t x, uint32_t y) {
                             const double cx = cen[3 * (size_t)x + axis], cy = cen[3 * (size_t)y + axis];
                             return cx < cy || (cx == cy && x < y);
                         });
        node.link = 0;            // patched by the right child
        node.count = 0;
        out.nodes.push_back(node);
        // depth-first: the left child must be the next node emitted -> push right first
        todo.push_back({mid, t.end, (int)me, t.depth + 1});
        todo.push_back({t.begin, mid, -1, t.depth + 1});
    }
    return out;
}

}  // namespace rtx
