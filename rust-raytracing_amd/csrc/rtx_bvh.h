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
#include <cstring>
#include <numeric>
#include <vector>

namespace rtx {

// 32-byte node of the flat array (depth-first order: the left child of node i is node i + 1).
//   interior: count == 0, link = index of the right child
//   leaf:     count  > 0, link = first entry in bvh_prims[], count entries
// Boxes are f32, rounded OUTWARD from the f64 sphere bounds and inflated (relative 2^-20 of the sphere's reach plus
// BvhBuild::abs_pad), so that the f32 slab test of the traversal can never exclude a sphere the exact f64 test
// would accept.
struct BvhNode {
    float lo[3];
    uint32_t link;
    float hi[3];
    uint32_t count;
};
static_assert(sizeof(BvhNode) == 32, "BvhNode must be 32 bytes");

constexpr int kBvhLeafSize = 1;            // spheres per leaf; measured on C2 at 16 spp: leaf 1/2/4/8/16 = 1123/999/919/840/668 Mrays/s
                                           // (storing the sphere's filter record in place of the leaf box was slower: 1057)
constexpr int kBvhMaxDepth = 30;          // traversal stack entries per ray

struct BvhBuild {
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> prims;          // local sphere indices, leaf-contiguous
    int depth = 0;
    // The traversal's slab test runs in f32: t = fl(fl(b - fl(o)) * fl(1/d)).  Rounding the origin shifts both faces
    // of an axis by at most 2^-24 |o|, every other rounding is a relative error <= 2^-22 on t (handled in the test).
    // Every box is therefore also inflated by `abs_pad` >= 2^-22 * origin_limit on each side; rays whose origin lies
    // outside |o|_inf <= origin_limit do not use the tree (they sweep the spheres exhaustively).
    double origin_limit = 0.0;
    double abs_pad = 0.0;
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
    double scale = 0.0;
    for (uint32_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) scale = std::max(scale, std::max(std::fabs(box[i].lo[a]), std::fabs(box[i].hi[a])));
    out.origin_limit = 4.0 * scale + 1.0;
    out.abs_pad = out.origin_limit * (1.0 / 4194304.0);            // 2^-22 * limit: 4x the origin-rounding shift
    if (!(out.origin_limit < 1.0e30)) return BvhBuild();
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
        for (int a = 0; a < 3; ++a) {
            node.lo[a] = round_down_f32(b.lo[a] - out.abs_pad);
            node.hi[a] = round_up_f32(b.hi[a] + out.abs_pad);
        }
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

// ---- 4-wide nodes -------------------------------------------------------------------------------------------
// The traversal is bound by the latency of dependent node fetches, so the binary tree is collapsed into nodes
// that hold the boxes of up to four children inline: one 128-byte fetch per step, half the depth.
//   child c: A = {lo.xyz, link}, B = {hi.xyz, count}
//     count == 0          interior child, link = index of its Bvh4Node
//     1 <= count <= leaf  leaf child, link = first entry in prims[], count entries
//     count == 0xFFFFFFFF empty slot (box inverted: never hit)
struct Bvh4Node {
    float4 a[4];
    float4 b[4];
};
static_assert(sizeof(Bvh4Node) == 128, "Bvh4Node must be 128 bytes");

constexpr int kBvh4StackEntries = 24;     // per-ray traversal stack (LDS); deeper trees may overflow it (handled)

struct Bvh4Build {
    std::vector<Bvh4Node> nodes;
    int depth = 0;
};

inline float u32_as_f32(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

inline Bvh4Build collapse_to_bvh4(const BvhBuild &b2)
{
    Bvh4Build out;
    if (b2.nodes.empty()) return out;
    auto area = [&](uint32_t i) {
        const BvhNode &n = b2.nodes[i];
        const double dx = (double)n.hi[0] - n.lo[0], dy = (double)n.hi[1] - n.lo[1], dz = (double)n.hi[2] - n.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    struct Task { uint32_t bin; uint32_t wide; int depth; };
    std::vector<Task> todo;
    out.nodes.emplace_back();
    todo.push_back({0u, 0u, 1});
    while (!todo.empty()) {
        const Task t = todo.back();
        todo.pop_back();
        out.depth = std::max(out.depth, t.depth);
        uint32_t kids[4];
        int nk = 0;
        if (b2.nodes[t.bin].count != 0) {
            kids[nk++] = t.bin;                                   // a leaf root: one leaf child
        } else {
            kids[nk++] = t.bin + 1;
            kids[nk++] = b2.nodes[t.bin].link;
            while (nk < 4) {                                      // open the interior child with the largest box
                int best = -1;
                double best_area = -1.0;
                for (int k = 0; k < nk; ++k)
                    if (b2.nodes[kids[k]].count == 0 && area(kids[k]) > best_area) { best = k; best_area = area(kids[k]); }
                if (best < 0) break;
                const uint32_t open = kids[best];
                kids[best] = open + 1;
                kids[nk++] = b2.nodes[open].link;
            }
        }
        Bvh4Node w;
        for (int c = 0; c < 4; ++c) {
            if (c < nk) {
                const BvhNode &n = b2.nodes[kids[c]];
                uint32_t link = n.link, count = n.count;
                if (count == 0) {                                 // interior child: gets its own wide node
                    link = (uint32_t)out.nodes.size();
                    out.nodes.emplace_back();
                    todo.push_back({kids[c], link, t.depth + 1});
                }
                w.a[c] = make_float4(n.lo[0], n.lo[1], n.lo[2], u32_as_f32(link));
                w.b[c] = make_float4(n.hi[0], n.hi[1], n.hi[2], u32_as_f32(count));
            } else {
                w.a[c] = make_float4(INFINITY, INFINITY, INFINITY, u32_as_f32(0u));
                w.b[c] = make_float4(-INFINITY, -INFINITY, -INFINITY, u32_as_f32(0xFFFFFFFFu));
            }
        }
        out.nodes[t.wide] = w;
    }
    return out;
}

}  // namespace rtx
