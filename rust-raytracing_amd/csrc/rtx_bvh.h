// rtx_bvh.h -- flat BVH over the scene's spheres and triangles: host builder + node layout.
//
// The reference walks every object per segment (scene.rs:243-251).  The same winner can be found by visiting
// only the boxes a ray touches, as long as a box can never hide a hit the reference would report:
//
//  * Spheres.  A sphere the reference reports (near root, is_normal, > 0) is a real intersection with t > 0
//    (sphere.rs:19-30; SURVEY H2 "spheres are BVH-safe"), so the hit point lies inside the sphere's (inflated) 3-D
//    bounding box, which lies inside every ancestor's box.
//  * Triangles.  Triangle::distance (triangle.rs:108-127) takes |t| of the plane distance and Triangle::contains
//    (triangle.rs:37-101) solves only two rows of the system, so it reports "phantom" hits for rays that miss the
//    triangle's 3-D box -- a 3-D BVH would change the image.  But every hit it reports, phantom or
//    not, is decided at the point q = p + dir * |t| with |t| >= 0, and (for the x, y pivot rows) only by q's
//    (x, y) projection being inside the projected triangle.  Hence: hit at distance |t|  =>  the ray, walked
//    FORWARD from its origin for a length |t|, is above the triangle's (x, y) footprint.  Triangles therefore go
//    into the tree with their (x, y) footprint rectangle and an unbounded z interval: the slab test prunes in
//    two dimensions, and entry-distance pruning against the best hit stays valid.  Triangles whose elimination
//    pivots are not rows (x, y), or whose projection is ill-conditioned, stay outside the tree and are tested for
//    every segment (rtx_api.hip).
//
// The leaves run the SAME exact f64 tests as the brute-force kernels and the winner is chosen with the same
// (t, scene index) order, so the image has the same bits.
//
// The reference's own GPU path already attaches a box to each shape -- sphere: position -/+ radius
// (object/sphere.rs:82-86), triangle: min/max of the vertices (triangle.rs:190-194), planes unbounded
// (plane.rs:83-85) -- the builder uses those box formulas (the triangle's restricted to x, y).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

namespace rtx {

// 32-byte node of the flat binary tree (depth-first order: the left child of node i is node i + 1).
//   interior: count == 0, link = index of the right child
//   leaf:     count  > 0, link = first leaf entry, (count & 0xFFFF) entries; kBvhTriLeaf set = triangle entries
// Boxes are f32, rounded OUTWARD from the f64 bounds and inflated (relative 2^-20 of the shape's reach plus
// BvhBuild::abs_pad), so that the f32 slab test of the traversal can never exclude a shape the exact f64 test
// would accept.
struct BvhNode {
    float lo[3];
    uint32_t link;
    float hi[3];
    uint32_t count;
};
static_assert(sizeof(BvhNode) == 32, "BvhNode must be 32 bytes");

constexpr uint32_t kBvhTriLeaf = 0x10000u;  // flag in a leaf's count: the entries are triangle filter records
constexpr int kBvhLeafSize = 1;            // spheres per leaf; measured on C2 at 16 spp: leaf 1/2/4/8/16 = 1123/999/919/840/668 Mrays/s
                                           // (round 1); with round 2's two-stage kernel at 64 spp: leaf 1/2/3 = 1616/1544/1484
                                           // (storing the sphere's filter record in place of the leaf box was slower: 1057)
constexpr int kBvhTriLeafMax = 8;          // upper bound of triangles per leaf (the build's leaf size is a parameter)

struct BvhBox { double lo[3], hi[3]; };    // hi[2] = -lo[2] = inf for a triangle footprint

struct BvhBuild {
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> prims;          // local sphere indices, leaf-contiguous
    std::vector<uint32_t> tri_order;      // triangle (footprint) indices, leaf-contiguous
    std::vector<uint8_t> flat;            // per node: 1 = a node of the (x, y) footprint sub-tree (gets the flat 2-D layout)
    bool has_spheres = false, has_tris = false;
    // The traversal's slab test runs in f32: t = fl(b * inv + noi), inv = fl(1/d), noi = fl(-o * inv) (rtx_traverse.h).
    // Rounding noi shifts both faces of an axis by at most 2^-24 |o|, the other roundings are a relative error
    // <= 2^-22 on t (handled in the test).
    // Every box is therefore also inflated by `abs_pad` >= 2^-22 * origin_limit on each side; rays whose origin lies
    // outside |o|_inf <= origin_limit do not use the tree (they sweep the shapes exhaustively).
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

// object/sphere.rs:82-86: position -/+ radius, inflated by 2^-20 relative + a denormal-safe absolute term.
// Returns false when the sphere is not finite.
inline bool sphere_box(const double s[4], BvhBox &b)
{
    const double r = std::fabs(s[3]);
    for (int a = 0; a < 3; ++a) {
        if (!std::isfinite(s[a]) || !std::isfinite(r)) return false;
        const double pad = (std::fabs(s[a]) + r) * (1.0 / 1048576.0) + 1e-300;
        b.lo[a] = s[a] - r - pad;
        b.hi[a] = s[a] + r + pad;
    }
    return true;
}

// triangle.rs:190-194 restricted to the coordinate plane Triangle::contains solves in -- `free_axis` is the axis the
// elimination does not read (2: the usual (x, y) rows; 1 / 0: the (x, z) / (y, z) rows a zero pivot swaps in,
// triangle.rs:60-71,81-87): min/max of the vertices in the plane, same inflation; unbounded along free_axis.
inline bool triangle_footprint(const double v[9], BvhBox &b, int free_axis = 2)
{
    for (int a = 0; a < 3; ++a) {
        if (a == free_axis) { b.lo[a] = -INFINITY; b.hi[a] = INFINITY; continue; }
        const double x0 = v[a], x1 = v[3 + a], x2 = v[6 + a];
        if (!std::isfinite(x0) || !std::isfinite(x1) || !std::isfinite(x2)) return false;
        const double lo = std::fmin(x0, std::fmin(x1, x2)), hi = std::fmax(x0, std::fmax(x1, x2));
        const double pad = std::fmax(std::fabs(lo), std::fabs(hi)) * (1.0 / 1048576.0) + 1e-300;
        b.lo[a] = lo - pad;
        b.hi[a] = hi + pad;
    }
    return true;
}

// Appends the binary tree over `box` to out.nodes (its root is the first node appended) and the leaf order to
// `order`.  Splits: binned surface-area heuristic over the box centres (16 bins per axis, the first `dims` axes;
// the "area" of a footprint node is its perimeter, which is what a 2-D ray can hit), i.e. the split that minimises
// area(L) * n(L) + area(R) * n(R).  When the SAH has nothing to offer (all centres in one bin) or the depth budget
// is used up, the node is split at the median of its widest axis, which bounds the depth by
// ceil(log2(n / leaf)) + kBvhSahExtraDepth + 1.
#ifndef RTX_BVH_SAH_BINS
#define RTX_BVH_SAH_BINS 16
#endif
constexpr int kBvhSahBins = RTX_BVH_SAH_BINS;
constexpr int kBvhSahExtraDepth = 6;

// `free_axis` (dims == 2 only): the axis the boxes are unbounded along; the splits use the other two.
inline void bvh_append_tree(const std::vector<BvhBox> &box, int dims, uint32_t leaf_size, uint32_t leaf_flag,
                            BvhBuild &out, std::vector<uint32_t> &order_out, bool use_sah = true, int free_axis = 2,
                            uint32_t order_base = 0)
{
    const uint32_t n = (uint32_t)box.size();
    const int ax[3] = { dims == 2 && free_axis == 0 ? 1 : 0, dims == 2 && free_axis != 2 ? 2 : 1, 2 };   // split axes
    std::vector<double> cen((size_t)dims * n);
    for (uint32_t i = 0; i < n; ++i)
        for (int a = 0; a < dims; ++a) cen[(size_t)dims * i + a] = 0.5 * (box[i].lo[ax[a]] + box[i].hi[ax[a]]);
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    int balanced_depth = 1;
    for (uint64_t c = leaf_size; c < n; c *= 2) ++balanced_depth;
    const int max_depth = balanced_depth + kBvhSahExtraDepth;
    auto measure = [dims, &ax](const BvhBox &b) {
        const double dx = b.hi[ax[0]] - b.lo[ax[0]], dy = b.hi[ax[1]] - b.lo[ax[1]];
        if (dims == 2) return dx + dy;
        const double dz = b.hi[2] - b.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    auto grow = [](BvhBox &b, const BvhBox &o) {
        for (int a = 0; a < 3; ++a) { b.lo[a] = std::min(b.lo[a], o.lo[a]); b.hi[a] = std::max(b.hi[a], o.hi[a]); }
    };
    auto empty_box = [] {
        BvhBox b;
        for (int a = 0; a < 3; ++a) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; }
        return b;
    };
    struct Task { uint32_t begin, end; int parent; int depth; };     // parent < 0: nothing to patch; else parent's link is set when this (right) child is emitted
    std::vector<Task> todo;
    todo.push_back({0u, n, -1, 1});
    while (!todo.empty()) {
        const Task t = todo.back();
        todo.pop_back();
        const uint32_t me = (uint32_t)out.nodes.size();
        if (t.parent >= 0) out.nodes[(size_t)t.parent].link = me;   // we are a RIGHT child (left children are emitted right after their parent)
        BvhBox b = empty_box();
        double clo[3], chi[3];
        for (int a = 0; a < 3; ++a) { clo[a] = INFINITY; chi[a] = -INFINITY; }
        for (uint32_t k = t.begin; k < t.end; ++k) {
            const uint32_t i = order[k];
            grow(b, box[i]);
            for (int a = 0; a < dims; ++a) {
                clo[a] = std::min(clo[a], cen[(size_t)dims * i + a]); chi[a] = std::max(chi[a], cen[(size_t)dims * i + a]);
            }
        }
        BvhNode node;
        for (int a = 0; a < 3; ++a) {
            node.lo[a] = round_down_f32(b.lo[a] - out.abs_pad);
            node.hi[a] = round_up_f32(b.hi[a] + out.abs_pad);
        }
        const uint32_t cnt = t.end - t.begin;
        if (cnt <= leaf_size) {
            node.link = (uint32_t)order_out.size();
            node.count = cnt | leaf_flag;
            for (uint32_t k = t.begin; k < t.end; ++k) order_out.push_back(order_base + order[k]);
            out.nodes.push_back(node);
            continue;
        }
        uint32_t mid = 0;
        // the depth budget left must still allow a balanced finish: 2^(max_depth - depth) * leaf >= cnt
        bool sah_allowed = use_sah && cnt > 2;
        if (sah_allowed) {
            int need = 1;
            for (uint64_t c = leaf_size; c < cnt; c *= 2) ++need;
            sah_allowed = t.depth + need < max_depth;       // one level of slack for the uneven split
        }
        if (sah_allowed) {
            double best_cost = INFINITY;
            int best_axis = -1, best_bin = -1;
            for (int a = 0; a < dims; ++a) {
                const double ext = chi[a] - clo[a];
                if (!(ext > 0.0)) continue;
                BvhBox bin_box[kBvhSahBins];
                uint32_t bin_cnt[kBvhSahBins];
                for (int k = 0; k < kBvhSahBins; ++k) { bin_box[k] = empty_box(); bin_cnt[k] = 0; }
                const double scale = (double)kBvhSahBins / ext;
                for (uint32_t k = t.begin; k < t.end; ++k) {
                    const uint32_t i = order[k];
                    int bi = (int)((cen[(size_t)dims * i + a] - clo[a]) * scale);
                    bi = bi < 0 ? 0 : (bi >= kBvhSahBins ? kBvhSahBins - 1 : bi);
                    grow(bin_box[bi], box[i]);
                    bin_cnt[bi] += 1;
                }
                double right_m[kBvhSahBins];
                uint32_t right_n[kBvhSahBins];
                BvhBox acc = empty_box();
                uint32_t accn = 0;
                for (int k = kBvhSahBins - 1; k >= 1; --k) {
                    if (bin_cnt[k]) grow(acc, bin_box[k]);
                    accn += bin_cnt[k];
                    right_m[k] = accn ? measure(acc) : 0.0;
                    right_n[k] = accn;
                }
                acc = empty_box();
                accn = 0;
                for (int k = 0; k + 1 < kBvhSahBins; ++k) {           // split after bin k
                    if (bin_cnt[k]) grow(acc, bin_box[k]);
                    accn += bin_cnt[k];
                    if (accn == 0 || right_n[k + 1] == 0) continue;
                    const double cost = measure(acc) * accn + right_m[k + 1] * right_n[k + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = k; }
                }
            }
            if (best_axis >= 0) {
                const double ext = chi[best_axis] - clo[best_axis];
                const double scale = (double)kBvhSahBins / ext;
                auto in_left = [&](uint32_t i) {
                    int bi = (int)((cen[(size_t)dims * i + best_axis] - clo[best_axis]) * scale);
                    bi = bi < 0 ? 0 : (bi >= kBvhSahBins ? kBvhSahBins - 1 : bi);
                    return bi <= best_bin;
                };
                mid = (uint32_t)(std::stable_partition(order.begin() + t.begin, order.begin() + t.end, in_left) - order.begin());
            }
        }
        if (mid <= t.begin || mid >= t.end) {
            // median split of the centres along their widest axis
            int axis = 0;
            for (int a = 1; a < dims; ++a)
                if (chi[a] - clo[a] > chi[axis] - clo[axis]) axis = a;
            mid = t.begin + cnt / 2;
            std::nth_element(order.begin() + t.begin, order.begin() + mid, order.begin() + t.end,
                             [&](uint32_t x, uint32_t y) {
                                 const double cx = cen[(size_t)dims * x + axis], cy = cen[(size_t)dims * y + axis];
                                 return cx < cy || (cx == cy && x < y);
                             });
        }
        node.link = 0;            // patched by the right child
        node.count = 0;
        out.nodes.push_back(node);
        // depth-first: the left child must be the next node emitted -> push right first
        todo.push_back({mid, t.end, (int)me, t.depth + 1});
        todo.push_back({t.begin, mid, -1, t.depth + 1});
    }
}

// One tree over the sphere boxes (3-D) and the triangle footprints.  tri_boxes[f] holds the footprints of the triangles
// whose elimination does not read axis f (f = 2: the (x, y) plane, the usual case; 1: (x, z); 0: (y, z)), unbounded along f.
// Every non-empty set gets its own sub-tree (a 2-D build in its plane); the sub-trees hang off a chain of joint nodes at
// the front, so one traversal orders and prunes across all of them.  The caller passes only sets it wants in the tree
// (more than 4 members).  tri_order indexes the concatenation [tri_boxes[2]] [tri_boxes[1]] [tri_boxes[0]].  Returns an
// empty build when the coordinates are too large for the f32 slab test.
inline BvhBuild build_bvh(const std::vector<BvhBox> &sphere_boxes, const std::vector<BvhBox> (&tri_boxes)[3], uint32_t tri_leaf_size,
                          bool use_sah = true)
{
    BvhBuild out;
    const bool want_s = !sphere_boxes.empty();
    const size_t n_tri = tri_boxes[0].size() + tri_boxes[1].size() + tri_boxes[2].size();
    if (!want_s && n_tri == 0) return out;
    double scale = 0.0;
    auto grow_scale = [&](const std::vector<BvhBox> &v) {
        for (const BvhBox &b : v)
            for (int a = 0; a < 3; ++a) {
                if (std::isfinite(b.lo[a])) scale = std::max(scale, std::fabs(b.lo[a]));
                if (std::isfinite(b.hi[a])) scale = std::max(scale, std::fabs(b.hi[a]));
            }
    };
    grow_scale(sphere_boxes);
    for (const auto &v : tri_boxes) grow_scale(v);
    out.origin_limit = 4.0 * scale + 1.0;
    out.abs_pad = out.origin_limit * (1.0 / 4194304.0);            // 2^-22 * limit: 4x the origin-rounding shift
    if (!(out.origin_limit < 1.0e28)) return BvhBuild();
    if (sphere_boxes.size() + n_tri >= 0x10000000ull) return BvhBuild();          // node / record indices must stay below 2^28 (kBvhFlatNode, the walk's leaf notes)
    if (tri_leaf_size < 1) tri_leaf_size = 1;
    if (tri_leaf_size > (uint32_t)kBvhTriLeafMax) tri_leaf_size = (uint32_t)kBvhTriLeafMax;
    out.nodes.reserve(2 * (sphere_boxes.size() + n_tri) + 8);
    struct Sub { const std::vector<BvhBox> *boxes; int free_axis; uint32_t base; };      // free_axis < 0: the spheres
    std::vector<Sub> subs;
    if (want_s) subs.push_back({&sphere_boxes, -1, 0u});
    uint32_t base = 0;
    for (int f = 2; f >= 0; --f) {
        if (!tri_boxes[f].empty()) subs.push_back({&tri_boxes[f], f, base});
        base += (uint32_t)tri_boxes[f].size();
    }
    std::vector<uint32_t> joints;
    for (size_t k = 0; k < subs.size(); ++k) {
        if (k + 1 < subs.size()) {                                  // a joint: left child = this sub-tree, right child = the rest
            joints.push_back((uint32_t)out.nodes.size());
            out.nodes.emplace_back();
        } else if (!joints.empty()) {
            out.nodes[joints.back()].link = (uint32_t)out.nodes.size();        // the last sub-tree is the last joint's right child
        }
        if (k > 0 && k + 1 < subs.size()) out.nodes[joints[k - 1]].link = joints[k];
        const size_t first = out.nodes.size();
        if (subs[k].free_axis < 0) {
            bvh_append_tree(*subs[k].boxes, 3, (uint32_t)kBvhLeafSize, 0u, out, out.prims, use_sah);
            out.has_spheres = true;
        } else {
            bvh_append_tree(*subs[k].boxes, 2, tri_leaf_size, kBvhTriLeaf, out, out.tri_order, use_sah, subs[k].free_axis, subs[k].base);
            out.has_tris = true;
        }
        out.flat.resize(out.nodes.size(), 0);
        if (subs[k].free_axis == 2) std::fill(out.flat.begin() + (ptrdiff_t)first, out.flat.end(), (uint8_t)1);
    }
    for (size_t k = joints.size(); k-- > 0;) {                     // joint boxes, innermost first
        BvhNode &j = out.nodes[joints[k]];
        const BvhNode &l = out.nodes[joints[k] + 1], &r = out.nodes[j.link];
        for (int a = 0; a < 3; ++a) { j.lo[a] = std::min(l.lo[a], r.lo[a]); j.hi[a] = std::max(l.hi[a], r.hi[a]); }
        j.count = 0;
    }
    return out;
}

// ---- 4-wide nodes -------------------------------------------------------------------------------------------
// The traversal is bound by the latency of dependent node fetches, so the binary tree is collapsed into nodes
// that hold the boxes of up to four children inline: one 128-byte fetch per step, half the depth.
//   child c: A = {lo.xyz, link}, B = {hi.xyz, count}
//     count == 0               interior child, link = index of its Bvh4Node
//     1 <= count <= leaf       sphere leaf, link = first entry in prims[] / leaf records, count entries
//     kBvhTriLeaf | count      triangle leaf, link = first triangle filter record, count records
//     count == 0xFFFFFFFF      empty slot (neither leaf nor interior: ignored whatever its box test says)
struct Bvh4Node {
    float4 a[4];
    float4 b[4];
};
static_assert(sizeof(Bvh4Node) == 128, "Bvh4Node must be 128 bytes");

#ifndef RTX_BVH_STACK
#define RTX_BVH_STACK 30
#endif
constexpr int kBvh4StackEntries = RTX_BVH_STACK;     // per-ray traversal stack (LDS); deeper trees may overflow it (handled)

// Footprint nodes (every child box unbounded in z: the triangle sub-tree) use a flat layout in the same 128 bytes,
// which needs 6 of the 8 loads and a 2-D slab test:
//   a[c] = {lo.x, lo.y, hi.x, hi.y} of child c;  b[0] = the four links, b[1] = the four counts;  b[2], b[3] unused.
// A link (or the root) that points to such a node has kBvhFlatNode set.
constexpr uint32_t kBvhFlatNode = 0x80000000u;

struct Bvh4Build {
    std::vector<Bvh4Node> nodes;
    int depth = 0;
    uint32_t root = 0;                    // 0, or kBvhFlatNode when the root itself is a footprint node
};

inline float u32_as_f32(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
constexpr int kBvhTopLevels = 6;           // collapse_to_bvh4 numbers the nodes of the first kBvhTopLevels + 1 levels breadth-first

inline Bvh4Build collapse_to_bvh4(const BvhBuild &b2)
{
    Bvh4Build out;
    if (b2.nodes.empty()) return out;
    auto area = [&](uint32_t i) {                                  // what a ray can hit: the area, or -- for a footprint node,
        const BvhNode &n = b2.nodes[i];                            // unbounded along one axis -- the perimeter in its plane
        double e[3], fin[3];
        int nf = 0;
        for (int a = 0; a < 3; ++a) { e[a] = (double)n.hi[a] - n.lo[a]; if (std::isfinite(e[a])) fin[nf++] = e[a]; }
        if (nf == 3) return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
        if (nf == 2) return fin[0] + fin[1];
        return (double)INFINITY;                                   // a joint over several planes: open it first
    };
    auto flat = [&](uint32_t bin) { return bin < b2.flat.size() && b2.flat[bin] != 0; };     // a node of the (x, y) sub-tree
    // Node numbering: the top of the tree breadth-first (every node of depth <= kBvhTopLevels + 1 before any deeper one, level by
    // level), the rest depth-first (a sub-tree's nodes stay together).  The children of a node always get consecutive indices.
    // A kernel that keeps the nodes every walk passes through close to the lanes (the slot kernel's LDS copy of nodes [0, K)) can
    // then take any prefix of the array: a prefix is the top of the tree.
    struct Task { uint32_t bin; uint32_t wide; int depth; };
    std::vector<Task> todo, top;
    size_t top_head = 0;
    out.nodes.emplace_back();
    out.root = flat(0u) ? kBvhFlatNode : 0u;
    top.push_back({0u, 0u, 1});
    while (top_head < top.size() || !todo.empty()) {
        Task t;
        if (top_head < top.size()) t = top[top_head++];
        else { t = todo.back(); todo.pop_back(); }
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
        const bool wflat = flat(t.bin);
        uint32_t links[4] = { 0u, 0u, 0u, 0u }, counts[4] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu };
        for (int c = 0; c < 4; ++c) {
            if (c < nk) {
                const BvhNode &n = b2.nodes[kids[c]];
                uint32_t link = n.link, count = n.count;
                if (count == 0) {                                 // interior child: gets its own wide node
                    link = (uint32_t)out.nodes.size();
                    out.nodes.emplace_back();
                    (t.depth + 1 <= kBvhTopLevels ? top : todo).push_back({kids[c], link, t.depth + 1});
                    if (flat(kids[c])) link |= kBvhFlatNode;
                }
                links[c] = link; counts[c] = count;
                if (wflat) w.a[c] = make_float4(n.lo[0], n.lo[1], n.hi[0], n.hi[1]);
                else {
                    w.a[c] = make_float4(n.lo[0], n.lo[1], n.lo[2], u32_as_f32(link));
                    w.b[c] = make_float4(n.hi[0], n.hi[1], n.hi[2], u32_as_f32(count));
                }
            } else if (wflat) {
                w.a[c] = make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);
            } else {
                w.a[c] = make_float4(INFINITY, INFINITY, INFINITY, u32_as_f32(0u));
                w.b[c] = make_float4(-INFINITY, -INFINITY, -INFINITY, u32_as_f32(0xFFFFFFFFu));
            }
        }
        if (wflat) {
            w.b[0] = make_float4(u32_as_f32(links[0]), u32_as_f32(links[1]), u32_as_f32(links[2]), u32_as_f32(links[3]));
            w.b[1] = make_float4(u32_as_f32(counts[0]), u32_as_f32(counts[1]), u32_as_f32(counts[2]), u32_as_f32(counts[3]));
            w.b[2] = make_float4(0.f, 0.f, 0.f, 0.f);
            w.b[3] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        out.nodes[t.wide] = w;
    }
    return out;
}

// ---- 64-byte footprint nodes ------------------------------------------------------------------------------------------
// The walk of a mesh is bound by how many address-divergent 16-byte requests the L1 serves: a footprint
// node costs 6.  When the tree holds nothing but (x, y) footprints and no leaf has more than 2 records, every wide node
// also exists in a 4-request form: the four child rectangles as 16-bit offsets from the node's own rectangle,
//     {ox, oy, sx, sy}   {lo.x | hi.x << 16} x 4   {lo.y | hi.y << 16} x 4   {type << 29 | index} x 4
// child plane = ox + q * sx with sx a power of two (so q * sx is exact), lo rounded down, hi rounded up: the decoded
// rectangle contains the f32 rectangle of the 128-byte node, which already carries the padding the f32 slab test needs;
// type 0 = interior (index = wide node), 1..6 = triangle leaf with that many records (index = first record), 7 = empty.
constexpr uint32_t kQNodeShift = 29, kQNodeIndexMask = (1u << 29) - 1u, kQNodeEmpty = 7u, kQNodeLeafMax = 6u;
struct BvhQNode {
    float ox, oy, sx, sy;
    uint32_t qx[4], qy[4], link[4];
};
static_assert(sizeof(BvhQNode) == 64, "BvhQNode must be 64 bytes");

inline bool build_qnodes(const Bvh4Build &b4, std::vector<BvhQNode> &out)
{
    out.clear();
    out.reserve(b4.nodes.size());
    auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    for (const Bvh4Node &w : b4.nodes) {
        const uint32_t lk[4] = { bits(w.b[0].x), bits(w.b[0].y), bits(w.b[0].z), bits(w.b[0].w) };
        const uint32_t ct[4] = { bits(w.b[1].x), bits(w.b[1].y), bits(w.b[1].z), bits(w.b[1].w) };
        double lo[2] = { INFINITY, INFINITY }, hi[2] = { -INFINITY, -INFINITY };
        for (int c = 0; c < 4; ++c) {
            if (ct[c] == 0xFFFFFFFFu) continue;
            const float r[4] = { w.a[c].x, w.a[c].y, w.a[c].z, w.a[c].w };
            for (int k = 0; k < 4; ++k) if (!std::isfinite(r[k])) return false;
            lo[0] = std::min(lo[0], (double)r[0]); lo[1] = std::min(lo[1], (double)r[1]);
            hi[0] = std::max(hi[0], (double)r[2]); hi[1] = std::max(hi[1], (double)r[3]);
        }
        BvhQNode q;
        double o[2], sc[2];
        for (int a = 0; a < 2; ++a) {
            o[a] = std::isfinite(lo[a]) ? lo[a] : 0.0;
            const double ext = std::isfinite(lo[a]) ? hi[a] - lo[a] : 0.0;
            sc[a] = ext > 0.0 ? std::exp2(std::ceil(std::log2(ext / 65535.0))) : 1.0;
            while (ext / sc[a] > 65535.0) sc[a] *= 2.0;                     // (log2 rounding)
            if (!(sc[a] >= 1.1754944e-38 && sc[a] <= 1.0e30)) return false;   // must stay a normal f32
        }
        q.ox = (float)o[0]; q.oy = (float)o[1]; q.sx = (float)sc[0]; q.sy = (float)sc[1];
        for (int c = 0; c < 4; ++c) {
            if (ct[c] == 0xFFFFFFFFu) { q.qx[c] = 0x0000FFFFu; q.qy[c] = 0x0000FFFFu; q.link[c] = kQNodeEmpty << kQNodeShift; continue; }
            const double l0 = std::floor(((double)w.a[c].x - o[0]) / sc[0]), h0 = std::ceil(((double)w.a[c].z - o[0]) / sc[0]);
            const double l1 = std::floor(((double)w.a[c].y - o[1]) / sc[1]), h1 = std::ceil(((double)w.a[c].w - o[1]) / sc[1]);
            if (l0 < 0 || l1 < 0 || h0 > 65535 || h1 > 65535) return false;
            q.qx[c] = (uint32_t)l0 | ((uint32_t)h0 << 16);
            q.qy[c] = (uint32_t)l1 | ((uint32_t)h1 << 16);
            uint32_t type;
            if (ct[c] == 0u) type = 0u;
            else if ((ct[c] & kBvhTriLeaf) && (ct[c] & 0xFFFFu) >= 1u && (ct[c] & 0xFFFFu) <= kQNodeLeafMax) type = ct[c] & 0xFFFFu;
            else return false;                                              // a sphere leaf or a bigger leaf: no 64-byte form
            const uint32_t idx = lk[c] & ~kBvhFlatNode;
            if (idx > kQNodeIndexMask) return false;
            if (type == 0u && !(lk[c] & kBvhFlatNode)) return false;        // an interior child that is not a footprint node
            q.link[c] = (type << kQNodeShift) | idx;
        }
        out.push_back(q);
    }
    return true;
}


// ---- 64-byte nodes of a sphere tree -----------------------------------------------------------------------------------------
// A per-lane walk fetches its node as address-divergent 16-byte requests, and the L1's request rate (about one per cycle per
// CU) is what bounds the bounced rays of a sphere scene before the VALUs do (2.5e10 requests per C2 launch).
// A 128-byte node costs 8 requests; this form costs 4: the child boxes as 8-bit offsets on the node's own grid,
//     {ox, oy, oz, sx}   {sy, sz, lo.x[4], lo.y[4]}   {lo.z[4], hi.x[4], hi.y[4], hi.z[4]}   {link[4]}
// (byte c of a packed word = child c).  plane = o + q * s with s a power of two and o a multiple of s, so the decoded plane
// is exact in f32 (checked at build: |o / s| + 255 < 2^24); lo is rounded down and hi up AFTER another abs_pad was added, so
// the decoded box contains the 128-byte node's box with the margin the extra roundings of the decode need.
// link = type << 29 | index, type 0 interior, 1..6 sphere leaf with that many entries, 7 empty (as BvhQNode).
struct BvhQ3Node {
    float ox, oy, oz, sx;
    float sy, sz;
    uint32_t lox, loy;
    uint32_t loz, hix, hiy, hiz;
    uint32_t link[4];
};
static_assert(sizeof(BvhQ3Node) == 64, "BvhQ3Node must be 64 bytes");

inline bool build_q3nodes(const Bvh4Build &b4, double abs_pad, std::vector<BvhQ3Node> &out)
{
    out.clear();
    out.reserve(b4.nodes.size());
    auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    for (const Bvh4Node &w : b4.nodes) {
        double lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
        uint32_t cnt[4], lnk[4];
        for (int c = 0; c < 4; ++c) {
            cnt[c] = bits(w.b[c].w); lnk[c] = bits(w.a[c].w);
            if (cnt[c] == 0xFFFFFFFFu) continue;
            const float l[3] = { w.a[c].x, w.a[c].y, w.a[c].z }, h[3] = { w.b[c].x, w.b[c].y, w.b[c].z };
            for (int a = 0; a < 3; ++a) {
                if (!std::isfinite(l[a]) || !std::isfinite(h[a])) return false;      // a footprint (unbounded) child: no 64-byte form
                lo[a] = std::min(lo[a], (double)l[a] - abs_pad); hi[a] = std::max(hi[a], (double)h[a] + abs_pad);
            }
        }
        BvhQ3Node q;
        std::memset(&q, 0, sizeof q);
        double o[3], sc[3];
        for (int a = 0; a < 3; ++a) {
            const double ext = std::isfinite(lo[a]) ? hi[a] - lo[a] : 0.0;
            sc[a] = ext > 0.0 ? std::exp2(std::ceil(std::log2(ext / 254.0))) : 1.0;
            if (!(sc[a] >= 1.1754944e-38 && sc[a] <= 1.0e30)) return false;
            o[a] = std::isfinite(lo[a]) ? std::floor(lo[a] / sc[a]) * sc[a] : 0.0;      // a multiple of the step
            while (std::isfinite(lo[a]) && (hi[a] - o[a]) / sc[a] > 255.0) { sc[a] *= 2.0; o[a] = std::floor(lo[a] / sc[a]) * sc[a]; }
            if (!(std::fabs(o[a] / sc[a]) + 256.0 < 16777216.0)) return false;          // o + q * s must be exact in f32
            if ((double)(float)o[a] != o[a]) return false;
        }
        q.ox = (float)o[0]; q.oy = (float)o[1]; q.oz = (float)o[2]; q.sx = (float)sc[0]; q.sy = (float)sc[1]; q.sz = (float)sc[2];
        uint32_t *lows[3] = { &q.lox, &q.loy, &q.loz }, *highs[3] = { &q.hix, &q.hiy, &q.hiz };
        for (int c = 0; c < 4; ++c) {
            if (cnt[c] == 0xFFFFFFFFu) {                          // empty: an inverted box (lo 255, hi 0) is never entered
                for (int a = 0; a < 3; ++a) *lows[a] |= 255u << (8 * c);
                q.link[c] = kQNodeEmpty << kQNodeShift;
                continue;
            }
            const float l[3] = { w.a[c].x, w.a[c].y, w.a[c].z }, h[3] = { w.b[c].x, w.b[c].y, w.b[c].z };
            for (int a = 0; a < 3; ++a) {
                const double ql = std::floor(((double)l[a] - abs_pad - o[a]) / sc[a]), qh = std::ceil(((double)h[a] + abs_pad - o[a]) / sc[a]);
                if (ql < 0 || qh > 255 || ql > qh) return false;
                *lows[a] |= (uint32_t)ql << (8 * c);
                *highs[a] |= (uint32_t)qh << (8 * c);
            }
            uint32_t type;
            if (cnt[c] == 0u) type = 0u;
            else if ((cnt[c] & kBvhTriLeaf) == 0u && cnt[c] >= 1u && cnt[c] <= kQNodeLeafMax) type = cnt[c];
            else return false;                                    // a triangle leaf or a bigger leaf: no 64-byte form
            if (lnk[c] > kQNodeIndexMask || (lnk[c] & kBvhFlatNode)) return false;
            q.link[c] = (type << kQNodeShift) | lnk[c];
        }
        out.push_back(q);
    }
    return true;
}

}  // namespace rtx
