// rtx_api.hip -- host side of librtx_hip.so: the extern "C" entry points of include/rtx_hip.h.
// Scene packing (Scene.objects -> per-type device arrays with scene-order ids), the
// ray-independent precompute, scratch management, kernel launches, timing.
// There is no CPU fallback in this file: every render path ends in a gfx950 kernel launch.
#include "../../include/rtx_hip.h"
#include "rtx_launch.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace rtx;

namespace {

thread_local std::string g_last_error;

int32_t fail(RtxStatus st, const std::string &msg)
{
    g_last_error = msg;
    return (int32_t)st;
}

#define RTX_HIP_CHECK(expr)                                                                              \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return fail(e_ == hipErrorOutOfMemory ? RTX_ERR_OUT_OF_MEMORY : RTX_ERR_HIP,                 \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                              \
    } while (0)

bool device_is_gfx950(int dev)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

int usable_device_count()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int ok = 0;
    for (int d = 0; d < n; ++d) ok += device_is_gfx950(d) ? 1 : 0;
    return ok;
}

// per-render scratch of one handle: sized for a 288 GB part (24 GiB: a C2 frame of 64 spp -- 12.7 GB of sample records and
// queue -- and 59 M rays of the wavefront form run as one launch; fewer, larger launches measured 3-10 % faster)
constexpr size_t kDefaultScratchBytes = (size_t)24576 << 20;

// Triangles per BVH leaf (RTX_TUNE_TRI_LEAF_SHIFT bits of RtxConfig.tuning: 1..6 -- a leaf must fit the walk's 6-entry
// candidate queue, or every visit of it ends in the exhaustive sweep).  A tree of (x, y) footprints alone (`plain`: C3, C5)
// gets 5: a packet tests a leaf's records for 64 rays at once and the regrouping kernel's leaf half reads one leaf per lane
// per iteration (rtx_mesh_step.h), so fewer, fuller leaves pay.  Measured, C3 1080p x 8 / C5 band x 4 in ms, the wavefront
// form | the regrouping kernel alone:  3: 41.7 / 59.5 | 79.8 / 75.2,  4: 39.5 / 56.4 | 76.5 / 72.3,  5: 38.0 / 55.2 | 77.4 / 72.1,
// 6: 38.6 / 56.6 | 78.9 / 74.0  (2, before the step was split: 53.2 / 72.6 | 99.2 / 94.7).  A joint tree keeps 2 (240k axis-aligned
// faces: 222 ms at 2, 242 at 4).
uint32_t tri_leaf_size(bool plain, uint32_t tuning)
{
    const uint32_t t = (tuning >> RTX_TUNE_TRI_LEAF_SHIFT) & 15u;
    const uint32_t v = t ? t : (plain ? 5u : 2u);
    return v > kQNodeLeafMax ? kQNodeLeafMax : v;
}

bool debug_prints()
{
    static const bool on = std::getenv("RTX_HIP_DEBUG") != nullptr;      // diagnostics on stderr only; changes no result
    return on;
}

}  // namespace

#ifndef RTX_GRABS_PER_WAVE
#define RTX_GRABS_PER_WAVE 16          // (render_band: grabs a wave makes from the ray queue)
#endif

// A two-stage sphere launch of at most this many rays runs as two halves in flight (render_band).  0: never by default -- the form
// is bit-identical and SLOWER (LAB_NOTEBOOK R4.11: the drain is nearly empty waves that still hold their slots, not free slots the
// other half could take); RTX_TUNE_HALVES keeps the experiment one bit away.
constexpr uint64_t kHalvesBelowRays = 0;

struct RtxSceneHandle_ {
    int device = 0;
    int n_cus = 0;
    RtxConfig cfg{};
    RtxCamera cam{};
    SceneView sv{};
    std::vector<RtxObject> objects;                       // Scene.objects as uploaded (rtx_scene_append_objects re-packs them)
    std::vector<void *> scene_allocs;
    // scratch, grown on demand
    double *samples = nullptr;  size_t samples_bytes = 0;
    double *acc = nullptr;      size_t acc_bytes = 0;
    double *tables = nullptr;   size_t tables_doubles = 0;
    double *h_tables = nullptr; size_t h_tables_doubles = 0;
    double *state = nullptr;    size_t state_bytes = 0;
    void *wf_state = nullptr;   size_t wf_bytes = 0;       // the wavefront kernels' ray state
    void *pool = nullptr;       size_t pool_bytes = 0;     // (lab) the pool / pair forms of the sphere kernel's stage 2
    void *slots = nullptr;      size_t slots_bytes = 0;    // the slot records of the sphere kernel's stage 2 (trace_sph_slots_kernel)
    void *tile_lists = nullptr; size_t tile_lists_bytes = 0;   // what each tile's primary rays can reach (build_tile_lists_kernel)
    PathStep *transcript = nullptr; uint32_t *transcript_counts = nullptr; uint32_t transcript_steps = 0;   // (lab) rtx_debug_paths, set for one call
    Counters *counters = nullptr;
    unsigned long long *work_counter = nullptr;
    SceneView *d_sv = nullptr;  bool sv_dirty = true;     // device copy of sv (kernels take it by pointer)
    RowsView *d_rv = nullptr;
    hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };      // [3]: the end of the sphere kernel's stage 1 (stats only)
    // The second half's own stream and buffers when a launch runs as two halves in flight (render_band, "halves"): created on first use.
    struct SecondHalf {
        hipStream_t stream = nullptr;
        hipEvent_t fork = nullptr, join = nullptr, stage1_done = nullptr;
        Counters *counters = nullptr, *counters_stage1 = nullptr;
        unsigned long long *work_counter = nullptr;
        RowsView *d_rv = nullptr;
        double *state = nullptr;    size_t state_bytes = 0;
        void *queue = nullptr;      size_t queue_bytes = 0;
    } half2;
    Counters *counters_stage1 = nullptr;                   // the counters as stage 1 left them (stats only)
    // The handle's device buffers (descriptors, tables, counters, scratch) are shared by all of its renders, so they
    // are ordered on ONE stream at a time: when a call brings a different stream the previous one is drained first.
    size_t scratch_limit = 0;                             // rtx_scene_set_scratch_limit (0 = default)
    hipStream_t last_stream = nullptr;                    // compared, never used: the caller may have destroyed it since
    bool have_last_stream = false;
    hipEvent_t ev_done = nullptr;                         // recorded at the end of every render, on the stream it ran on
    bool have_done = false;
    // Round-bound watchdog of the sweep kernel (ctr[1].pad_): copied to this pinned word after every launch and looked
    // at by the next entry point that finds the copy complete (and by rtx_scene_free), so that the asynchronous
    // stats == NULL path reports it too.
    unsigned long long *h_watchdog = nullptr;
    hipEvent_t ev_watchdog = nullptr;
    bool watchdog_pending = false;
    // what the trig tables currently hold
    uint32_t t_w = 0, t_h = 0, t_rb = 0, t_rs = 0, t_blk = 0, t_nr = 0;
    double t_fov = 0.0;
    bool t_valid = false;
};

namespace {

template <class T>
int32_t upload_vec(RtxSceneHandle_ *h, const std::vector<T> &v, const T **out)
{
    *out = nullptr;
    if (v.empty()) return RTX_OK;
    void *d = nullptr;
    RTX_HIP_CHECK(hipMalloc(&d, v.size() * sizeof(T)));
    h->scene_allocs.push_back(d);
    RTX_HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T *>(d);
    return RTX_OK;
}

void apply_config(RtxSceneHandle_ *h, const RtxConfig &cfg)
{
    h->cfg = cfg;
    h->sv.rays_per_pixel = cfg.rays_per_pixel;
    h->sv.max_bounces = cfg.max_bounces;
    h->sv.focal_length = cfg.focal_length;
    h->sv.focal_offset = cfg.focal_offset;
    h->sv.non_focal_offset = cfg.non_focal_offset;
    h->sv.seed = cfg.seed;
    h->sv.tuning = cfg.tuning;
    h->sv_dirty = true;
}

#ifdef RTX_LAB
constexpr bool kLabBuild = true;
#else
constexpr bool kLabBuild = false;
#endif

int32_t check_config(const RtxConfig &cfg)
{
    if (cfg.kernel > RTX_KERNEL_WAVEFRONT) return fail(RTX_ERR_INVALID_ARGUMENT, "RtxConfig.kernel: unknown kernel id");
    if (cfg.tuning & ~(uint32_t)RTX_TUNE_KNOWN_MASK)
        return fail(RTX_ERR_UNSUPPORTED, "RtxConfig.tuning: a bit include/rtx_hip.h does not name");
    if (!kLabBuild && (cfg.tuning & (uint32_t)RTX_TUNE_LAB_MASK))
        return fail(RTX_ERR_UNSUPPORTED, "RtxConfig.tuning: a RTX_TUNE_LAB_MASK bit selects a kernel that exists in librtx_hip_lab.so only "
                                         "(this is the product library)");
    if (cfg.max_bounces == UINT64_MAX)       // max_bounces + 1 overflows in the reference (scene.rs:227)
        return fail(RTX_ERR_INVALID_ARGUMENT, "RtxConfig.max_bounces + 1 overflows");
    if (cfg.rays_per_pixel > 0xFFFFFFFFull) return fail(RTX_ERR_INVALID_ARGUMENT, "RtxConfig.rays_per_pixel exceeds 2^32-1");
    return RTX_OK;
}

int32_t grow(void **p, size_t *have, size_t want)
{
    if (*have >= want && *p) return RTX_OK;
    if (*p) { RTX_HIP_CHECK(hipFree(*p)); *p = nullptr; *have = 0; }
    if (want == 0) return RTX_OK;
    RTX_HIP_CHECK(hipMalloc(p, want));
    *have = want;
    return RTX_OK;
}

void free_handle(RtxSceneHandle_ *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    for (void *p : h->scene_allocs) (void)hipFree(p);
    if (h->samples) (void)hipFree(h->samples);
    if (h->acc) (void)hipFree(h->acc);
    if (h->tables) (void)hipFree(h->tables);
    if (h->h_tables) (void)hipHostFree(h->h_tables);
    if (h->state) (void)hipFree(h->state);
    if (h->wf_state) (void)hipFree(h->wf_state);
    if (h->pool) (void)hipFree(h->pool);
    if (h->slots) (void)hipFree(h->slots);
    if (h->tile_lists) (void)hipFree(h->tile_lists);
    if (h->counters) (void)hipFree(h->counters);
    if (h->counters_stage1) (void)hipFree(h->counters_stage1);
    if (h->work_counter) (void)hipFree(h->work_counter);
    if (h->d_sv) (void)hipFree(h->d_sv);
    if (h->d_rv) (void)hipFree(h->d_rv);
    {
        auto &b = h->half2;
        if (b.stream) { (void)hipStreamSynchronize(b.stream); (void)hipStreamDestroy(b.stream); }
        for (hipEvent_t e : { b.fork, b.join, b.stage1_done }) if (e) (void)hipEventDestroy(e);
        for (void *q : { (void *)b.counters, (void *)b.counters_stage1, (void *)b.work_counter, (void *)b.d_rv, (void *)b.state, b.queue })
            if (q) (void)hipFree(q);
    }
    if (h->h_watchdog) (void)hipHostFree(h->h_watchdog);
    if (h->ev_watchdog) (void)hipEventDestroy(h->ev_watchdog);
    if (h->ev_done) (void)hipEventDestroy(h->ev_done);
    for (auto &e : h->ev) if (e) (void)hipEventDestroy(e);
    delete h;
}

}  // namespace

// Everything rtx_scene_upload prepares on the host: per-type shape arrays with scene-order ids, the ray-independent
// precompute, the f32 filter records and the BVH.  No HIP call in here (rtx_debug_host_scene runs it without a GPU).
namespace {

struct PackedScene {
    SceneView sv{};
    std::vector<SphereX> spheres; std::vector<uint32_t> sphere_id;
    std::vector<PlaneX> planes; std::vector<TriX> tris;
    std::vector<MaterialX> mats;
    std::vector<float4> sph32, tri32, leaf32, leaf_cr, tri_geo;
    std::vector<uint32_t> tri_fidx;
    std::vector<uint8_t> tri_rec_free_axis;          // per tree record (leaf order): the axis its footprint is unbounded along
    BvhBuild bvh;
    Bvh4Build bvh4;
    std::vector<BvhQNode> qnodes;                    // the 64-byte form of bvh4's nodes, when the tree allows it
    std::vector<BvhQ3Node> q3nodes;                  // the 64-byte form of a sphere tree's nodes
};

int32_t pack_scene(const RtxScene *scene, PackedScene &p)
{
    const uint64_t n = scene->n_objects;
    std::vector<SphereX> &spheres = p.spheres; std::vector<uint32_t> &sphere_id = p.sphere_id;
    std::vector<PlaneX> &planes = p.planes; std::vector<TriX> &tris = p.tris;
    std::vector<MaterialX> &mats = p.mats;
    mats.assign(n, MaterialX());
    double lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (uint64_t k = 0; k < n; ++k) {
        const RtxObject &o = scene->objects[k];
        mats[k].base_color = mk(o.base_color[0], o.base_color[1], o.base_color[2]);
        mats[k].emission_color = mk(o.emission_color[0], o.emission_color[1], o.emission_color[2]);
        mats[k].roughness = o.roughness;
        switch (o.kind) {
            case RTX_SPHERE:
                spheres.push_back(make_sphere(o.geom));
                sphere_id.push_back((uint32_t)k);
                for (int c = 0; c < 3; ++c) {
                    if (std::isfinite(o.geom[c])) { lo[c] = std::fmin(lo[c], o.geom[c]); hi[c] = std::fmax(hi[c], o.geom[c]); }
                }
                break;
            case RTX_PLANE: planes.push_back(make_plane(o.geom, (uint32_t)k)); break;
            case RTX_TRIANGLE:
                tris.push_back(make_triangle(o.geom, (uint32_t)k));
                for (int c = 0; c < 9; ++c)
                    if (std::isfinite(o.geom[c])) { lo[c % 3] = std::fmin(lo[c % 3], o.geom[c]); hi[c % 3] = std::fmax(hi[c % 3], o.geom[c]); }
                break;
            default:
                return fail(RTX_ERR_UNSUPPORTED, "object " + std::to_string(k) + ": kind " + std::to_string(o.kind) +
                                                     " has no device primitive (user CustomShape impls cannot run on the GPU)");
        }
    }
    // f32 filter records, relative to the centre of the spheres' bounding box, pair-interleaved for
    // v_pk_fma_f32: record 2k = {x[2k], x[2k+1], y[2k], y[2k+1]}, record 2k+1 = {z.., w..}, w = |c|^2 - r^2;
    // padded to a multiple of 4 spheres with a sentinel that never passes (w = 1e30)
    const size_t ns4 = (spheres.size() + 3) & ~(size_t)3;
    std::vector<float4> &sph32 = p.sph32;
    sph32.assign(ns4, make_float4(0.f, 0.f, 0.f, 0.f));
    std::vector<float> fx(ns4, 0.f), fy(ns4, 0.f), fz(ns4, 0.f), fw(ns4, 1.0e30f);
    double centre[3] = { 0, 0, 0 };
    for (int c = 0; c < 3; ++c) if (lo[c] <= hi[c]) centre[c] = 0.5 * (lo[c] + hi[c]);
    double cmax = 0.0;
    for (size_t k = 0; k < spheres.size(); ++k) {
        const RtxObject &o = scene->objects[sphere_id[k]];
        double cx = o.geom[0] - centre[0], cy = o.geom[1] - centre[1], cz = o.geom[2] - centre[2];
        double cc = cx * cx + cy * cy + cz * cz;
        double reach = std::sqrt(cc) + std::fabs(o.geom[3]);
        if (!(reach <= cmax)) cmax = reach;                   // NaN/inf propagate: every ray then takes the exact sweep
        fx[k] = (float)cx; fy[k] = (float)cy; fz[k] = (float)cz; fw[k] = (float)(cc - spheres[k].rr);
    }
    if (!(cmax < 1.0e14)) {
        // non-finite or enormous sphere data: the f32 records cannot represent it.  Every record becomes
        // "always a candidate" (w = -1e30), so every ray overflows its queue and takes the exact f64 sweep.
        for (size_t k = 0; k < ns4; ++k) { fx[k] = fy[k] = fz[k] = 0.f; fw[k] = -1.0e30f; }
    }
    for (size_t k = 0; k < ns4; k += 2) {
        sph32[k] = make_float4(fx[k], fx[k + 1], fy[k], fy[k + 1]);
        sph32[k + 1] = make_float4(fz[k], fz[k + 1], fw[k], fw[k + 1]);
    }
    // triangle filter records (rtx_device.h "triangle filter"): only triangles that can be hit at all.  Triangle::contains
    // decides a hit from two coordinates of the hit point -- rows (i, j) of its elimination: (x, y) unless a zero pivot
    // swaps another row in (triangle.rs:60-71,81-87; e.g. every face in a plane x = const is solved in (y, z)) -- so the
    // triangle enters the tree with its footprint in THAT coordinate plane, unbounded along the third axis (rtx_bvh.h).
    // Triangles solved in (x, y) also get a real filter record; the others a pass-all record (their leaf box already
    // says "the ray passes over the footprint").  An ill-conditioned projection (the rounding of a, b could report a
    // hit outside the footprint) keeps a triangle outside the tree: tested for every segment.
    struct TriRec { float4 A, B, g0, g1; uint32_t tri; };
    std::vector<TriRec> tree_recs[3], always_recs;                     // tree_recs[f]: the elimination does not read axis f
    std::vector<BvhBox> tri_boxes[3];
    double tri_extent = 0.0;
    for (size_t k = 0; k < tris.size(); ++k) {
        const TriX &t = tris[k];
        const RtxObject &o = scene->objects[t.id];
        if (t.degenerate) continue;                                   // Triangle::contains is always false (triangle.rs:64,83)
        const double kabs = dot(t.n, t.v0);
        if (!std::isfinite(kabs) || !std::isfinite(t.n.x) || !std::isfinite(t.n.y) || !std::isfinite(t.n.z))
            continue;                                                 // NaN normal: every comparison of the exact test fails
        if (kabs < -(1.0 + 1e-9)) continue;                           // n.(v0 - dir) < 0 for every unit dir (triangle.rs:115)
        double v[3][3];
        bool finite = true;
        for (int c = 0; c < 9; ++c) { v[c / 3][c % 3] = o.geom[c] - centre[c % 3]; finite = finite && std::isfinite(o.geom[c]); }
        const int free_axis = 3 - (int)t.i - (int)t.j;                // t.i != t.j: the two rows the elimination reads
        const int a0 = free_axis == 0 ? 1 : 0, a1 = free_axis == 2 ? 1 : 2;
        const double r0 = v[1][a0] - v[0][a0], r1 = v[1][a1] - v[0][a1], s0 = v[2][a0] - v[0][a0], s1 = v[2][a1] - v[0][a1];
        const double det = r0 * s1 - r1 * s0;
        BvhBox fp;
        const bool in_tree = finite && t.i != t.j && std::fabs(det) > 1e-6 * std::hypot(r0, r1) * std::hypot(s0, s1) &&
                             triangle_footprint(o.geom, fp, free_axis);
        TriRec rec;
        rec.A = make_float4(0.f, 0.f, 0.f, 0.f);                       // "always a candidate"
        rec.B = make_float4(0.f, 0.f, 0.f, 0.f);
        rec.g0 = make_float4(0.f, 0.f, 0.f, 0.f);
        rec.g1 = make_float4(0.f, 0.f, NAN, 0.f);                      // n.v0 = NaN: no f32 bounds for this record (outside the tree)
        rec.tri = (uint32_t)k;
        if (in_tree) {
            // (a, b) = M (q - v0)_uv with M the inverse of [r s] in the two rows Triangle::contains' elimination reads
            // (triangle.rs:55-100; (u, v) = (x, y), (x, z) or (y, z)); n.v0 in absolute coordinates for the cull test
            // (triangle.rs:115); the plane's code for tri_bounds (rtx_mesh_step.h)
            const double m00 = s1 / det, m01 = -s0 / det, m10 = -r1 / det, m11 = r0 / det;
            const double kc = t.n.x * v[0][0] + t.n.y * v[0][1] + t.n.z * v[0][2];
            rec.g0 = make_float4((float)v[0][a0], (float)v[0][a1], (float)m00, (float)m01);
            rec.g1 = make_float4((float)m10, (float)m11, (float)kabs, (float)(2 - free_axis));
            rec.A = make_float4((float)t.n.x, (float)t.n.y, (float)t.n.z, (float)kc);
            rec.B = make_float4(0.f, 0.f, 1.0e30f, 1.0e30f);           // no (x, y) rectangle to test: every ray passes the footprint filter
            for (int c = 0; c < 9; ++c) tri_extent = std::fmax(tri_extent, std::fabs(v[c / 3][c % 3]));
        }
        if (in_tree && free_axis == 2) {
            const double xlo = std::fmin(v[0][0], std::fmin(v[1][0], v[2][0])), xhi = std::fmax(v[0][0], std::fmax(v[1][0], v[2][0]));
            const double ylo = std::fmin(v[0][1], std::fmin(v[1][1], v[2][1])), yhi = std::fmax(v[0][1], std::fmax(v[1][1], v[2][1]));
            const double grow = 1.0 + 1.0 / 1048576.0;
            rec.B = make_float4((float)(0.5 * (xlo + xhi)), (float)(0.5 * (ylo + yhi)),
                                round_up_f32(0.5 * (xhi - xlo) * grow + 1e-30), round_up_f32(0.5 * (yhi - ylo) * grow + 1e-30));
        }
        if (in_tree) { tree_recs[free_axis].push_back(rec); tri_boxes[free_axis].push_back(fp); }
        else always_recs.push_back(rec);
    }
    // a plane with fewer than 5 triangles gets no sub-tree: those are tested for every segment ((x, y) ones keep their
    // filter record, which is valid outside the tree too)
    std::vector<TriRec> demoted;
    for (int f = 0; f < 3; ++f)
        if (tree_recs[f].size() <= 4) {
            for (const TriRec &r : tree_recs[f]) (f == 2 ? demoted : always_recs).push_back(r);
            tree_recs[f].clear(); tri_boxes[f].clear();
        }

    // flat BVH over the sphere boxes and the triangle footprints (rtx_bvh.h); fewer than 5 spheres, or non-finite
    // spheres, get no sub-tree (the BVH kernel then tests those shapes for every segment)
    std::vector<BvhBox> sphere_boxes(spheres.size());
    bool spheres_finite = true;
    for (size_t k = 0; k < spheres.size() && spheres_finite; ++k)
        spheres_finite = sphere_box(scene->objects[sphere_id[k]].geom, sphere_boxes[k]);
    if (!spheres_finite || sphere_boxes.size() <= 4) sphere_boxes.clear();
    const bool use_sah = (scene->config.tuning & RTX_TUNE_BVH_MEDIAN) == 0u;
    BvhBuild &bvh = p.bvh;
    const bool plain_tree = sphere_boxes.empty() && tri_boxes[0].empty() && tri_boxes[1].empty();   // (x, y) footprints alone (free axis 2)
    bvh = build_bvh(sphere_boxes, tri_boxes, tri_leaf_size(plain_tree, scene->config.tuning), use_sah);
    Bvh4Build &bvh4 = p.bvh4;
    bvh4 = collapse_to_bvh4(bvh);

    // records in leaf order first (a triangle leaf's link indexes them; tri_order indexes [plane xy][xz][yz]), then the
    // ones outside the tree
    std::vector<float4> &tri32 = p.tri32;
    std::vector<uint32_t> &tri_fidx = p.tri_fidx;
    std::vector<TriRec> all_tree;
    std::vector<uint8_t> all_free;
    for (int f = 2; f >= 0; --f)
        for (const TriRec &r : tree_recs[f]) { all_tree.push_back(r); all_free.push_back((uint8_t)f); }
    tri32.reserve(2 * (all_tree.size() + demoted.size() + always_recs.size()) + 2);
    p.tri_rec_free_axis.clear();
    size_t n_in_tree = 0;
    if (bvh.has_tris) {
        for (uint32_t idx : bvh.tri_order) {
            const TriRec &r = all_tree[idx];
            tri32.push_back(r.A); tri32.push_back(r.B); tri_fidx.push_back(r.tri); p.tri_rec_free_axis.push_back(all_free[idx]);
            p.tri_geo.push_back(r.g0); p.tri_geo.push_back(r.g1);
        }
        n_in_tree = bvh.tri_order.size();
    } else {                                    // (the build was refused: coordinates too large for the f32 slab test)
        for (size_t k = 0; k < all_tree.size(); ++k) (all_free[k] == 2 ? demoted : always_recs).push_back(all_tree[k]);
    }
    p.sv.n_tri_tree = (uint32_t)n_in_tree;
    for (const TriRec &r : demoted) { tri32.push_back(r.A); tri32.push_back(r.B); tri_fidx.push_back(r.tri); }
    for (const TriRec &r : always_recs) { tri32.push_back(r.A); tri32.push_back(r.B); tri_fidx.push_back(r.tri); }
    if (!tri32.empty()) {                       // one pad record: bvh_step reads records in pairs
        tri32.push_back(make_float4(0.f, 0.f, 0.f, 0.f));
        tri32.push_back(make_float4(0.f, 0.f, 0.f, 0.f));
    }
    p.sv.n_tri_filter = (uint32_t)tri_fidx.size();
    p.sv.tri_extent = tri_extent;
    p.sv.n_objects = (uint32_t)n;
    p.sv.n_spheres = (uint32_t)spheres.size();
    p.sv.n_planes = (uint32_t)planes.size();
    p.sv.n_tris = (uint32_t)tris.size();
    for (int c = 0; c < 3; ++c) p.sv.sphere_center[c] = centre[c];
    p.sv.sphere_cmax = cmax;

    p.sv.n_bvh_nodes = (uint32_t)bvh4.nodes.size();
    p.sv.bvh_depth = (uint32_t)bvh4.depth;
    p.sv.bvh_root = bvh4.root;
    p.sv.bvh_origin_limit = (float)bvh.origin_limit;
    p.sv.bvh_inv_max = (float)std::fmin(1.0e30, 1.0e37 / std::fmax(bvh.origin_limit, 1.0));     // |o * inv|, |b * inv| stay finite in f32
    p.sv.bvh_flags = (bvh.has_spheres ? 1u : 0u) | (bvh.has_tris ? 2u : 0u) |
                     (bvh.has_tris && !bvh.has_spheres && tree_recs[1].empty() && tree_recs[0].empty() ? 4u : 0u);
    if ((p.sv.bvh_flags & 4u) && build_qnodes(bvh4, p.qnodes)) p.sv.bvh_flags |= 8u;
    else p.qnodes.clear();
#ifndef RTX_LAB
    // the product library holds the pure-footprint kernels in their 64-byte-node instances only: a tree whose nodes have no such
    // form (coordinates beyond the quantisation's range) walks as a joint tree, which takes any node
    if ((p.sv.bvh_flags & 12u) == 4u) p.sv.bvh_flags &= ~4u;
#endif
    if ((p.sv.bvh_flags & 3u) == 1u && build_q3nodes(bvh4, bvh.abs_pad, p.q3nodes)) p.sv.bvh_flags |= 16u;     // spheres only
    else p.q3nodes.clear();
    if (debug_prints())
        std::fprintf(stderr, "[rtx_hip] upload: %zu spheres, %zu triangles (%zu in the tree: %zu xy / %zu xz / %zu yz footprints, %zu tested per segment), bvh: %zu binary nodes, %zu wide nodes, depth %d\n",
                     spheres.size(), tris.size(), n_in_tree, tree_recs[2].size(), tree_recs[1].size(), tree_recs[0].size(),
                     demoted.size() + always_recs.size(), bvh.nodes.size(), bvh4.nodes.size(), bvh4.depth);
    std::vector<float4> &leaf32 = p.leaf32;
    leaf32.assign(bvh.prims.size(), make_float4(0.f, 0.f, 0.f, 0.f));
    for (size_t k = 0; k < bvh.prims.size(); ++k) {
        const uint32_t p = bvh.prims[k];
        leaf32[k] = make_float4(fx[p], fy[p], fz[p], fw[p]);
    }
    // the spheres kernel's leaf record {c - centre, |r|} (sphere.rs:24 squares the radius: its sign does not matter);
    // |r| rounded up, which only adds candidates
    p.leaf_cr.assign(bvh.prims.size(), make_float4(0.f, 0.f, 0.f, 0.f));
    for (size_t k = 0; k < bvh.prims.size(); ++k) {
        const uint32_t q = bvh.prims[k];
        p.leaf_cr[k] = make_float4(fx[q], fy[q], fz[q], round_up_f32(std::fabs(scene->objects[sphere_id[q]].geom[3])));
    }

    return RTX_OK;
}

// Puts a packed scene (shape arrays, filter records, BVH) on the handle's device, replacing what was there.  Config
// and camera of the handle are kept.  The new arrays are uploaded first and swapped in on success, so a failure
// leaves the resident scene as it was.
int32_t upload_packed(RtxSceneHandle_ *h, const PackedScene &p)
{
    const SceneView sv_before = h->sv;                    // restored if an upload fails: the handle then still holds the old scene
    std::vector<void *> old_allocs;
    old_allocs.swap(h->scene_allocs);
    {   // keep what apply_config / the camera put into h->sv, take the rest from the packed scene
        SceneView sv = p.sv;
        sv.rays_per_pixel = h->sv.rays_per_pixel; sv.max_bounces = h->sv.max_bounces;
        sv.focal_length = h->sv.focal_length; sv.focal_offset = h->sv.focal_offset; sv.non_focal_offset = h->sv.non_focal_offset;
        sv.seed = h->sv.seed; sv.tuning = h->sv.tuning;
        h->sv = sv;
    }
    h->sv.cam_pos = mk(h->cam.position[0], h->cam.position[1], h->cam.position[2]);
    h->sv.to_world_x = mk(h->cam.to_world_space[0], h->cam.to_world_space[1], h->cam.to_world_space[2]);
    h->sv.to_world_y = mk(h->cam.to_world_space[3], h->cam.to_world_space[4], h->cam.to_world_space[5]);
    h->sv.to_world_z = mk(h->cam.to_world_space[6], h->cam.to_world_space[7], h->cam.to_world_space[8]);

    int32_t rc = RTX_OK;
    if (!rc) rc = upload_vec(h, p.bvh4.nodes, &h->sv.bvh_nodes);
    if (!rc) rc = upload_vec(h, p.qnodes, &h->sv.bvh_qnodes);
    if (!rc) rc = upload_vec(h, p.q3nodes, &h->sv.bvh_q3nodes);
    if (!rc) rc = upload_vec(h, p.bvh.prims, &h->sv.bvh_prims);
    if (!rc) rc = upload_vec(h, p.leaf32, &h->sv.bvh_leaf_f32);
    if (!rc) rc = upload_vec(h, p.leaf_cr, &h->sv.bvh_leaf_cr);
    if (!rc) rc = upload_vec(h, p.spheres, &h->sv.spheres);
    if (!rc) rc = upload_vec(h, p.sphere_id, &h->sv.sphere_id);
    if (!rc) rc = upload_vec(h, p.planes, &h->sv.planes);
    if (!rc) rc = upload_vec(h, p.tris, &h->sv.tris);
    if (!rc) rc = upload_vec(h, p.mats, &h->sv.materials);
    if (!rc) rc = upload_vec(h, p.sph32, &h->sv.sphere_f32);
    if (!rc) rc = upload_vec(h, p.tri32, &h->sv.tri_f32);
    if (!rc) rc = upload_vec(h, p.tri_fidx, &h->sv.tri_fidx);
    if (!rc) rc = upload_vec(h, p.tri_geo, &h->sv.tri_geo);
    if (rc) {
        for (void *d : h->scene_allocs) (void)hipFree(d);
        h->scene_allocs.swap(old_allocs);
        h->sv = sv_before;
        return rc;
    }
    for (void *d : old_allocs) (void)hipFree(d);
    h->sv_dirty = true;
    return RTX_OK;
}

int32_t install_scene(RtxSceneHandle_ *h, const RtxScene *scene)
{
    PackedScene p;
    if (int32_t prc = pack_scene(scene, p)) return prc;
    return upload_packed(h, p);
}

int32_t check_scene_args(const RtxScene *scene, const char *who)
{
    if (!scene) return fail(RTX_ERR_INVALID_ARGUMENT, std::string(who) + ": null scene");
    if (scene->n_objects && !scene->objects) return fail(RTX_ERR_INVALID_ARGUMENT, std::string(who) + ": objects is null");
    if (scene->n_objects > 0xFFFFFFF0ull) return fail(RTX_ERR_INVALID_ARGUMENT, std::string(who) + ": too many objects");
    return check_config(scene->config);
}

int32_t check_device(int32_t device, const char *who)
{
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) {
        (void)hipGetLastError();
        return fail(RTX_ERR_NO_DEVICE, "no HIP device visible; librtx_hip has no CPU fallback");
    }
    if (device < 0 || device >= n_dev) return fail(RTX_ERR_INVALID_ARGUMENT, std::string(who) + ": bad device index");
    if (!device_is_gfx950(device)) return fail(RTX_ERR_NO_DEVICE, "device is not gfx950 (MI355X); this library targets gfx950 only");
    return RTX_OK;
}

// A handle on `device` holding an already packed scene (one pack serves every device of rtx_render_devices).
int32_t create_handle(const RtxScene *scene, const PackedScene &p, int32_t device, RtxSceneHandle_ **out)
{
    *out = nullptr;
    RTX_HIP_CHECK(hipSetDevice(device));
    RtxSceneHandle_ *h = new RtxSceneHandle_();
    h->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { free_handle(h); return fail(RTX_ERR_HIP, "hipGetDeviceProperties failed"); }
    h->n_cus = prop.multiProcessorCount;
    h->cam = scene->camera;
    apply_config(h, scene->config);
    if (int32_t irc = upload_packed(h, p)) { free_handle(h); return irc; }
    h->objects.assign(scene->objects, scene->objects + scene->n_objects);

    hipError_t e = hipMalloc((void **)&h->counters, sizeof(Counters) * kCounterShards);
    if (e == hipSuccess) e = hipMalloc((void **)&h->counters_stage1, sizeof(Counters) * kCounterShards);
    if (e == hipSuccess) e = hipMalloc((void **)&h->work_counter, sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_sv, sizeof(SceneView));
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_rv, sizeof(RowsView));
    if (e == hipSuccess) e = hipHostMalloc((void **)&h->h_watchdog, 2 * sizeof(unsigned long long), hipHostMallocDefault);   // [1]: the second half's
    if (e == hipSuccess) { h->h_watchdog[0] = h->h_watchdog[1] = 0ull; e = hipEventCreateWithFlags(&h->ev_watchdog, hipEventDisableTiming); }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming);
    for (int k = 0; k < 4 && e == hipSuccess; ++k) e = hipEventCreate(&h->ev[k]);
    if (e != hipSuccess) { free_handle(h); return fail(RTX_ERR_HIP, std::string("scene scratch: ") + hipGetErrorString(e)); }
    *out = h;
    return RTX_OK;
}

// The watchdog word of the last launch, if its copy has arrived (wait = true: drain the stream first).
int32_t check_watchdog(RtxSceneHandle_ *h, bool wait)
{
    if (!h->watchdog_pending) return RTX_OK;
    if (wait) RTX_HIP_CHECK(hipEventSynchronize(h->ev_watchdog));
    else {
        const hipError_t q = hipEventQuery(h->ev_watchdog);
        if (q == hipErrorNotReady) return RTX_OK;            // still running: looked at by a later call
        RTX_HIP_CHECK(q);
    }
    h->watchdog_pending = false;
    if ((h->h_watchdog[0] | h->h_watchdog[1]) != 0ull) {
        const unsigned long long n = h->h_watchdog[0] + h->h_watchdog[1];
        h->h_watchdog[0] = h->h_watchdog[1] = 0ull;
        return fail(RTX_ERR_HIP, "an earlier render on this scene raised its watchdog word (internal error: the sweep kernel's round "
                                 "bound, or a full survivors' queue): " + std::to_string(n) + " event(s); its image is incomplete");
    }
    return RTX_OK;
}

// Orders a call on `stream` after everything the handle enqueued on the stream it used before: the new stream waits, on
// the device, for the event the previous render recorded at its end.  The previous stream handle itself is not touched (a
// caller that rotates streams may have destroyed it) and the host does not block.
int32_t adopt_stream(RtxSceneHandle_ *h, hipStream_t stream)
{
    if (h->have_last_stream && h->last_stream != stream && h->have_done) RTX_HIP_CHECK(hipStreamWaitEvent(stream, h->ev_done, 0));
    h->last_stream = stream;
    h->have_last_stream = true;
    return RTX_OK;
}

}  // namespace

extern "C" {

int32_t rtx_version(void) { return RTX_HIP_VERSION; }

const char *rtx_last_error(void) { return g_last_error.c_str(); }

int32_t rtx_device_count(void) { return usable_device_count(); }

int32_t rtx_lab_build(void) { return kLabBuild ? 1 : 0; }

int32_t rtx_camera_new(const double position[3], const double direction[3], double fov, RtxCamera *out)
{
    if (!position || !direction || !out) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_camera_new: null argument");
    // camera.rs:42-49 derive_to_world_space_mat
    V3 cam_forward = vnorm(mk(direction[0], direction[1], direction[2]));
    V3 cam_right = cross(cam_forward, mk(0., 0., -1.));
    V3 cam_up = cross(cam_forward, cam_right);
    // Mat3x3::new(right, up, forward).transpose()  (specific_math.rs:16-21)
    V3 rx = mk(cam_right.x, cam_up.x, cam_forward.x);
    V3 ry = mk(cam_right.y, cam_up.y, cam_forward.y);
    V3 rz = mk(cam_right.z, cam_up.z, cam_forward.z);
    // inverse = adjugate / determinant  (specific_math.rs:10-14, :23-71; mat/div.rs:10-20)
    double a = rx.x, b = rx.y, c = rx.z, d = ry.x, e = ry.y, f = ry.z, g = rz.x, hh = rz.y, i = rz.z;
    double sum1 = a * e * i, sum2 = b * f * g, sum3 = c * d * hh;
    double sub1 = g * e * c, sub2 = hh * f * a, sub3 = i * d * b;
    double det = (sum1 + sum2 + sum3) - (sub1 + sub2 + sub3);
    V3 ax = mk(e * i - f * hh, c * hh - b * i, b * f - c * e);
    V3 ay = mk(f * g - d * i, a * i - c * g, c * d - a * f);
    V3 az = mk(d * hh - e * g, b * g - a * hh, a * e - b * d);
    V3 ix = vdivs(ax, det), iy = vdivs(ay, det), iz = vdivs(az, det);
    out->fov = fov;
    for (int k = 0; k < 3; ++k) { out->position[k] = position[k]; out->direction[k] = direction[k]; }
    const V3 w[3] = { rx, ry, rz }, cinv[3] = { ix, iy, iz };
    for (int r = 0; r < 3; ++r) {
        out->to_world_space[3 * r] = w[r].x; out->to_world_space[3 * r + 1] = w[r].y; out->to_world_space[3 * r + 2] = w[r].z;
        out->to_cam_space[3 * r] = cinv[r].x; out->to_cam_space[3 * r + 1] = cinv[r].y; out->to_cam_space[3 * r + 2] = cinv[r].z;
    }
    return RTX_OK;
}

int32_t rtx_scene_upload(const RtxScene *scene, int32_t device, RtxSceneHandle *out)
{
    if (!scene || !out) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_scene_upload: null argument");
    *out = nullptr;
    if (int32_t rc = check_scene_args(scene, "rtx_scene_upload")) return rc;
    if (int32_t rc = check_device(device, "rtx_scene_upload")) return rc;
    PackedScene p;
    if (int32_t rc = pack_scene(scene, p)) return rc;
    return create_handle(scene, p, device, out);
}

int32_t rtx_scene_free(RtxSceneHandle scene)
{
    if (!scene) return RTX_OK;
    (void)hipSetDevice(scene->device);
    const int32_t rc = check_watchdog(scene, true);         // the last chance to report a render that left early
    free_handle(scene);
    return rc;
}

int32_t rtx_scene_set_config(RtxSceneHandle scene, const RtxConfig *config)
{
    if (!scene || !config) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_scene_set_config: null argument");
    if (int32_t rc = check_config(*config)) return rc;
    if (int32_t rc = check_watchdog(scene, false)) return rc;
    apply_config(scene, *config);
    return RTX_OK;
}

int32_t rtx_scene_set_scratch_limit(RtxSceneHandle scene, uint64_t bytes)
{
    if (!scene) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_scene_set_scratch_limit: null scene");
    scene->scratch_limit = (size_t)bytes;
    return RTX_OK;
}

int32_t rtx_scene_set_camera(RtxSceneHandle scene, const RtxCamera *camera)
{
    if (!scene || !camera) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_scene_set_camera: null argument");
    if (int32_t rc = check_watchdog(scene, false)) return rc;
    scene->cam = *camera;
    scene->sv.cam_pos = mk(camera->position[0], camera->position[1], camera->position[2]);
    scene->sv.to_world_x = mk(camera->to_world_space[0], camera->to_world_space[1], camera->to_world_space[2]);
    scene->sv.to_world_y = mk(camera->to_world_space[3], camera->to_world_space[4], camera->to_world_space[5]);
    scene->sv.to_world_z = mk(camera->to_world_space[6], camera->to_world_space[7], camera->to_world_space[8]);
    scene->sv_dirty = true;                   // the trig tables are keyed on cam.fov and rebuilt when it changed
    return RTX_OK;
}

int32_t rtx_scene_append_objects(RtxSceneHandle scene, const RtxObject *objects, uint64_t n_objects)
{
    if (!scene || (n_objects && !objects)) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_scene_append_objects: null argument");
    if (n_objects == 0) return RTX_OK;
    if (scene->objects.size() + n_objects > 0xFFFFFFF0ull) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_scene_append_objects: too many objects");
    RTX_HIP_CHECK(hipSetDevice(scene->device));
    RTX_HIP_CHECK(hipDeviceSynchronize());                  // a render may still read the arrays about to be replaced
    if (int32_t rc = check_watchdog(scene, true)) return rc;
    std::vector<RtxObject> all = scene->objects;
    all.insert(all.end(), objects, objects + n_objects);
    RtxScene sc{};
    sc.config = scene->cfg;
    sc.camera = scene->cam;
    sc.n_objects = all.size();
    sc.objects = all.data();
    if (int32_t rc = install_scene(scene, &sc)) return rc;      // (on failure the resident scene is as it was)
    scene->objects.swap(all);
    return RTX_OK;
}

}  // extern "C"

// Rows of part `part` of `n_parts` when the frame is cut into blocks of `block` rows dealt out round-robin.
static uint32_t blocks_row_count(uint32_t height, uint32_t block, uint32_t part, uint32_t n_parts)
{
    if (block == 0 || n_parts == 0 || part >= n_parts) return 0;
    const uint64_t n_blocks = ((uint64_t)height + block - 1) / block;
    if (part >= n_blocks) return 0;
    const uint64_t mine = (n_blocks - part + n_parts - 1) / n_parts;            // blocks part, part + n_parts, ...
    uint64_t rows = mine * block;
    const uint64_t last = part + (mine - 1) * (uint64_t)n_parts;                // my last block; only the frame's last block can be partial
    if (last == n_blocks - 1) rows -= n_blocks * block - height;
    return (uint32_t)rows;
}

// The band local row k -> image row row_begin + (k / row_block) * row_stride + k % row_block, k < n_rows (rtx_device.h, image_row).
static int32_t render_band(RtxSceneHandle h, uint32_t width, uint32_t height, uint32_t row_begin, uint32_t row_stride,
                           uint32_t row_block, uint32_t n_rows, double *d_out_rgb, void *stream_, RtxStats *stats)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (n_rows == 0 || width == 0) return RTX_OK;
    if (!d_out_rgb) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_render_rows: null output");
    if ((uint64_t)n_rows * width > 0xFFFFFFF0ull) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_render_rows: more than 2^32 pixels per call");
    RTX_HIP_CHECK(hipSetDevice(h->device));
    if (int32_t rc = check_watchdog(h, false)) return rc;
    if (int32_t rc = adopt_stream(h, stream)) return rc;
    // From here on work may be enqueued on `stream`, and last_stream already names it: whatever way this call leaves -- a failed
    // HIP call after some launches, the watchdog return, the empty-scene answer -- the event a later call on ANOTHER stream waits
    // for (adopt_stream) is recorded behind everything enqueued so far, so that call never reuses the handle's scratch early.
    struct DoneGuard {
        RtxSceneHandle_ *h; hipStream_t stream;
        ~DoneGuard() {
            if (hipEventRecord(h->ev_done, stream) == hipSuccess) h->have_done = true;
            else { (void)hipGetLastError(); (void)hipDeviceSynchronize(); h->have_done = false; }   // (no event: drain instead)
        }
    } done_guard{h, stream};

    const uint32_t npix = n_rows * width;
    const uint64_t spp = h->cfg.rays_per_pixel;
    if (h->sv.n_objects == 0 && spp > 0) {
        // render_ray of an empty scene returns resulting_color = 0 for every sample (scene.rs:224-226) and avg() of
        // zeros is 0/len = +0.0: nothing to trace.  (The sweep kernel's workgroups leave as soon as a round finds no
        // live slot, which an empty scene produces at once: most sample planes would stay unwritten.)
        RTX_HIP_CHECK(hipMemsetAsync(d_out_rgb, 0, (size_t)npix * 3 * sizeof(double), stream));
        if (stats) {
            RTX_HIP_CHECK(hipStreamSynchronize(stream));
            stats->primary_rays = (uint64_t)npix * spp;
            stats->kernel = h->cfg.kernel;
        }
        return RTX_OK;
    }
    // AUTO: a BVH kernel when a tree was built at upload and few shapes stay outside it (those are tested for every
    // segment in f64), else the LDS sweep with its sphere and triangle filters.  Which BVH kernel: lock-step waves for
    // sphere scenes; the regrouping schedule when the tree holds a triangle mesh (measured, regroup vs lock-step:
    // C3 at 1 / 16 spp 116 / 129 vs 81 / 86 Mrays/s, C5 band at 16 spp 32.4 vs 11.4; C2 752 vs 1337)
    const uint64_t outside_tree = ((h->sv.bvh_flags & 1u) ? 0u : h->sv.n_spheres) +
                                  (uint64_t)(h->sv.n_tri_filter - h->sv.n_tri_tree);
    uint32_t kernel = h->cfg.kernel;
    const uint32_t tuning = h->cfg.tuning;
    const bool want_tiles = (tuning & RTX_TUNE_NO_TILES) == 0u, classic = (tuning & RTX_TUNE_BVH_CLASSIC) != 0u;
    if (kernel == RTX_KERNEL_AUTO) {
        if (h->sv.n_bvh_nodes != 0 && outside_tree <= 64) {
            const bool mesh = (h->sv.bvh_flags & 2u) != 0u && h->sv.n_tri_tree >= 1024u;
            kernel = mesh ? RTX_KERNEL_BVH_REGROUP : RTX_KERNEL_BVH;
            // a triangle mesh rendered with >= 2^20 rays: the wavefront form, whose level 0 walks a tile's
            // primary rays as one packet; the regrouping megakernel continues from its queue, or the deeper levels stay in
            // that form too (rtx_wavefront.hip).  Measured on C3 (100k triangles), wavefront vs megakernel: 1920x1080x8
            // 315 vs 166 Mrays/s, 960x540x8 272 vs 163, 480x270x8 160 vs 130; C5 band (1M triangles) 57 vs 44
            // (a joint tree -- spheres, faces solved in other planes -- takes the same form when its level 0 can walk as
            // packets: 240k axis-aligned cube faces at 1080p x 8: 74.5 vs 46.7 Mrays/s, with 2k spheres 65.0 vs 43.2)
            if (mesh && (uint64_t)npix * spp >= (1ull << 20) && want_tiles && wavefront_mesh_supported(h->sv, true))
                kernel = RTX_KERNEL_WAVEFRONT;
        } else {
            kernel = RTX_KERNEL_MIXED;
        }
    }

#ifndef RTX_LAB
    // the product library: ONE tree-kernel family per kind of tree, whichever of the three tree ids was asked for (include/rtx_hip.h,
    // "Product and lab") -- a sphere tree (with its 64-byte nodes) runs the sphere kernels, a tree that holds triangles the mesh
    // kernel (behind the packet kernels when RTX_KERNEL_WAVEFRONT can take it), no tree the LDS sweep
    if (kernel == RTX_KERNEL_BVH || kernel == RTX_KERNEL_BVH_REGROUP || kernel == RTX_KERNEL_WAVEFRONT) {
        const bool sphere_tree = h->sv.n_bvh_nodes != 0 && (h->sv.bvh_flags & 19u) == 17u;
        const bool tri_tree = h->sv.n_bvh_nodes != 0 && (h->sv.bvh_flags & 2u) != 0u;
        if (sphere_tree) kernel = RTX_KERNEL_BVH;
        else if (!tri_tree) kernel = RTX_KERNEL_MIXED;
        else if (kernel == RTX_KERNEL_BVH) kernel = RTX_KERNEL_BVH_REGROUP;
    }
#endif
    // the wavefront form exists for pure (x, y)-footprint triangle trees and (lab) for trees that hold spheres only; any other
    // scene takes the regrouping kernel
    const bool wf_mesh = wavefront_mesh_supported(h->sv, want_tiles);
    const bool wf_spheres = kLabBuild && !wf_mesh && h->sv.n_bvh_nodes != 0 && (h->sv.bvh_flags & 3u) == 1u;
    if (kernel == RTX_KERNEL_WAVEFRONT && !wf_mesh && !wf_spheres) kernel = RTX_KERNEL_BVH_REGROUP;
    // the BVH kernels' ray queue runs over 8x8 pixel tiles (a wave's 64 rays = one tile); a sample then has
    // tiles_x * tiles_y * 64 queue slots (the padding of partial tiles included), else npix
    const bool tiled = (kernel == RTX_KERNEL_BVH || kernel == RTX_KERNEL_BVH_REGROUP || kernel == RTX_KERNEL_WAVEFRONT) && want_tiles;
    const uint32_t tiles_x = tiled ? (width + 7u) / 8u : 0u;
    const uint64_t per_sample64 = tiled ? (uint64_t)tiles_x * ((n_rows + 7u) / 8u) * 64u : (uint64_t)npix;
    if (per_sample64 > 0xFFFFFFF0ull) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_render_rows: more than 2^32 ray slots per sample");
    const uint32_t per_sample = (uint32_t)per_sample64;
    // samples per launch: all of them unless the sample planes would exceed the scratch cap (or 2^32 rays)
    uint64_t batch = spp;
    // tile lists for the primary rays (sphere trees: build_tile_lists_kernel; pure footprint trees in the wavefront form:
    // build_mesh_tile_lists_kernel) -- 2 KB resp. 4 KB per tile, so only where they stay a small part of the scratch cap (decided below)
    uint64_t tile_list_bytes = 0;
    {
        // bytes per ray of a batch: the 32-byte sample record (+ the wavefront kernels' state, ~270 B)
        // (+ the survivors' queue of the sphere kernel's two-stage form, 64 B)
        const bool sph2 = kernel == RTX_KERNEL_BVH && (h->sv.bvh_flags & 2u) == 0u && h->sv.n_bvh_nodes != 0;
        const uint64_t per_ray = 4 * sizeof(double) + (kernel == RTX_KERNEL_WAVEFRONT ? wavefront_state_bytes(1u << 20, 1) >> 20 : 0) +
                                 (sph2 ? 64 + 5 : 0);
        // the cap: the handle's limit, never more than 3/4 of what is free on the device now (other handles, ranks or
        // frameworks may share it; what this handle already holds counts as free for it)
        size_t cap_bytes = h->scratch_limit ? h->scratch_limit : kDefaultScratchBytes;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t mine = h->samples_bytes + h->wf_bytes + h->tile_lists_bytes;
            const size_t avail = (free_b + mine) / 4 * 3;
            if (cap_bytes > avail) cap_bytes = avail;
        } else (void)hipGetLastError();
        // what a launch allocates whatever its batch -- the survivors' queue's chunk per resident wave and counters, the
        // wavefront form's level counters and overflow list, the HBM stack columns -- comes off the cap first; the floor is ONE
        // sample per launch (a frame cannot be cut finer), which a limit below that size gets with the overhead on top
        if ((tuning & RTX_TUNE_NO_TILE_LISTS) == 0u && tiled) {
            if (sph2) tile_list_bytes = bvh_spheres_tile_list_bytes(per_sample64);
            else if (kernel == RTX_KERNEL_WAVEFRONT && wf_mesh) tile_list_bytes = wavefront_tile_list_bytes(per_sample64);
            if (tile_list_bytes > cap_bytes / 4) tile_list_bytes = 0;           // (a frame of ~10^8 pixels: the packets walk)
        }
        const uint64_t fixed = tile_list_bytes +
                               (sph2 ? (uint64_t)bvh_spheres_queue_bytes(0, h->n_cus) + bvh_spheres_spill_bytes(h->sv, h->n_cus) : 0) +
                               (kernel == RTX_KERNEL_WAVEFRONT ? (uint64_t)wavefront_state_bytes(0, 1) + wavefront_spill_bytes(h->sv, h->n_cus) : 0) +
                               (kernel == RTX_KERNEL_MIXED || kernel == RTX_KERNEL_MIXED_VERIFY ? (uint64_t)mixed_state_bytes(h->n_cus) : 0);
        const uint64_t room = cap_bytes > fixed ? cap_bytes - fixed : 0;
        const uint64_t fit = room / (per_sample64 * per_ray);
        const uint64_t fit32 = 0xFFFFFFF0ull / per_sample64;
        if (batch > fit) batch = fit ? fit : 1;
        if (batch > fit32) batch = fit32;
    }

    // ---- trig tables of get_ray_dir (scene.rs:213-220), host libm, one value per column / local row.
    // glibc's sincos(), called by name: the reference takes f64::sin and f64::cos of the same angle, and a compiler on a GNU target
    // turns such a pair into ONE sincos call (LLVM does for rustc's output, gcc does for the test suite's CPU checker) -- whose results
    // differ from sin() / cos() in the last place on ~0.07 % of arguments (glibc 2.35).  Left to the optimiser, this code had one loop
    // merged and one not: one image row in ~100 started its primary rays an ulp off the CPU's (found by the path transcripts,
    // rtx_debug_paths).  Spelled out, both sides call the same function whatever the optimiser does.
    const size_t tdbl = 2 * (size_t)width + 2 * (size_t)n_rows;
    const bool t_same = h->t_valid && h->t_w == width && h->t_h == height && h->t_rb == row_begin &&
                        h->t_rs == row_stride && h->t_blk == row_block && h->t_nr == n_rows && h->t_fov == h->cam.fov;
    if (!t_same) {
        if (h->have_done) RTX_HIP_CHECK(hipEventSynchronize(h->ev_done));   // an earlier render's enqueued copy may still read h_tables
        if (h->h_tables_doubles < tdbl) {
            if (h->h_tables) RTX_HIP_CHECK(hipHostFree(h->h_tables));
            h->h_tables = nullptr; h->h_tables_doubles = 0;
            RTX_HIP_CHECK(hipHostMalloc((void **)&h->h_tables, tdbl * sizeof(double), hipHostMallocDefault));
            h->h_tables_doubles = tdbl;
        }
        {
            size_t have = h->tables_doubles * sizeof(double);
            if (int32_t rc = grow((void **)&h->tables, &have, tdbl * sizeof(double))) return rc;
            h->tables_doubles = have / sizeof(double);
        }
        const double fov = h->cam.fov;
        const double vertical_fov = (double)height / (double)width * fov;             // scene.rs:145
        double *sx = h->h_tables, *cx = sx + width, *sy = cx + width, *cy = sy + n_rows;
        for (uint32_t x = 0; x < width; ++x) {
            double u = (double)x / (double)width;                                       // scene.rs:157
            double angle_x = fov * (u - 0.5);                                           // scene.rs:214
            ::sincos(angle_x, &sx[x], &cx[x]);
        }
        for (uint32_t k = 0; k < n_rows; ++k) {
            const uint32_t kb = k / row_block;
            double v = (double)(row_begin + kb * row_stride + (k - kb * row_block)) / (double)height;   // scene.rs:153 (image_row)
            double angle_y = vertical_fov * (v - 0.5);                                  // scene.rs:215
            ::sincos(angle_y, &sy[k], &cy[k]);
        }
        RTX_HIP_CHECK(hipMemcpyAsync(h->tables, h->h_tables, tdbl * sizeof(double), hipMemcpyHostToDevice, stream));
        h->t_w = width; h->t_h = height; h->t_rb = row_begin; h->t_rs = row_stride; h->t_blk = row_block; h->t_nr = n_rows;
        h->t_fov = fov; h->t_valid = true;
    }

    // ---- scratch: one RGB per ray of a sample batch, the running per-pixel sum, the SoA ray state
    if (spp > 0) {
        if (int32_t rc = grow((void **)&h->samples, &h->samples_bytes, (size_t)(batch * per_sample64 * 4 * sizeof(double)))) return rc;
    }
    if (batch < spp) {
        if (int32_t rc = grow((void **)&h->acc, &h->acc_bytes, (size_t)npix * 3 * sizeof(double))) return rc;
    }
    // a tree without triangle leaves runs the spheres kernel (f32-only loop, more waves per SIMD; RTX_TUNE_BVH_CLASSIC: the
    // general lock-step kernel, for A/B runs)
    const bool spheres_kernel = kernel == RTX_KERNEL_BVH && (h->sv.bvh_flags & 2u) == 0u && h->sv.n_bvh_nodes != 0 && !classic;
    // the regrouping schedule on a tree with triangle leaves runs the mesh kernel (f32-only traversal step;
    // RTX_TUNE_BVH_CLASSIC: trace_bvh_regroup_kernel)
    const bool mesh_kernel = kernel == RTX_KERNEL_BVH_REGROUP && (h->sv.bvh_flags & 2u) != 0u && !classic;
    // the regrouping schedule on a tree without triangle leaves: the pool kernel (RTX_TUNE_BVH_CLASSIC: round 1's)
#ifdef RTX_LAB
    const bool pool_kernel = kernel == RTX_KERNEL_BVH_REGROUP && (h->sv.bvh_flags & 3u) == 1u && h->sv.n_bvh_nodes != 0 && !classic;
#endif
    if (kernel == RTX_KERNEL_BVH || kernel == RTX_KERNEL_BVH_REGROUP) {
#ifdef RTX_LAB
        const size_t need = pool_kernel ? bvh_spheres_pool_bytes(h->sv, h->n_cus)
                            : spheres_kernel ? bvh_spheres_spill_bytes(h->sv, h->n_cus)
                            : mesh_kernel ? bvh_mesh_spill_bytes(h->sv, h->n_cus) : bvh_spill_bytes(h->sv, h->n_cus);
#else
        if (!spheres_kernel && !mesh_kernel) return fail(RTX_ERR_HIP, "internal: no tree kernel for this scene (product dispatch)");
        const size_t need = spheres_kernel ? bvh_spheres_spill_bytes(h->sv, h->n_cus) : bvh_mesh_spill_bytes(h->sv, h->n_cus);
#endif
        if (int32_t rc = grow((void **)&h->state, &h->state_bytes, need)) return rc;
    }
    if (kernel == RTX_KERNEL_MIXED || kernel == RTX_KERNEL_MIXED_VERIFY) {
        if (int32_t rc = grow((void **)&h->state, &h->state_bytes, mixed_state_bytes(h->n_cus))) return rc;
    }
    // the sphere kernel's two-stage form from 2^20 rays per launch on (RTX_TUNE_ONE_STAGE: one launch, for A/B runs)
    // (the product's stage 1 exists as packets only: a queue that is not tiled, or a tree too deep for the wave-uniform stack, stays one stage)
    const bool spheres_two_stage = spheres_kernel && h->cfg.max_bounces > 0 && (tuning & RTX_TUNE_ONE_STAGE) == 0u &&
                                   (batch * per_sample64 >= (1ull << 20) || (tuning & RTX_TUNE_TWO_STAGE) != 0u) &&
                                   bvh_spheres_two_stage_ok(h->sv, tiled);
#ifdef RTX_LAB
    const bool stage2_slots = spheres_two_stage && (tuning & RTX_TUNE_STAGE2_SLOTS) != 0u && (h->sv.bvh_flags & 16u) != 0u &&
                              (tuning & (RTX_TUNE_NO_QNODES | RTX_TUNE_INLINE_LEAVES | RTX_TUNE_STAGE2_POOL | RTX_TUNE_STAGE2_PAIR)) == 0u;
    if (stage2_slots) {
        if (int32_t rc = grow(&h->slots, &h->slots_bytes, bvh_spheres_slots_bytes(h->n_cus))) return rc;
    }
#else
    const bool stage2_slots = false;
#endif
    if (spheres_two_stage) {
        if (int32_t rc = grow(&h->wf_state, &h->wf_bytes, bvh_spheres_queue_bytes(batch * per_sample64, h->n_cus))) return rc;
        if (tile_list_bytes != 0)
            if (int32_t rc = grow(&h->tile_lists, &h->tile_lists_bytes, (size_t)tile_list_bytes)) return rc;
#ifdef RTX_LAB
        if (tuning & (RTX_TUNE_STAGE2_POOL | RTX_TUNE_STAGE2_PAIR)) {
            const size_t a = bvh_spheres_pool2_bytes(h->n_cus), b = bvh_spheres_pair_bytes(h->n_cus);
            if (int32_t rc = grow(&h->pool, &h->pool_bytes, a > b ? a : b)) return rc;
        }
#endif
    }
    if (kernel == RTX_KERNEL_WAVEFRONT) {
#ifdef RTX_LAB
        const size_t need = wf_mesh ? wavefront_spill_bytes(h->sv, h->n_cus) : wavefront_spheres_spill_bytes(h->sv, h->n_cus);
#else
        const size_t need = wavefront_spill_bytes(h->sv, h->n_cus);
#endif
        if (int32_t rc = grow((void **)&h->state, &h->state_bytes, need)) return rc;
        if (int32_t rc = grow(&h->wf_state, &h->wf_bytes, wavefront_state_bytes(batch * per_sample64, wavefront_levels(h->sv)))) return rc;
        if (wf_mesh && tile_list_bytes != 0)
            if (int32_t rc = grow(&h->tile_lists, &h->tile_lists_bytes, (size_t)tile_list_bytes)) return rc;
    }

    RowsView rv{};
    rv.width = width; rv.height = height;
    rv.row_begin = row_begin; rv.row_stride = row_stride; rv.n_rows = n_rows; rv.row_block = row_block;
    rv.npix = npix;
    rv.div_width = make_fastdiv(width); rv.div_row_block = make_fastdiv(row_block); rv.div_npix = make_fastdiv(npix);
    rv.div_tiles_x = make_fastdiv(tiles_x); rv.div_per_sample = make_fastdiv(per_sample);
    rv.sin_x = h->tables; rv.cos_x = h->tables + width;
    rv.sin_y = h->tables + 2 * (size_t)width; rv.cos_y = rv.sin_y + n_rows;

    if (h->sv_dirty) {
        RTX_HIP_CHECK(hipMemcpyAsync(h->d_sv, &h->sv, sizeof(SceneView), hipMemcpyHostToDevice, stream));
        h->sv_dirty = false;
    }
    RTX_HIP_CHECK(hipMemsetAsync(h->counters, 0, sizeof(Counters) * kCounterShards, stream));
    float trace_ms = 0.f, resolve_ms = 0.f, stage1_ms = 0.f;
    unsigned long long s1_exact = 0, s1_filter = 0, s1_box = 0, prev_exact = 0, prev_filter = 0, prev_box = 0;
    uint32_t launches = 0;

    if (spp == 0) {
        // avg() of an empty iterator: 0/0 = NaN per component (scene.rs:253-259)
        rv.n_samples = 0; rv.n_rays = 0; rv.tiles_x = 0;
        RTX_HIP_CHECK(launch_resolve(nullptr, nullptr, d_out_rgb, rv, npix, 0, true, true, stream));
    }
    // ---- two halves in flight (sphere trees, two stages; RTX_TUNE_HALVES -- an experiment kept behind its bit, not the default).
    // A persistent launch ends with a drain: the last rays of the last waves have up to ten dependent rounds to go (~0.7 ms of a
    // 6.7 ms band of an 8-GPU frame).  Sample batches are independent (the left fold happens in resolve), so the launch is cut into
    // two halves of the samples with their own queues, counters and stack columns, on two streams (the second at the lowest priority:
    // the halves take turns, A1 B1 A2 B2), hoping that a later kernel's workgroups become resident as an earlier one's leave.  They
    // do not, soon enough: a draining wave keeps its slot until its last ray is done, so the chip is full of nearly empty waves, not
    // of free slots -- and two kernels sharing it are slower than one after the other (band 6.7 -> 7.8 ms, frame 46.7 -> 51.7;
    // timeline in LAB_NOTEBOOK R4.11).  The samples of both halves lie where one launch would have put them: ONE resolve, the same
    // bits (tests).  A launch counts once (RtxStats.trace_launches), trace_ms is the whole region's time, stage1_ms the time until
    // both halves' primary rays were done.
    const uint32_t sph_flags = ((tuning & RTX_TUNE_NO_PACKETS) ? 1u : 0u) | ((tuning & RTX_TUNE_SORT_SURVIVORS) ? 2u : 0u) |
                               ((tuning & RTX_TUNE_STAGE2_POOL) ? 4u : 0u) | ((tuning & RTX_TUNE_STAGE2_PAIR) ? 8u : 0u);   // (lab forms of the sphere kernel)
#ifdef RTX_LAB
    const bool halves_can = spheres_two_stage && batch == spp && spp >= 2 && !stage2_slots &&
                            (tuning & (RTX_TUNE_STAGE2_POOL | RTX_TUNE_STAGE2_PAIR)) == 0u;
#else
    const bool halves_can = spheres_two_stage && batch == spp && spp >= 2;
#endif
    const bool halves = halves_can && (tuning & RTX_TUNE_NO_HALVES) == 0u &&
                        ((tuning & RTX_TUNE_HALVES) != 0u || per_sample64 * spp <= kHalvesBelowRays);
    if (halves) {
        auto &b = h->half2;
        if (!b.stream) {
            // the LOWEST priority: when a kernel of each stream is waiting for workgroup slots, the caller's goes first -- the halves
            // take turns on the chip (A1 B1 A2 B2, each starting in the previous one's drain) instead of sharing it from the start
            int least = 0, greatest = 0;
            RTX_HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            RTX_HIP_CHECK(hipStreamCreateWithPriority(&b.stream, hipStreamNonBlocking, least));
            RTX_HIP_CHECK(hipEventCreateWithFlags(&b.fork, hipEventDisableTiming));
            RTX_HIP_CHECK(hipEventCreateWithFlags(&b.join, hipEventDisableTiming));
            RTX_HIP_CHECK(hipEventCreate(&b.stage1_done));
            RTX_HIP_CHECK(hipMalloc((void **)&b.counters, sizeof(Counters) * kCounterShards));
            RTX_HIP_CHECK(hipMalloc((void **)&b.counters_stage1, sizeof(Counters) * kCounterShards));
            RTX_HIP_CHECK(hipMalloc((void **)&b.work_counter, sizeof(unsigned long long)));
            RTX_HIP_CHECK(hipMalloc((void **)&b.d_rv, sizeof(RowsView)));
        }
        const uint64_t ns[2] = { (spp + 1) / 2, spp / 2 };
        if (int32_t rc = grow((void **)&b.state, &b.state_bytes, bvh_spheres_spill_bytes(h->sv, h->n_cus))) return rc;
        if (int32_t rc = grow(&b.queue, &b.queue_bytes, bvh_spheres_queue_bytes(ns[1] * per_sample64, h->n_cus))) return rc;
        // whatever way this block is left, the caller's stream waits for the second one (the handle's buffers are ordered on ONE stream)
        struct Join {
            RtxSceneHandle_ *h; hipStream_t main; bool forked = false;
            ~Join() {
                if (!forked) return;
                if (hipEventRecord(h->half2.join, h->half2.stream) != hipSuccess || hipStreamWaitEvent(main, h->half2.join, 0) != hipSuccess) {
                    (void)hipGetLastError(); (void)hipStreamSynchronize(h->half2.stream);
                }
            }
        } join{h, stream};
        RTX_HIP_CHECK(hipMemsetAsync(b.counters, 0, sizeof(Counters) * kCounterShards, stream));
        if (stats) RTX_HIP_CHECK(hipEventRecord(h->ev[0], stream));
        RTX_HIP_CHECK(hipEventRecord(b.fork, stream));                 // (behind the tables, the scene descriptor and the zeroed counters)
        RTX_HIP_CHECK(hipStreamWaitEvent(b.stream, b.fork, 0));
        join.forked = true;
        RowsView rvk[2] = { rv, rv };
        for (int k = 0; k < 2; ++k) {
            RowsView &r = rvk[k];
            r.sample_begin = k == 0 ? 0u : (uint32_t)ns[0];
            r.n_samples = (uint32_t)ns[k];
            r.n_rays = per_sample64 * ns[k];
            r.tiles_x = tiles_x;
            const uint64_t per_wave = r.n_rays / ((uint64_t)h->n_cus * 16u * (RTX_GRABS_PER_WAVE / 2));   // (half the rays: half the grabs)
            r.grab = (uint32_t)(per_wave >= 512 ? 512 : (per_wave <= 64 ? 64 : (per_wave & ~(uint64_t)63)));
            hipStream_t st = k == 0 ? stream : b.stream;
            RTX_HIP_CHECK(hipMemcpyAsync(k == 0 ? h->d_rv : b.d_rv, &r, sizeof(RowsView), hipMemcpyHostToDevice, st));   // pageable: staged before return
            RTX_HIP_CHECK(hipMemsetAsync(k == 0 ? h->work_counter : b.work_counter, 0, sizeof(unsigned long long), st));
            RTX_HIP_CHECK(launch_trace_bvh_spheres(h->d_sv, h->sv, k == 0 ? h->d_rv : b.d_rv, r,
                                                   h->samples + (k == 0 ? 0 : (size_t)(per_sample64 * ns[0]) * 4), k == 0 ? h->counters : b.counters,
                                                   k == 0 ? h->work_counter : b.work_counter, reinterpret_cast<uint32_t *>(k == 0 ? h->state : b.state),
                                                   h->n_cus, k == 0 ? h->wf_state : b.queue, sph_flags, st,
                                                   stats ? (k == 0 ? h->counters_stage1 : b.counters_stage1) : nullptr,
                                                   stats ? (k == 0 ? h->ev[3] : b.stage1_done) : nullptr, nullptr, nullptr,
                                                   nullptr, false));            // (the halves' packets walk: no tile lists in this form)
        }
        RTX_HIP_CHECK(hipEventRecord(b.join, b.stream));
        RTX_HIP_CHECK(hipStreamWaitEvent(stream, b.join, 0));
        join.forked = false;                                            // joined
        launches = 1;
        if (stats) RTX_HIP_CHECK(hipEventRecord(h->ev[1], stream));
        rv.sample_begin = 0; rv.n_samples = (uint32_t)spp; rv.n_rays = per_sample64 * spp; rv.tiles_x = tiles_x;
        RTX_HIP_CHECK(launch_resolve(h->samples, h->acc, d_out_rgb, rv, per_sample, spp, true, true, stream));
        if (stats) {
            RTX_HIP_CHECK(hipEventRecord(h->ev[2], stream));
            RTX_HIP_CHECK(hipEventSynchronize(h->ev[2]));
            RTX_HIP_CHECK(hipEventElapsedTime(&trace_ms, h->ev[0], h->ev[1]));
            RTX_HIP_CHECK(hipEventElapsedTime(&resolve_ms, h->ev[1], h->ev[2]));
            float c0 = 0.f, c1 = 0.f;
            RTX_HIP_CHECK(hipEventElapsedTime(&c0, h->ev[0], h->ev[3]));
            RTX_HIP_CHECK(hipEventElapsedTime(&c1, h->ev[0], b.stage1_done));
            stage1_ms = c0 > c1 ? c0 : c1;
            for (Counters *snap : { h->counters_stage1, b.counters_stage1 }) {
                Counters s1[kCounterShards];
                RTX_HIP_CHECK(hipMemcpy(s1, snap, sizeof s1, hipMemcpyDeviceToHost));
                for (int k = 0; k < kCounterShards; ++k) { s1_exact += s1[k].exact_tests; s1_filter += s1[k].filter_tests; if (k >= 2) s1_box += s1[k].pad_; }
            }
        }
    }
    for (uint64_t s0 = 0; s0 < spp && !halves; s0 += batch) {
        const uint64_t ns = (spp - s0 < batch) ? spp - s0 : batch;
        rv.sample_begin = (uint32_t)s0;
        rv.n_samples = (uint32_t)ns;
        rv.n_rays = per_sample64 * ns;                      // (the padded tile grid when tiled)
        rv.tiles_x = tiles_x;
        {
            // grabs per wave: what is left in a wave's last grab when the queue runs dry is the launch's tail.  Measured on the band
            // rank 0 of 8 owns of the C2 frame (1.67e7 rays, same box): 8: 7.33 ms, 16: 7.16, 32 and 64: 8.8 (a grab of 64 rays is one
            // atomic per tile: the one address retires ~80 M adds per second); the full frame does not care (50.4 / 50.3 / 50.2 / 50.4)
            const uint64_t per_wave = rv.n_rays / ((uint64_t)h->n_cus * 16u * RTX_GRABS_PER_WAVE);       // 16 resident waves per CU
            rv.grab = (uint32_t)(per_wave >= 512 ? 512 : (per_wave <= 64 ? 64 : (per_wave & ~(uint64_t)63)));
        }
        RTX_HIP_CHECK(hipMemcpyAsync(h->d_rv, &rv, sizeof(RowsView), hipMemcpyHostToDevice, stream));   // pageable: staged before return
        if (stats) RTX_HIP_CHECK(hipEventRecord(h->ev[0], stream));
        if (kernel == RTX_KERNEL_EXACT) {
            RTX_HIP_CHECK(launch_trace_exact(h->d_sv, h->d_rv, rv, h->samples, h->counters, stream));
#ifdef RTX_LAB
            if (h->transcript)
                RTX_HIP_CHECK(launch_trace_transcript(h->d_sv, h->d_rv, rv, h->transcript, h->transcript_counts, h->transcript_steps, stream));
#endif
        } else if (kernel == RTX_KERNEL_WAVEFRONT) {
#ifdef RTX_LAB
            if (!wf_mesh)
                RTX_HIP_CHECK(launch_trace_wavefront_spheres(h->d_sv, h->sv, h->d_rv, rv, h->samples, h->wf_state, h->counters,
                                                             reinterpret_cast<uint32_t *>(h->state), h->n_cus, stream));
            else
#endif
                RTX_HIP_CHECK(launch_trace_wavefront(h->d_sv, h->sv, h->d_rv, rv, h->samples, h->wf_state, h->counters,
                                                     reinterpret_cast<uint32_t *>(h->state), h->n_cus, stream,
                                                     wf_mesh && tile_list_bytes != 0 ? h->tile_lists : nullptr, s0 == 0));   // (a frame's later sample batches reuse its lists)
        } else if (kernel == RTX_KERNEL_BVH_REGROUP) {
            RTX_HIP_CHECK(hipMemsetAsync(h->work_counter, 0, sizeof(unsigned long long), stream));
#ifdef RTX_LAB
            if (pool_kernel)
                RTX_HIP_CHECK(launch_trace_bvh_spheres_pool(h->d_sv, h->sv, h->d_rv, rv, h->samples, h->counters, h->work_counter,
                                                            h->state, h->n_cus, stream));
            else if (!mesh_kernel)
                RTX_HIP_CHECK(launch_trace_bvh_regroup(h->d_sv, h->sv, h->d_rv, rv, h->samples, h->counters, h->work_counter,
                                                       reinterpret_cast<uint32_t *>(h->state), h->n_cus, stream));
            else
#endif
                RTX_HIP_CHECK(launch_trace_bvh_mesh(h->d_sv, h->sv, h->d_rv, rv, h->samples, h->counters, h->work_counter,
                                                    reinterpret_cast<uint32_t *>(h->state), h->n_cus, stream));
        } else if (kernel == RTX_KERNEL_BVH) {
            RTX_HIP_CHECK(hipMemsetAsync(h->work_counter, 0, sizeof(unsigned long long), stream));
#ifdef RTX_LAB
            if (!spheres_kernel)
                RTX_HIP_CHECK(launch_trace_bvh(h->d_sv, h->sv, h->d_rv, rv, h->samples, h->counters, h->work_counter,
                                               reinterpret_cast<uint32_t *>(h->state), h->n_cus, stream));
            else
#endif
                RTX_HIP_CHECK(launch_trace_bvh_spheres(h->d_sv, h->sv, h->d_rv, rv, h->samples, h->counters, h->work_counter,
                                                       reinterpret_cast<uint32_t *>(h->state), h->n_cus,
                                                       spheres_two_stage ? h->wf_state : nullptr, sph_flags, stream,
                                                       stats && spheres_two_stage ? h->counters_stage1 : nullptr,
                                                       stats && spheres_two_stage ? h->ev[3] : nullptr, spheres_two_stage && (tuning & (RTX_TUNE_STAGE2_POOL | RTX_TUNE_STAGE2_PAIR)) ? h->pool : nullptr,
                                                       stage2_slots ? h->slots : nullptr,
                                                       spheres_two_stage && tile_list_bytes != 0 ? h->tile_lists : nullptr, s0 == 0));
        } else {
            RTX_HIP_CHECK(hipMemsetAsync(h->work_counter, 0, sizeof(unsigned long long), stream));
            RTX_HIP_CHECK(launch_trace_mixed(h->d_sv, h->sv, h->d_rv, rv, h->samples, h->state, h->counters, h->work_counter, h->n_cus,
                                             kernel == RTX_KERNEL_MIXED_VERIFY, stream));
        }
        ++launches;
        if (stats) RTX_HIP_CHECK(hipEventRecord(h->ev[1], stream));
        RTX_HIP_CHECK(launch_resolve(h->samples, h->acc, d_out_rgb, rv, per_sample, spp, s0 == 0, s0 + ns == spp, stream));
        if (stats) {
            RTX_HIP_CHECK(hipEventRecord(h->ev[2], stream));
            RTX_HIP_CHECK(hipEventSynchronize(h->ev[2]));
            float a = 0.f, b = 0.f;
            RTX_HIP_CHECK(hipEventElapsedTime(&a, h->ev[0], h->ev[1]));
            RTX_HIP_CHECK(hipEventElapsedTime(&b, h->ev[1], h->ev[2]));
            trace_ms += a; resolve_ms += b;
            if (spheres_two_stage) {                        // stage 1's share of this batch (launch_trace_bvh_spheres recorded ev[3], took the snapshot)
                float c = 0.f;
                RTX_HIP_CHECK(hipEventElapsedTime(&c, h->ev[0], h->ev[3]));
                stage1_ms += c;
                Counters s1[kCounterShards];
                RTX_HIP_CHECK(hipMemcpy(s1, h->counters_stage1, sizeof s1, hipMemcpyDeviceToHost));
                // (the counters accumulate over the batches of a call: the snapshot holds batches 0..k-1 in full + stage 1 of
                //  batch k, the previous totals are subtracted below)
                unsigned long long e1 = 0, f1 = 0, b1 = 0;
                for (int k = 0; k < kCounterShards; ++k) { e1 += s1[k].exact_tests; f1 += s1[k].filter_tests; if (k >= 2) b1 += s1[k].pad_; }
                Counters now[kCounterShards];
                RTX_HIP_CHECK(hipMemcpy(now, h->counters, sizeof now, hipMemcpyDeviceToHost));
                unsigned long long e2 = 0, f2 = 0, b2 = 0;
                for (int k = 0; k < kCounterShards; ++k) { e2 += now[k].exact_tests; f2 += now[k].filter_tests; if (k >= 2) b2 += now[k].pad_; }
                s1_exact += e1 - prev_exact; s1_filter += f1 - prev_filter; s1_box += b1 - prev_box;
                prev_exact = e2; prev_filter = f2; prev_box = b2;
            }
        }
    }
    if (kernel == RTX_KERNEL_MIXED || kernel == RTX_KERNEL_MIXED_VERIFY || spheres_two_stage) {
        // the launch's watchdog word (ctr[1].pad_, sticky within this call: the sweep kernel's round bound, a survivor the
        // sphere kernel's queue could not take): mirrored to a pinned word and looked at by the next entry point that finds
        // the copy complete, whether or not stats were asked for
        if (h->watchdog_pending) RTX_HIP_CHECK(hipEventSynchronize(h->ev_watchdog));
        if (int32_t rc = check_watchdog(h, false)) return rc;
        RTX_HIP_CHECK(hipMemcpyAsync(h->h_watchdog, &h->counters[1].pad_, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        if (halves) RTX_HIP_CHECK(hipMemcpyAsync(h->h_watchdog + 1, &h->half2.counters[1].pad_, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        RTX_HIP_CHECK(hipEventRecord(h->ev_watchdog, stream));
        h->watchdog_pending = true;
    }
    // (ev_done -- what a later call on another stream waits for, adopt_stream -- is recorded by done_guard on every way out)
    if (stats) {
        RTX_HIP_CHECK(hipStreamSynchronize(stream));
        Counters host[kCounterShards];
        RTX_HIP_CHECK(hipMemcpy(host, h->counters, sizeof host, hipMemcpyDeviceToHost));
        if (halves) {                                                  // the second half counted in its own shards
            Counters more[kCounterShards];
            RTX_HIP_CHECK(hipMemcpy(more, h->half2.counters, sizeof more, hipMemcpyDeviceToHost));
            for (int k = 0; k < kCounterShards; ++k) {
                host[k].segments += more[k].segments; host[k].exact_tests += more[k].exact_tests;
                host[k].filter_tests += more[k].filter_tests; host[k].pad_ += more[k].pad_;
            }
        }
        for (int k = 0; k < kCounterShards; ++k) {
            stats->segments += host[k].segments;
            stats->exact_tests += host[k].exact_tests;
            stats->filter_tests += host[k].filter_tests;
        }
        stats->filter_mismatches = host[0].pad_;
        for (int k = 2; k < kCounterShards; ++k) stats->box_tests += host[k].pad_;
        if (host[1].pad_ != 0) {
            h->watchdog_pending = false; h->h_watchdog[0] = h->h_watchdog[1] = 0ull;        // reported right here
            return fail(RTX_ERR_HIP, "trace kernel raised its watchdog word (internal error: round bound / full survivors' queue): " +
                                         std::to_string(host[1].pad_) + " event(s)");
        }
        stats->primary_rays = (uint64_t)npix * spp;
        stats->trace_ms = trace_ms;
        stats->resolve_ms = resolve_ms;
        stats->trace_launches = launches;
        stats->kernel = kernel;                       // the kernel that actually ran (RTX_KERNEL_*)
        stats->stage1_ms = stage1_ms;
        stats->stage1_box_tests = s1_box; stats->stage1_filter_tests = s1_filter; stats->stage1_exact_tests = s1_exact;
    }
    return RTX_OK;
}

extern "C" {

int32_t rtx_render_rows(RtxSceneHandle h, uint32_t width, uint32_t height, uint32_t row_begin, uint32_t row_stride,
                        uint32_t n_rows, double *d_out_rgb, void *stream, RtxStats *stats)
{
    if (!h) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_render_rows: null scene");
    if (n_rows != 0 && width != 0) {
        if (row_stride == 0) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_render_rows: row_stride == 0");
        if ((uint64_t)row_begin + (uint64_t)(n_rows - 1) * row_stride >= height)
            return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_render_rows: rows exceed the image height");
    }
    return render_band(h, width, height, row_begin, row_stride, 1u, n_rows, d_out_rgb, stream, stats);
}

uint32_t rtx_blocks_row_count(uint32_t height, uint32_t block_rows, uint32_t part, uint32_t n_parts)
{
    return blocks_row_count(height, block_rows, part, n_parts);
}

int32_t rtx_render_blocks(RtxSceneHandle h, uint32_t width, uint32_t height, uint32_t block_rows, uint32_t part, uint32_t n_parts,
                          double *d_out_rgb, void *stream, RtxStats *stats)
{
    if (!h) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_render_blocks: null scene");
    if (block_rows == 0 || n_parts == 0 || part >= n_parts) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_render_blocks: bad partition");
    if ((uint64_t)block_rows * n_parts > 0xFFFFFFF0ull) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_render_blocks: block_rows * n_parts overflows");
    const uint32_t n_rows = blocks_row_count(height, block_rows, part, n_parts);
    return render_band(h, width, height, part * block_rows, n_parts * block_rows, block_rows, n_rows, d_out_rgb, stream, stats);
}

int32_t rtx_quantize_image_device(const double *d_rgb, uint32_t width, uint32_t height, uint8_t *d_rgb8, int32_t device,
                                  void *stream)
{
    if ((uint64_t)width * height == 0) return RTX_OK;
    if (!d_rgb || !d_rgb8) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_quantize_image_device: null argument");
    RTX_HIP_CHECK(hipSetDevice(device));
    RTX_HIP_CHECK(launch_quantize(d_rgb, d_rgb8, width, height, static_cast<hipStream_t>(stream)));
    return RTX_OK;
}

// Scene::render / render_to_image over a list of devices (one entry = rtx_render's single-GPU form).
//
// The frame is cut into blocks of kRowBlock = 8 rows dealt out round-robin (device k of n renders blocks k, k + n, ...:
// interleaving balances the uneven per-pixel cost, and 8 rows keep the BVH kernels' 8x8 ray tiles whole -- with single
// interleaved rows a "tile" of device k was 8 columns x 8n image rows, which cost the packet walks a quarter to a third of
// their rate), the scene is packed ONCE on the host and replicated, one host thread per device uploads, renders its band
// (asynchronously on the thread's own stream), for render_to_image quantises it there (3 bytes per pixel travel instead of
// 24) and hands it to the staging buffer on devices[0]; then one de-interleave kernel (with the image's vertical flip in
// the u8 form) and ONE device-to-host copy.  There is no exchange during the render (pixels are independent,
// scene.rs:149-160).
//
// The gather is hipMemcpyPeerAsync, not RCCL: inside one process the bands are n - 1 independent point-to-point
// copies into disjoint regions of one buffer -- each peer's DMA engine pushes over its own xGMI link to devices[0], which
// is what a gather does on this topology -- and a peer copy needs no communicator (ncclCommInitAll costs more than a
// C2 frame) and accepts a device list with repeated entries, which is how the path is tested on a one-GPU box.  The
// process-per-GPU form of the same partition (bench.py, rust-raytracing_amd/tiles.py) gathers with RCCL.
constexpr uint32_t kRowBlock = 8;

static int32_t render_common(const RtxScene *scene, uint32_t width, uint32_t height, const int32_t *devices, uint32_t n_dev,
                             double *out_rgb, uint8_t *out_rgb8)
{
    if (int32_t rc = check_scene_args(scene, "rtx_render")) return rc;
    const uint64_t npix = (uint64_t)width * height;
    if (npix == 0) return RTX_OK;                     // vec![vec![..; 0]; h] renders nothing (scene.rs:146)
    if (!out_rgb && !out_rgb8) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_render: null output");
    if (n_dev == 0 || !devices) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_render_devices: empty device list");
    if (n_dev > 64) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_render_devices: more than 64 devices");
    for (uint32_t k = 0; k < n_dev; ++k)
        if (int32_t rc = check_device(devices[k], "rtx_render_devices")) return rc;
    PackedScene packed;
    if (int32_t rc = pack_scene(scene, packed)) return rc;

    const int dev0 = devices[0];
    const uint32_t cap_rows = blocks_row_count(height, kRowBlock, 0, n_dev);   // rows of the largest band (part 0's)
    const size_t band_vals = (size_t)cap_rows * width * 3;
    const bool as_u8 = out_rgb == nullptr;             // render_to_image: the bands travel as bytes
    const size_t val_bytes = as_u8 ? 1 : sizeof(double);
    char *d_parts = nullptr, *d_full = nullptr;        // staging on devices[0]: n bands; the assembled frame
    RTX_HIP_CHECK(hipSetDevice(dev0));
    hipError_t e = hipMalloc((void **)&d_parts, band_vals * n_dev * val_bytes);
    if (e == hipSuccess && (n_dev > 1 || as_u8)) e = hipMalloc((void **)&d_full, npix * 3 * val_bytes);
    int32_t rc = RTX_OK;
    if (e != hipSuccess) rc = fail(RTX_ERR_OUT_OF_MEMORY, std::string("rtx_render: ") + hipGetErrorString(e));

    struct Worker { int32_t rc = RTX_OK; std::string msg; };
    std::vector<Worker> res(n_dev);
    const bool dbg = debug_prints();
    auto work = [&](uint32_t k) {
        Worker &w = res[k];
        const int dev = devices[k];
        RtxSceneHandle_ *h = nullptr;
        hipStream_t stream = nullptr;
        double *d_band = nullptr;                                       // f64 band on this device
        uint8_t *d_band8 = nullptr;                                     // its u8 form (render_to_image)
        const uint32_t n_rows = blocks_row_count(height, kRowBlock, k, n_dev);
        const size_t n_vals = (size_t)n_rows * width * 3;
        auto step = [&](int32_t r) { if (r && !w.rc) { w.rc = r; w.msg = g_last_error; } return w.rc == RTX_OK; };
        auto hip = [&](hipError_t he, const char *what) {
            if (he != hipSuccess && !w.rc) { w.rc = he == hipErrorOutOfMemory ? RTX_ERR_OUT_OF_MEMORY : RTX_ERR_HIP; w.msg = std::string(what) + ": " + hipGetErrorString(he); }
            return w.rc == RTX_OK;
        };
        if (step(create_handle(scene, packed, dev, &h)) && hip(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate")) {
            char *dst = d_parts + (size_t)k * band_vals * val_bytes;    // this band's region of the staging buffer on devices[0]
            const bool local = dev == dev0;
            if (!local) {
                int can = 0;                                            // direct xGMI DMA when the pair allows it (else the copy is staged)
                if (hipDeviceCanAccessPeer(&can, dev, dev0) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(dev0, 0);
                (void)hipGetLastError();                                // "already enabled" is not an error
            }
            if (local && !as_u8) d_band = reinterpret_cast<double *>(dst);   // render straight into the staging buffer
            else hip(hipMalloc((void **)&d_band, band_vals * sizeof(double)), "hipMalloc(band)");
            if (as_u8) {
                if (local) d_band8 = reinterpret_cast<uint8_t *>(dst);
                else hip(hipMalloc((void **)&d_band8, band_vals), "hipMalloc(band8)");
            }
            RtxStats st;
            if (w.rc == RTX_OK && n_rows) step(rtx_render_blocks(h, width, height, kRowBlock, k, n_dev, d_band, stream, dbg ? &st : nullptr));
            if (w.rc == RTX_OK && n_rows && dbg)
                std::fprintf(stderr, "[rtx_hip] device %d band %u/%u: rays %llu segments %llu exact %llu filter %llu mismatches %llu trace %.3f ms resolve %.3f ms\n",
                             dev, k, n_dev, (unsigned long long)st.primary_rays, (unsigned long long)st.segments, (unsigned long long)st.exact_tests,
                             (unsigned long long)st.filter_tests, (unsigned long long)st.filter_mismatches, st.trace_ms, st.resolve_ms);
            if (w.rc == RTX_OK && n_rows && as_u8) hip(launch_quantize_values(d_band, d_band8, n_vals, stream), "quantize_values_kernel");
            if (w.rc == RTX_OK && n_rows && !local)
                hip(hipMemcpyPeerAsync(dst, dev0, as_u8 ? (const void *)d_band8 : (const void *)d_band, dev, n_vals * val_bytes, stream),
                    "hipMemcpyPeerAsync");
            if (stream) hip(hipStreamSynchronize(stream), "hipStreamSynchronize");
        }
        if (h) step(rtx_scene_free(h));                                 // (reports the launch's watchdog word)
        if (d_band && !(dev == dev0 && !as_u8)) (void)hipFree(d_band);
        if (d_band8 && dev != dev0) (void)hipFree(d_band8);
        if (stream) (void)hipStreamDestroy(stream);
    };
    if (!rc) {
        if (n_dev == 1) work(0);
        else {
            std::vector<std::thread> threads;
            for (uint32_t k = 0; k < n_dev; ++k) threads.emplace_back(work, k);
            for (auto &t : threads) t.join();
        }
        for (uint32_t k = 0; k < n_dev && !rc; ++k)
            if (res[k].rc) rc = fail((RtxStatus)res[k].rc, "device " + std::to_string(devices[k]) + ": " + res[k].msg);
    }
    if (!rc) {
        e = hipSetDevice(dev0);
        const char *d_img = d_parts;                                    // one device, f64: the band IS the frame
        if (e == hipSuccess && as_u8) {
            e = launch_deinterleave_u8(reinterpret_cast<const uint8_t *>(d_parts), reinterpret_cast<uint8_t *>(d_full), width, height,
                                       n_dev, cap_rows, kRowBlock, true, nullptr);      // + the flip of scene.rs:176
            d_img = d_full;
        } else if (e == hipSuccess && n_dev > 1) {
            e = launch_deinterleave(reinterpret_cast<const double *>(d_parts), reinterpret_cast<double *>(d_full), width, height, n_dev,
                                    cap_rows, kRowBlock, nullptr);
            d_img = d_full;
        }
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        if (e == hipSuccess) e = hipMemcpy(as_u8 ? (void *)out_rgb8 : (void *)out_rgb, d_img, npix * 3 * val_bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(RTX_ERR_HIP, std::string("rtx_render: ") + hipGetErrorString(e));
    }
    (void)hipSetDevice(dev0);
    if (d_parts) (void)hipFree(d_parts);
    if (d_full) (void)hipFree(d_full);
    return rc;
}

int32_t rtx_render(const RtxScene *scene, uint32_t width, uint32_t height, double *out_rgb)
{
    const int32_t dev = 0;
    return render_common(scene, width, height, &dev, 1, out_rgb, nullptr);
}

int32_t rtx_render_to_image(const RtxScene *scene, uint32_t width, uint32_t height, uint8_t *out_rgb8)
{
    const int32_t dev = 0;
    return render_common(scene, width, height, &dev, 1, nullptr, out_rgb8);
}

int32_t rtx_render_devices(const RtxScene *scene, uint32_t width, uint32_t height, const int32_t *devices, uint32_t n_devices,
                           double *out_rgb)
{
    return render_common(scene, width, height, devices, n_devices, out_rgb, nullptr);
}

int32_t rtx_render_to_image_devices(const RtxScene *scene, uint32_t width, uint32_t height, const int32_t *devices,
                                    uint32_t n_devices, uint8_t *out_rgb8)
{
    return render_common(scene, width, height, devices, n_devices, nullptr, out_rgb8);
}

int32_t rtx_debug_host_scene(const RtxScene *scene, uint64_t *stats)
{
    if (!scene || !stats) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_debug_host_scene: null argument");
    if (scene->n_objects && !scene->objects) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_debug_host_scene: objects is null");
    PackedScene p;
    if (int32_t rc = pack_scene(scene, p)) return rc;
    for (int k = 0; k < 16; ++k) stats[k] = 0;
    stats[0] = p.spheres.size(); stats[1] = p.tris.size(); stats[2] = p.sv.n_tri_filter; stats[3] = p.sv.n_tri_tree;
    stats[4] = p.bvh4.nodes.size(); stats[5] = (uint64_t)p.bvh4.depth; stats[6] = p.bvh.nodes.size(); stats[7] = p.sv.bvh_flags;
    stats[12] = 3ull * (uint64_t)p.bvh4.depth + 2ull;
    if (p.bvh4.nodes.empty()) return RTX_OK;

    struct Box { float lo[3], hi[3]; };
    auto inside = [](const Box &c, const Box &o) {
        for (int a = 0; a < 3; ++a) if (!(c.lo[a] >= o.lo[a] && c.hi[a] <= o.hi[a])) return false;
        return true;
    };
    auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    const size_t n_nodes = p.bvh4.nodes.size();
    std::vector<uint8_t> node_seen(n_nodes, 0), sphere_seen(p.spheres.size(), 0), tri_seen(p.sv.n_tri_tree, 0);
    struct Item { uint32_t link; Box box; int depth; };
    std::vector<Item> todo;
    Box all;
    for (int a = 0; a < 3; ++a) { all.lo[a] = -INFINITY; all.hi[a] = INFINITY; }
    todo.push_back({p.bvh4.root, all, 1});
    int depth = 0;
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        const uint32_t idx = it.link & ~kBvhFlatNode;
        const bool flat = (it.link & kBvhFlatNode) != 0u;
        if (idx >= n_nodes) return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: link out of range");
        if (node_seen[idx]++) return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: node reached twice");
        depth = std::max(depth, it.depth);
        if (flat) stats[11] += 1;
        const Bvh4Node &w = p.bvh4.nodes[idx];
        for (int c = 0; c < 4; ++c) {
            Box b;
            uint32_t link, count;
            if (flat) {
                const float lk[4] = { w.b[0].x, w.b[0].y, w.b[0].z, w.b[0].w }, ct[4] = { w.b[1].x, w.b[1].y, w.b[1].z, w.b[1].w };
                b.lo[0] = w.a[c].x; b.lo[1] = w.a[c].y; b.hi[0] = w.a[c].z; b.hi[1] = w.a[c].w; b.lo[2] = -INFINITY; b.hi[2] = INFINITY;
                link = bits(lk[c]); count = bits(ct[c]);
            } else {
                b.lo[0] = w.a[c].x; b.lo[1] = w.a[c].y; b.lo[2] = w.a[c].z; b.hi[0] = w.b[c].x; b.hi[1] = w.b[c].y; b.hi[2] = w.b[c].z;
                link = bits(w.a[c].w); count = bits(w.b[c].w);
            }
            if (count == 0xFFFFFFFFu) continue;                                   // empty slot
            if (!inside(b, it.box)) return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: child box outside its parent's");
            if (count == 0u) { todo.push_back({link, b, it.depth + 1}); continue; }
            const uint32_t n = count & 0xFFFFu;
            stats[10] = std::max<uint64_t>(stats[10], n);
            if (count & kBvhTriLeaf) {
                for (uint32_t k = 0; k < n; ++k) {
                    const uint32_t rec = link + k;
                    if (rec >= p.sv.n_tri_tree) return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: triangle record out of range");
                    if (tri_seen[rec]++) return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: triangle in two leaves");
                    const TriX &tx = p.tris[p.tri_fidx[rec]];
                    const RtxObject &o = scene->objects[tx.id];
                    const int free_axis = p.tri_rec_free_axis[rec];
                    if (free_axis != 3 - (int)tx.i - (int)tx.j) return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: footprint plane differs from the elimination's rows");
                    if (flat && free_axis != 2) return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: a flat node holds a triangle that is not solved in (x, y)");
                    if (!(std::isinf(b.lo[free_axis]) && std::isinf(b.hi[free_axis]))) return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: triangle leaf bounded along the axis its test does not read");
                    BvhBox fp;
                    if (!triangle_footprint(o.geom, fp, free_axis)) return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: non-finite triangle in the tree");
                    for (int a = 0; a < 3; ++a)
                        if (a != free_axis && !((double)b.lo[a] <= fp.lo[a] && (double)b.hi[a] >= fp.hi[a]))
                            return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: footprint outside its leaf");
                    stats[9] += 1;
                    stats[13 + (free_axis == 2 ? 0 : 1)] += 1;          // 13: (x, y) footprints, 14: (x, z) and (y, z)
                }
            } else {
                for (uint32_t k = 0; k < n; ++k) {
                    if (link + k >= p.bvh.prims.size()) return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: sphere entry out of range");
                    const uint32_t sidx = p.bvh.prims[link + k];
                    if (sidx >= p.spheres.size() || sphere_seen[sidx]++) return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: sphere in two leaves");
                    BvhBox sb;
                    if (!sphere_box(scene->objects[p.sphere_id[sidx]].geom, sb)) return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: non-finite sphere in the tree");
                    for (int a = 0; a < 3; ++a)
                        if (!((double)b.lo[a] <= sb.lo[a] && (double)b.hi[a] >= sb.hi[a]))
                            return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: sphere outside its leaf");
                    stats[8] += 1;
                }
            }
        }
    }
    if (p.sv.bvh_flags & 8u) {                     // the 64-byte form: every decoded child rectangle contains the 128-byte node's
        if (p.qnodes.size() != n_nodes) return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: quantised node count differs");
        for (size_t k = 0; k < n_nodes; ++k) {
            const Bvh4Node &w = p.bvh4.nodes[k];
            const BvhQNode &q = p.qnodes[k];
            const float lk[4] = { w.b[0].x, w.b[0].y, w.b[0].z, w.b[0].w }, ct[4] = { w.b[1].x, w.b[1].y, w.b[1].z, w.b[1].w };
            for (int c = 0; c < 4; ++c) {
                const uint32_t count = bits(ct[c]), type = q.link[c] >> kQNodeShift;
                if ((count == 0xFFFFFFFFu) != (type == kQNodeEmpty)) return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: quantised node: empty slot differs");
                if (type == kQNodeEmpty) continue;
                if ((q.link[c] & kQNodeIndexMask) != (bits(lk[c]) & ~kBvhFlatNode) || (type == 0u) != (count == 0u) ||
                    (type != 0u && type != (count & 0xFFFFu)))
                    return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: quantised node: link / count differs");
                const double lx = (double)q.ox + (double)(q.qx[c] & 0xFFFFu) * q.sx, hx = (double)q.ox + (double)(q.qx[c] >> 16) * q.sx;
                const double ly = (double)q.oy + (double)(q.qy[c] & 0xFFFFu) * q.sy, hy = (double)q.oy + (double)(q.qy[c] >> 16) * q.sy;
                if (!(lx <= (double)w.a[c].x && ly <= (double)w.a[c].y && hx >= (double)w.a[c].z && hy >= (double)w.a[c].w))
                    return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: quantised rectangle does not contain the child's");
            }
        }
        stats[15] = p.qnodes.size();
    }
    if (p.sv.bvh_flags & 16u) {                    // a sphere tree's 64-byte form: every decoded child box contains the 128-byte node's
        if (p.q3nodes.size() != n_nodes) return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: 64-byte sphere node count differs");
        for (size_t k = 0; k < n_nodes; ++k) {
            const Bvh4Node &w = p.bvh4.nodes[k];
            const BvhQ3Node &q = p.q3nodes[k];
            const uint32_t *lows[3] = { &q.lox, &q.loy, &q.loz }, *highs[3] = { &q.hix, &q.hiy, &q.hiz };
            const float org[3] = { q.ox, q.oy, q.oz }, stp[3] = { q.sx, q.sy, q.sz };
            for (int c = 0; c < 4; ++c) {
                const uint32_t count = bits(w.b[c].w), type = q.link[c] >> kQNodeShift;
                if ((count == 0xFFFFFFFFu) != (type == kQNodeEmpty)) return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: 64-byte sphere node: empty slot differs");
                if (type == kQNodeEmpty) continue;
                if ((q.link[c] & kQNodeIndexMask) != bits(w.a[c].w) || (type == 0u) != (count == 0u) || (type != 0u && type != count))
                    return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: 64-byte sphere node: link / count differs");
                const float l[3] = { w.a[c].x, w.a[c].y, w.a[c].z }, hh[3] = { w.b[c].x, w.b[c].y, w.b[c].z };
                for (int a = 0; a < 3; ++a) {
                    // decoded exactly as the device does: one f32 FMA of an exact product
                    const float dl = std::fmaf((float)((*lows[a] >> (8 * c)) & 255u), stp[a], org[a]);
                    const float dh = std::fmaf((float)((*highs[a] >> (8 * c)) & 255u), stp[a], org[a]);
                    if (!((double)dl <= (double)l[a] - p.bvh.abs_pad * 0.999 && (double)dh >= (double)hh[a] + p.bvh.abs_pad * 0.999))
                        return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: decoded 64-byte box does not contain the child's (+ pad)");
                }
            }
        }
        stats[15] = p.q3nodes.size();
    }
    if (depth != p.bvh4.depth) return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: recorded depth differs");
    for (size_t k = 0; k < n_nodes; ++k) if (!node_seen[k]) return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: unreachable node");
    if ((p.sv.bvh_flags & 1u) && stats[8] != p.spheres.size()) return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: a sphere is in no leaf");
    if ((p.sv.bvh_flags & 2u) && stats[9] != p.sv.n_tri_tree) return fail(RTX_ERR_INVALID_ARGUMENT, "bvh check: a triangle is in no leaf");
    return RTX_OK;
}

int32_t rtx_debug_paths(RtxSceneHandle h, uint32_t width, uint32_t height, uint32_t row, uint32_t max_steps, RtxPathStep *steps,
                        uint32_t *counts)
{
#ifdef RTX_LAB
    static_assert(sizeof(RtxPathStep) == sizeof(PathStep) && sizeof(PathStep) == 64, "RtxPathStep layout");
    if (!h || !steps || !counts) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_debug_paths: null argument");
    if (width == 0 || row >= height || max_steps == 0) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_debug_paths: bad row / size");
    const uint64_t paths = (uint64_t)width * h->cfg.rays_per_pixel;
    if (paths == 0) return RTX_OK;
    if (paths * max_steps > (1ull << 24)) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_debug_paths: more than 2^24 steps asked for");
    RTX_HIP_CHECK(hipSetDevice(h->device));
    struct Bufs {                                   // freed, and the handle's pointers cleared, on every return path
        RtxSceneHandle_ *h; double *rgb = nullptr; uint32_t kernel;
        ~Bufs() {
            (void)hipDeviceSynchronize();
            if (h->transcript) (void)hipFree(h->transcript);
            if (h->transcript_counts) (void)hipFree(h->transcript_counts);
            if (rgb) (void)hipFree(rgb);
            h->transcript = nullptr; h->transcript_counts = nullptr; h->transcript_steps = 0; h->cfg.kernel = kernel;
        }
    } d{h, nullptr, h->cfg.kernel};
    RTX_HIP_CHECK(hipMalloc((void **)&h->transcript, paths * max_steps * sizeof(PathStep)));
    RTX_HIP_CHECK(hipMalloc((void **)&h->transcript_counts, paths * sizeof(uint32_t)));
    RTX_HIP_CHECK(hipMalloc((void **)&d.rgb, (size_t)width * 3 * sizeof(double)));
    RTX_HIP_CHECK(hipMemset(h->transcript, 0, paths * max_steps * sizeof(PathStep)));
    h->transcript_steps = max_steps;
    h->cfg.kernel = RTX_KERNEL_EXACT;
    RtxStats st;
    if (int32_t rc = render_band(h, width, height, row, 1u, 1u, 1u, d.rgb, nullptr, &st)) return rc;
    if (st.trace_launches != 1) return fail(RTX_ERR_UNSUPPORTED, "rtx_debug_paths: the row did not fit one launch");
    RTX_HIP_CHECK(hipMemcpy(steps, h->transcript, paths * max_steps * sizeof(PathStep), hipMemcpyDeviceToHost));
    RTX_HIP_CHECK(hipMemcpy(counts, h->transcript_counts, paths * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return RTX_OK;
#else
    (void)h; (void)width; (void)height; (void)row; (void)max_steps; (void)steps; (void)counts;
    return fail(RTX_ERR_UNSUPPORTED, "rtx_debug_paths: a lab-library hook (librtx_hip_lab.so)");
#endif
}

int32_t rtx_debug_math(int32_t op, const double *a, const double *b, double *out, uint64_t n)
{
#ifndef RTX_LAB
    (void)op; (void)a; (void)b; (void)out; (void)n;
    return fail(RTX_ERR_UNSUPPORTED, "rtx_debug_math: a lab-library hook (librtx_hip_lab.so: the same sources, the same arithmetic)");
#else
    if (n == 0) return RTX_OK;
    if (!a || !out) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_debug_math: null argument");
    if (usable_device_count() == 0) return fail(RTX_ERR_NO_DEVICE, "no gfx950 device");
    struct Bufs {                                   // freed on every return path
        double *p[3] = { nullptr, nullptr, nullptr };
        ~Bufs() { for (double *q : p) if (q) (void)hipFree(q); }
    } d;
    RTX_HIP_CHECK(hipSetDevice(0));
    for (double *&q : d.p) RTX_HIP_CHECK(hipMalloc((void **)&q, n * sizeof(double)));
    RTX_HIP_CHECK(hipMemcpy(d.p[0], a, n * sizeof(double), hipMemcpyHostToDevice));
    RTX_HIP_CHECK(hipMemcpy(d.p[1], b ? b : a, n * sizeof(double), hipMemcpyHostToDevice));
    RTX_HIP_CHECK(launch_debug_math(op, d.p[0], d.p[1], d.p[2], n, nullptr));
    RTX_HIP_CHECK(hipMemcpy(out, d.p[2], n * sizeof(double), hipMemcpyDeviceToHost));
    return RTX_OK;
#endif
}

}  // extern "C"
