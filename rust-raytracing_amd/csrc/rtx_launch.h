// rtx_launch.h -- host-callable launchers of the gfx950 kernels (rtx_kernels.hip).
#pragma once

#include "../../include/rtx_hip.h"      // RTX_TUNE_* (SceneView::tuning)
#include "rtx_device.h"

namespace rtx {

// samples: rv.n_rays records {r, g, b, 0} (32 B) in ray-queue order (store_sample, rtx_device.h), written by the trace kernels.
// d_sv / d_rv: device copies of the scene and launch descriptors (sv / rv are the host originals).
hipError_t launch_trace_exact(const SceneView *d_sv, const RowsView *d_rv, const RowsView &rv, double *samples,
                              Counters *counters, hipStream_t stream);

// MIXED kernel: persistent workgroups, f32 sphere filter staged through LDS, exact f64 re-test of
// candidates.  work_counter: one zeroed u64 on the device (ray queue head).
// state: mixed_state_bytes(n_cus) bytes of device scratch (SoA ray state).  verify: also run the exact
// sweep per segment and count disagreements into counters[0].pad_ (debug).
size_t mixed_state_bytes(int n_cus);
hipError_t launch_trace_mixed(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                              double *samples, double *state, Counters *counters, unsigned long long *work_counter,
                              int n_cus, bool verify, hipStream_t stream);

#ifdef RTX_LAB       // round 1's lock-step kernel (rtx_bvh.hip): librtx_hip_lab.so only
// BVH kernel: persistent waves, one ray per lane, per-lane stack traversal of the flat BVH (sphere boxes, triangle
// footprints) with a conservative f32 slab test; leaves and the shapes outside the tree use the exact f64 tests.  work_counter: zeroed u64 ray-queue head.
// spill: bvh_spill_bytes(sv, n_cus) bytes of device scratch for stack entries beyond the LDS stack (may be null when 0).
uint32_t bvh_spill_entries(const SceneView &sv);
size_t bvh_spill_bytes(const SceneView &sv, int n_cus);
hipError_t launch_trace_bvh(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                            double *samples, Counters *counters, unsigned long long *work_counter, uint32_t *spill, int n_cus,
                            hipStream_t stream);
#endif

// RTX_KERNEL_BVH for trees that hold spheres only (rtx_bvh_spheres.hip): an f32-only traversal loop with conservative
// distance bounds, the exact tests after the walk, 5 waves per SIMD.  spill: bvh_spheres_spill_bytes() bytes (may be 0).
uint32_t bvh_spheres_spill_entries(const SceneView &sv);
size_t bvh_spheres_spill_bytes(const SceneView &sv, int n_cus);
// queue_mem: bvh_spheres_queue_bytes(rv.n_rays, n_cus) bytes for the two-stage form (primary rays in one launch -- one
// wave-uniform packet walk per 8x8 tile when the ray queue is tiled; flags bit 0: per lane, for A/B runs -- the rays that
// survive their first hit in a second one fed from a queue of 64-byte records), or null: one launch.  A survivor the queue
// cannot take raises counters[1].pad_ (the launch's watchdog word, reported by the API).
size_t bvh_spheres_queue_bytes(uint64_t n_rays, int n_cus);
size_t bvh_spheres_tile_list_bytes(uint64_t rays_per_sample);
// The tile lists of a sphere tree (rtx_bvh_spheres.hip, "Tile lists"): per tile a count (kTileListWalk: this tile walks) and up to
// kTileListCap entries {the leaf's f32 record, its index, a lower bound of the distance at which any ray of the tile can enter its box},
// 32 bytes each -- one s_load_dwordx8 in the packet kernel.  Built by the wave-per-tile builder of rtx_wavefront.hip.
constexpr uint32_t kTileListCap = 64;
constexpr uint32_t kTileListWalk = 0xFFFFFFFFu;
struct TileEntry { float4 rec; uint32_t prim; float t_lb; uint32_t pad0, pad1; };
static_assert(sizeof(TileEntry) == 32, "TileEntry");
struct TileLists {
    uint32_t *count;                                      // [tiles]; null: no lists (every tile walks)
    TileEntry *entries;                                   // [tiles][kTileListCap]
};
hipError_t launch_build_sphere_tile_lists(const SceneView *d_sv, const RowsView *d_rv, const SceneView &sv, uint32_t n_tiles, uint32_t *count,
                                          TileEntry *entries, hipStream_t stream);
// may a launch run in two stages?  The product's stage 1 exists as packets only (tiled ray queue, a tree the wave-uniform stack
// holds); the lab library falls back to per-lane primary rays (MODE 1) and always may.
bool bvh_spheres_two_stage_ok(const SceneView &sv, bool tiled);
hipError_t launch_trace_bvh_spheres(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                                    double *samples, Counters *counters, unsigned long long *work_counter, uint32_t *spill, int n_cus,
                                    void *queue_mem, uint32_t flags, hipStream_t stream, Counters *stage1_snapshot = nullptr,
                                    hipEvent_t stage1_done = nullptr, void *pool_mem = nullptr, void *slots_mem = nullptr,
                                    void *tile_list_mem = nullptr, bool build_tile_lists = true);      // tile_list_mem: bvh_spheres_tile_list_bytes(rays per sample) bytes, or null: every tile walks;
                                    // build_tile_lists false: the lists of an earlier launch of the same frame (another sample batch) are still in it
#ifdef RTX_LAB
// slots_mem: bvh_spheres_slots_bytes(n_cus) bytes of device memory, or null: with it stage 2 of the two-stage form runs over ray
// slots (trace_sph_slots_kernel: the f64 path state of the rays in flight, one 128-byte record per slot)
size_t bvh_spheres_slots_bytes(int n_cus);
// pool_mem: bvh_spheres_pool2_bytes(n_cus) bytes: stage 2 runs as a wave-local pool of ray slots (flags bit 2: the lock-step form)
size_t bvh_spheres_pool2_bytes(int n_cus);
size_t bvh_spheres_pair_bytes(int n_cus);      // flags bit 3: stage 2 with two rays per lane; pool_mem then holds this many bytes
#endif
// (stage1_snapshot: kCounterShards Counters that receive a copy of `counters` as stage 1 left them; stage1_done: recorded
// after stage 1 -- both only for the two-stage form, both optional: what RtxStats' stage1_* fields are made of)

#ifdef RTX_LAB       // librtx_hip_lab.so only: the pool kernel (rtx_bvh_spheres_pool.hip), round 1's regrouping kernel (rtx_bvh_regroup.hip)
// RTX_KERNEL_BVH_REGROUP for trees that hold spheres only (rtx_bvh_spheres_pool.hip): every lane owns a pool of rays, the
// f64 phase serves all of them, the walk runs the lane's pending segments one after the other with the waiting lanes
// served together.  scratch: bvh_spheres_pool_bytes() bytes (the pools + the HBM stack columns).
size_t bvh_spheres_pool_bytes(const SceneView &sv, int n_cus);
hipError_t launch_trace_bvh_spheres_pool(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                                         double *samples, Counters *counters, unsigned long long *work_counter, void *scratch,
                                         int n_cus, hipStream_t stream);

// The same traversal scheduled as a per-lane state machine (rtx_bvh_regroup.hip): lanes that finished their traversal
// wait until enough of them can shade together instead of the whole wave waiting for its longest traversal.
hipError_t launch_trace_bvh_regroup(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                                    double *samples, Counters *counters, unsigned long long *work_counter, uint32_t *spill,
                                    int n_cus, hipStream_t stream);
#endif

// RTX_KERNEL_BVH_REGROUP for trees that hold triangles (rtx_bvh_mesh.hip): the regrouping schedule with an f32-only
// traversal step (certain-hit bounds, exact tests in the f64 phase).  spill: bvh_mesh_spill_bytes() bytes (may be 0).
uint32_t bvh_mesh_spill_entries(const SceneView &sv);
size_t bvh_mesh_spill_bytes(const SceneView &sv, int n_cus);
hipError_t launch_trace_bvh_mesh(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                                 double *samples, Counters *counters, unsigned long long *work_counter, uint32_t *spill, int n_cus,
                                 hipStream_t stream);

// The same kernel fed from a queue of rays in flight instead of generating primary rays (the hybrid of rtx_wavefront.hip:
// level 0 in the wavefront form, the deeper levels here).  src: structure-of-arrays state in queue order, the rays' indices
// (ridx[k * ridx_stride]), the queue length on the device; the rays enter at path level 1.  head: a zeroed u64.
struct MeshRaySource {
    const double *pos[3], *dir[3], *res[3], *lig[3];
    const uint32_t *left;                     // the triangle the ray has just left (index in tris[]), kNone = none
    const uint32_t *ridx;
    uint32_t ridx_stride;
    const unsigned long long *count;
};
hipError_t launch_trace_bvh_mesh_from_queue(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                                            double *samples, Counters *counters, unsigned long long *head, const MeshRaySource &src,
                                            uint32_t *spill, int n_cus, hipStream_t stream);

// RTX_KERNEL_WAVEFRONT (rtx_wavefront.hip): the path of a pure (x, y)-footprint triangle tree as generate / walk / shade
// kernels per bounce level, the ray state in HBM.  state_mem: wavefront_state_bytes(rv.n_rays, ...) bytes; spill:
// wavefront_spill_bytes() bytes (may be 0).  Enqueues everything on `stream`; synchronises it only when max_bounces + 1
// exceeds 16 levels.
bool wavefront_mesh_supported(const SceneView &sv, bool tiled);   // may RTX_KERNEL_WAVEFRONT take this tree (one that holds triangles)?
size_t wavefront_state_bytes(uint64_t n_rays, uint32_t levels);
uint32_t wavefront_levels(const SceneView &sv);
size_t wavefront_spill_bytes(const SceneView &sv, int n_cus);
hipError_t launch_trace_wavefront(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                                  double *samples, void *state_mem, Counters *counters, uint32_t *spill, int n_cus, hipStream_t stream,
                                  void *tile_list_mem = nullptr, bool build_tile_lists = true);      // wavefront_tile_list_bytes(rays per sample) bytes, or null: every tile walks
size_t wavefront_tile_list_bytes(uint64_t rays_per_sample);

#ifdef RTX_LAB       // librtx_hip_lab.so only
// RTX_KERNEL_WAVEFRONT for trees that hold spheres only (rtx_wavefront_spheres.hip): walk / shade kernels per bounce level,
// the walk's lanes refilled from the level's queue.  state_mem: wavefront_state_bytes(); spill: wavefront_spheres_spill_bytes().
size_t wavefront_spheres_spill_bytes(const SceneView &sv, int n_cus);
hipError_t launch_trace_wavefront_spheres(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                                          double *samples, void *state_mem, Counters *counters, uint32_t *spill, int n_cus,
                                          hipStream_t stream);
#endif

// Folds the batch's samples into acc (scene.rs:253-259, iter_ops.rs:4-8: left fold from zeros in sample order).
// samples: rv.n_rays 32-byte records in ray-queue order (store_sample, rtx_device.h); per_sample = queue slots
// of one sample (npix, or the padded 8x8-tile grid when rv.tiles_x != 0).  first: acc starts from zero.
// last: out[p] = acc / rays_per_pixel (rays_per_pixel == 0 with rv.n_samples == 0: 0/0 = NaN, as avg() of nothing).
hipError_t launch_resolve(const double *samples, double *acc, double *out, const RowsView &rv, uint32_t per_sample,
                          uint64_t rays_per_pixel, bool first, bool last, hipStream_t stream);

// render_to_image epilogue (scene.rs:175-178)
hipError_t launch_quantize(const double *rgb, uint8_t *rgb8, uint32_t width, uint32_t height, hipStream_t stream);

// rtx_render_devices' gather epilogue: parts = n bands of cap_rows rows each (band p = the blocks of `block` image rows
// b = p, p + n, ... in order) -> the full frame; the u8 form optionally flips (render_to_image, scene.rs:176)
hipError_t launch_deinterleave(const double *parts, double *full, uint32_t width, uint32_t height, uint32_t n, uint32_t cap_rows,
                               uint32_t block, hipStream_t stream);
hipError_t launch_deinterleave_u8(const uint8_t *parts, uint8_t *full, uint32_t width, uint32_t height, uint32_t n, uint32_t cap_rows,
                                  uint32_t block, bool flip, hipStream_t stream);
// `* 256`, saturating `as u8` (scene.rs:175-178) of n values in place order (a band before it travels: no flip)
hipError_t launch_quantize_values(const double *rgb, uint8_t *rgb8, uint64_t n, hipStream_t stream);

// device evaluation of single f64 ops (tests: are / and sqrt correctly rounded, how far are sin/cos)
// op: 0 a/b, 1 sqrt(a), 2 sin(a), 3 cos(a)
#ifdef RTX_LAB
hipError_t launch_trace_transcript(const SceneView *d_sv, const RowsView *d_rv, const RowsView &rv, PathStep *steps, uint32_t *counts,
                                   uint32_t max_steps, hipStream_t stream);
#endif
#ifdef RTX_LAB
hipError_t launch_debug_math(int op, const double *a, const double *b, double *out, uint64_t n, hipStream_t stream);
#endif

}  // namespace rtx
