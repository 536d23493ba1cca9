/*
 * rtx_hip.h -- C ABI of librtx_hip.so: the MI355X (gfx950) implementation of the
 * Scene::render() hot path of Schatten2021/rust-raytracing (crate `rtx` 0.1.0).
 *
 * The reference has no FFI of its own (no `extern`, no `unsafe`); the boundary this library
 * replaces is the Rust method
 *     Scene::render(&self, width: usize, height: usize) -> Vec<Vec<Vector3>>
 *                                                         (src/raytracing/scene.rs:144-170)
 * and everything below it: render_pixel / get_ray_dir / render_ray / closest_object / ray_hit /
 * random_bounce_dir (scene.rs:194-292), Object::distance / normal_at (object.rs:37-51),
 * Sphere / Plane / Triangle distance + normal (object/sphere.rs:19-33, plane.rs:20-35,
 * triangle.rs:20-127) and the Vector3 ops they use (src/math/vector*.rs).  A Rust shim
 * binds these entry points with a plain `extern "C"` block (INTEGRATION.md shows it).
 *
 * Plain C: fixed-width integers, doubles, caller-owned memory, no C++ or torch types.
 * All entry points return 0 on success; on failure they return a non-zero RtxStatus and
 * rtx_last_error() (thread-local) describes the failure.  There is NO CPU fallback: without
 * a usable gfx950 device every render entry point fails with RTX_ERR_NO_DEVICE.
 *
 * Threads: Scene::render(&self) is re-entrant in the reference (clones share the shapes behind Arc<Mutex>,
 * object.rs:9-15); here rtx_render / rtx_render_to_image may be called from any number of threads at once (each call
 * owns its device buffers; the library keeps no global mutable state besides the thread-local error string).  An
 * RtxSceneHandle owns scratch memory and is for one thread at a time; upload one handle per rendering thread or rank.
 * Streams: a handle's device buffers are shared by all of its renders, so its work is ordered on one stream at a time --
 * a call that brings a different stream than the previous call first waits for the previous stream to drain.
 * (on the device: an event the handle owns, recorded at the end of every render -- the caller's previous stream may be
 * gone by then and is never touched again).
 * With stats == NULL rtx_render_rows / rtx_render_blocks are asynchronous; an internal error of such a render (the launch's
 * watchdog word: the sweep kernel's round bound, a full survivors' queue) is reported by the next call on the handle that
 * finds it, at the latest by rtx_scene_free.
 */
#ifndef RTX_HIP_H
#define RTX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTX_HIP_VERSION 100  /* 0.1.0, the crate version this drop-in tracks (Cargo.toml:3) */

typedef enum RtxStatus {
    RTX_OK = 0,
    RTX_ERR_INVALID_ARGUMENT = 1,
    RTX_ERR_NO_DEVICE = 2,       /* no HIP device / not gfx950 */
    RTX_ERR_HIP = 3,             /* a HIP runtime call failed; see rtx_last_error() */
    RTX_ERR_UNSUPPORTED = 4,     /* e.g. an object kind the device path has no primitive for */
    RTX_ERR_OUT_OF_MEMORY = 5
} RtxStatus;

/* Shape tags of the three built-in shapes (object/sphere.rs, plane.rs, triangle.rs).
 * User-defined CustomShape impls (object.rs:53-76) have no device primitive -> RTX_ERR_UNSUPPORTED. */
enum { RTX_SPHERE = 0, RTX_PLANE = 1, RTX_TRIANGLE = 2 };

/* Kernel selection (RtxConfig.kernel). */
enum {
    RTX_KERNEL_AUTO  = 0,  /* RTX_KERNEL_BVH when a tree was built at upload (more than 4 finite spheres and/or more than
                              4 triangles with a footprint) and at most 64 spheres/triangles stay outside it;
                              RTX_KERNEL_BVH_REGROUP instead when the tree holds a triangle mesh (>= 1024 triangles);
                              RTX_KERNEL_WAVEFRONT for such a mesh when the call renders >= 2^20 rays;
                              else RTX_KERNEL_MIXED.  All kernels produce the same bits, AUTO picks the fastest */
    RTX_KERNEL_EXACT = 1,  /* every shape test in f64, reference operation order */
    RTX_KERNEL_MIXED = 2,  /* f32 conservative LDS filter + exact f64 re-evaluation of candidates;
                              produces the same bits as RTX_KERNEL_EXACT */
    RTX_KERNEL_MIXED_VERIFY = 3, /* debug: MIXED that also runs the exact sweep per segment and counts
                              disagreements in RtxStats.filter_mismatches (must stay 0) */
    RTX_KERNEL_BVH = 4,    /* spheres and triangles found by traversal of a flat BVH built at upload (SURVEY 8f N2:
                              3-D sphere boxes; (x, y) footprints for triangles, which keeps the reference's phantom
                              hits), exact f64 leaf tests; planes and shapes outside the tree tested for every
                              segment; same bits as RTX_KERNEL_EXACT.  RtxStats.box_tests counts child boxes tested */
    RTX_KERNEL_BVH_REGROUP = 5, /* RTX_KERNEL_BVH's tree and step under a different schedule: lanes whose traversal ended
                              wait until enough of them can go on together, instead of every lane waiting for the
                              wave's longest traversal; pays off on large triangle meshes; same bits */
    RTX_KERNEL_WAVEFRONT = 6   /* the same tree walked by a kernel of its own per bounce level (f32 only; level 0 of a mesh:
                              one wave-uniform walk per 8x8 tile of primary rays), the f64 exact tests + ray_hit in a
                              second kernel, the ray state structure-of-arrays in HBM between them.  For a mesh the
                              levels after the first run in RTX_KERNEL_BVH_REGROUP's kernel, fed from the level-1 queue.
                              For trees that hold triangles (level 0 of a joint tree needs the tiled ray queue and a
                              depth <= 42) or nothing but spheres, any other scene takes RTX_KERNEL_BVH_REGROUP; same bits */
};

/* One entry of Scene.objects (scene.rs:80), flattened: Object{shape, material} (object.rs:9-15)
 * with Material{base_color, emission_color, roughness} (object.rs:78-86).  136 bytes. */
typedef struct RtxObject {
    uint32_t kind;               /* RTX_SPHERE | RTX_PLANE | RTX_TRIANGLE */
    uint32_t reserved;           /* must be 0 */
    double   geom[9];            /* sphere: position.xyz, radius           (sphere.rs:9-12)
                                    plane:  position.xyz, normal.xyz       (plane.rs:9-12)
                                    triangle: vertices[0..3].xyz           (triangle.rs:9-11) */
    double   base_color[3];
    double   emission_color[3];
    double   roughness;
} RtxObject;

/* Config (scene.rs:16-28) plus the two fields the build adds. */
typedef struct RtxConfig {
    uint64_t rays_per_pixel;
    uint64_t max_bounces;
    double   focal_length;
    double   focal_offset;
    double   non_focal_offset;
    uint64_t seed;               /* counter-RNG key; the reference's fastrand stream is unseedable */
    uint32_t kernel;             /* RTX_KERNEL_* */
    uint32_t tuning;             /* RTX_TUNE_* bits: A/B switches for tests and lab runs; 0 = what ships */
} RtxConfig;

/* RtxConfig.tuning.  Every combination renders the same bits (tests/test_gpu_parity.py::test_ab_knobs_keep_the_bits);
 * the library reads no environment variable for any of this.  The two tree-build fields (TRI_LEAF, BVH_MEDIAN) are read
 * when the tree is built (rtx_scene_upload / rtx_scene_append_objects / rtx_render), the others per render.
 *
 * Product and lab.  librtx_hip.so holds the kernels RTX_KERNEL_AUTO can reach plus RTX_KERNEL_EXACT / RTX_KERNEL_MIXED.  The
 * experiments that lost (profiles/LAB_NOTEBOOK.md) are compiled only into librtx_hip_lab.so -- the same sources with -DRTX_LAB,
 * the same ABI, loaded by the lab tests and A/B tools.  In the product library
 *   - a tuning word with a bit of RTX_TUNE_LAB_MASK (or a bit this header does not name) is refused with RTX_ERR_UNSUPPORTED by
 *     every entry point that takes a config;
 *   - RTX_KERNEL_BVH, RTX_KERNEL_BVH_REGROUP and RTX_KERNEL_WAVEFRONT each run the tree kernel that exists for the scene's tree
 *     (sphere tree: trace_sph_packet_kernel / trace_bvh_spheres_kernel; a tree that holds triangles: trace_bvh_mesh_kernel, as
 *     RTX_KERNEL_WAVEFRONT with the packet kernels in front); RtxStats.kernel reports the id whose kernels ran.  The lab library
 *     keeps one kernel family per (id, tree kind) pair, as the enumerators' comments describe. */
enum {
    RTX_TUNE_NO_TILES    = 1u << 0,  /* BVH kernels: ray queue in image rows instead of 8x8 pixel tiles (no packets then) */
    RTX_TUNE_BVH_CLASSIC = 1u << 1,  /* round 1's trace_bvh_kernel / trace_bvh_regroup_kernel instead of the f32-only steps */
    RTX_TUNE_NO_QNODES   = 1u << 2,  /* 96-byte footprint nodes instead of the 64-byte quantised ones */
    RTX_TUNE_NO_PACKETS  = 1u << 3,  /* primary rays walk per lane instead of one wave-uniform walk per tile */
    RTX_TUNE_WF_PURE     = 1u << 4,  /* RTX_KERNEL_WAVEFRONT on a pure (x, y)-footprint tree: every level in that form */
    RTX_TUNE_ONE_STAGE   = 1u << 5,  /* sphere trees: primary and bounced rays in one launch whatever the ray count */
    RTX_TUNE_TWO_STAGE   = 1u << 6,  /* sphere trees: two stages even below 2^20 rays per launch */
    RTX_TUNE_BVH_MEDIAN  = 1u << 7,  /* tree build: median splits instead of the binned SAH */
    RTX_TUNE_TRI_LEAF_SHIFT = 8,     /* bits 8..11: triangles per leaf, 1..6 (0 = default: 5 pure footprint tree, 2 joint) */
    RTX_TUNE_THRESH_SHIFT   = 12,    /* bits 12..18: lanes that wait before the regrouping kernels' f64 phase runs, 1..64
                                        (0 = default) */
    RTX_TUNE_PK_LDS_STACK = 1u << 20, /* mesh packets: the wave-uniform stack in LDS (round 2) instead of in the lanes of a VGPR */
    RTX_TUNE_STAGE2_POOL = 1u << 21, /* sphere trees, two stages: stage 2 as a wave-local pool of ray slots (trace_sph_pool_kernel;
                                        an experiment, slower than the lock-step form that ships) */
    RTX_TUNE_STAGE2_PAIR = 1u << 22, /* sphere trees, two stages: stage 2 with two rays per lane (trace_sph_pair_kernel) */
    RTX_TUNE_BEAMS = 1u << 24,       /* RTX_KERNEL_WAVEFRONT on a pure footprint tree: level 0 as beams (the lanes across nodes, one interval
                                        test per node for the tile's 64 rays: wf_trace_beam_kernel; an experiment, slower than the
                                        packets that ship: the leaf records, not the node visits, are what a mesh tile pays for) */
    RTX_TUNE_INLINE_LEAVES = 1u << 25, /* sphere trees, 64-byte nodes: a node's leaf children are bounded inside the node visit (the
                                        earlier form) instead of being pushed and visited when popped, leaf visits apart from node visits */
    RTX_TUNE_NO_CUT = 1u << 23,      /* sphere trees: every round of a lock-step wave lasts until its longest walk ends (round 2's
                                        form) instead of leaving the last few walkers to continue beside the next segments */
    RTX_TUNE_STAGE2_SLOTS = 1u << 26, /* sphere trees, two stages: stage 2 over ray slots (trace_sph_slots_kernel: a wave owns ~90 rays, a lane
                                        whose walk ends takes the next READY one from LDS, the f64 phase runs for 64 finished walks at once;
                                        round 4's experiment: 12 % fewer instructions, the same time -- LAB_NOTEBOOK R4.3) */
    RTX_TUNE_HALVES = 1u << 27,      /* sphere trees, two stages: the launch as two halves of the samples in flight on two streams
                                      * (bit-identical, measured slower: not the default) */
    RTX_TUNE_NO_HALVES = 1u << 28,   /* ... never (today's default, spelled out) */
    RTX_TUNE_NO_TILE_LISTS = 1u << 29, /* sphere trees, two stages: every packet of primary rays walks the tree (rounds 3's form) instead of
                                        * running over its tile's list of reachable spheres */
    RTX_TUNE_SORT_SURVIVORS = 1u << 19 /* sphere trees, two stages: stage 2 reads the survivors ordered by the distance at which
                                        their ray leaves the scene's box and by direction octant (a counting sort in between) */
};

#define RTX_TUNE_LAB_MASK (RTX_TUNE_BVH_CLASSIC | RTX_TUNE_NO_QNODES | RTX_TUNE_NO_PACKETS | RTX_TUNE_WF_PURE | RTX_TUNE_PK_LDS_STACK | \
                           RTX_TUNE_STAGE2_POOL | RTX_TUNE_STAGE2_PAIR | RTX_TUNE_BEAMS | RTX_TUNE_INLINE_LEAVES | RTX_TUNE_SORT_SURVIVORS | \
                           RTX_TUNE_STAGE2_SLOTS)
#define RTX_TUNE_KNOWN_MASK (RTX_TUNE_LAB_MASK | RTX_TUNE_NO_TILES | RTX_TUNE_ONE_STAGE | RTX_TUNE_TWO_STAGE | RTX_TUNE_BVH_MEDIAN | \
                             (15u << RTX_TUNE_TRI_LEAF_SHIFT) | (127u << RTX_TUNE_THRESH_SHIFT) | RTX_TUNE_NO_CUT | RTX_TUNE_HALVES | \
                             RTX_TUNE_NO_HALVES | RTX_TUNE_NO_TILE_LISTS)

/* Camera (camera.rs:7-15).  to_world_space / to_cam_space are the three ROWS of each matrix
 * (mat.rs:11-18), row-major.  Only fov, position and to_world_space are read by render
 * (scene.rs:145,196,214-221; camera.rs:65-67). */
typedef struct RtxCamera {
    double fov;
    double position[3];
    double direction[3];
    double to_cam_space[9];
    double to_world_space[9];
} RtxCamera;

/* Scene (scene.rs:78-85). */
typedef struct RtxScene {
    RtxConfig        config;
    RtxCamera        camera;
    uint64_t         n_objects;
    const RtxObject *objects;    /* Scene.objects, in order (order decides ties, scene.rs:250) */
} RtxScene;

/* Counters of one render call (optional out-parameter). */
typedef struct RtxStats {
    uint64_t primary_rays;       /* rows * width * rays_per_pixel */
    uint64_t segments;           /* closest_object queries (scene.rs:231) */
    uint64_t exact_tests;        /* f64 shape evaluations (EXACT: segments * n_objects) */
    uint64_t filter_tests;       /* f32 filter evaluations (MIXED only) */
    double   trace_ms;           /* device time of the trace kernel(s), hipEvent on the launch stream */
    double   resolve_ms;         /* device time of the per-pixel sample fold */
    uint64_t filter_mismatches;  /* RTX_KERNEL_MIXED_VERIFY only */
    uint64_t box_tests;          /* RTX_KERNEL_BVH only: ray/box slab tests (32 B node each) */
    uint32_t trace_launches;
    uint32_t kernel;             /* the RTX_KERNEL_* that ran (resolves RTX_KERNEL_AUTO) */
    /* the two-stage form of RTX_KERNEL_BVH on a sphere tree (else 0): what its stage 1 -- the primary rays, one segment
     * each -- accounts for of the totals above */
    double   stage1_ms;
    uint64_t stage1_box_tests;
    uint64_t stage1_filter_tests;
    uint64_t stage1_exact_tests;
} RtxStats;

typedef struct RtxSceneHandle_ *RtxSceneHandle;

/* The layouts a binding has to reproduce (rust/src/raytracing/hip.rs: #[repr(C)]; rust-raytracing_amd/abi.py: ctypes;
 * tests/test_abi_and_host.py checks both against these numbers). */
#ifdef __cplusplus
#define RTX_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define RTX_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif
RTX_STATIC_ASSERT(sizeof(RtxObject) == 136 && offsetof(RtxObject, geom) == 8 && offsetof(RtxObject, base_color) == 80 &&
                  offsetof(RtxObject, emission_color) == 104 && offsetof(RtxObject, roughness) == 128, "RtxObject layout");
RTX_STATIC_ASSERT(sizeof(RtxConfig) == 56 && offsetof(RtxConfig, focal_length) == 16 && offsetof(RtxConfig, seed) == 40 &&
                  offsetof(RtxConfig, kernel) == 48 && offsetof(RtxConfig, tuning) == 52, "RtxConfig layout");
RTX_STATIC_ASSERT(sizeof(RtxCamera) == 200 && offsetof(RtxCamera, position) == 8 && offsetof(RtxCamera, direction) == 32 &&
                  offsetof(RtxCamera, to_cam_space) == 56 && offsetof(RtxCamera, to_world_space) == 128, "RtxCamera layout");
RTX_STATIC_ASSERT(sizeof(RtxScene) == 272 && offsetof(RtxScene, camera) == 56 && offsetof(RtxScene, n_objects) == 256 &&
                  offsetof(RtxScene, objects) == 264, "RtxScene layout");
RTX_STATIC_ASSERT(sizeof(RtxStats) == 104 && offsetof(RtxStats, trace_ms) == 32 && offsetof(RtxStats, box_tests) == 56 &&
                  offsetof(RtxStats, trace_launches) == 64 && offsetof(RtxStats, kernel) == 68 && offsetof(RtxStats, stage1_ms) == 72 &&
                  offsetof(RtxStats, stage1_exact_tests) == 96, "RtxStats layout");

/* -- library ----------------------------------------------------------------------------- */
int32_t     rtx_version(void);
const char *rtx_last_error(void);
int32_t     rtx_device_count(void);     /* number of usable gfx950 devices (0 when none) */
int32_t     rtx_lab_build(void);        /* 0: librtx_hip.so (the product); 1: librtx_hip_lab.so (built with -DRTX_LAB, see RtxConfig.tuning) */

/* Camera::new (camera.rs:19-28, derive_to_world_space_mat :42-49, Mat3x3::inverse
 * specific_math.rs:10-14).  Host-side, f64, reference operation order. */
int32_t rtx_camera_new(const double position[3], const double direction[3], double fov, RtxCamera *out);

/* Scene::render (scene.rs:144-170).  out_rgb: height*width*3 doubles, out_rgb[(y*width+x)*3+c]
 * == reference img[y][x].{x,y,z}; unclamped; row 0 is the reference's row 0.
 * Uploads the scene to device 0, renders, copies back. */
int32_t rtx_render(const RtxScene *scene, uint32_t width, uint32_t height, double *out_rgb);

/* Scene::render_to_image (scene.rs:172-178): `* 256`, saturating `as u8`, vertical flip.
 * out_rgb8: height*width*3 bytes. */
int32_t rtx_render_to_image(const RtxScene *scene, uint32_t width, uint32_t height, uint8_t *out_rgb8);

/* Scene::render / render_to_image on several GPUs of one node (BASELINE north_star: "tile-partitioned across the 8 GPUs
 * ... single gather over xGMI at the end"): still ONE call that returns the whole frame (scene.rs:144-170).  The scene is
 * packed once and replicated; device devices[k] renders the blocks of 8 rows k, k + n, ... (rtx_render_blocks; one host
 * thread and one stream per device, no exchange during the render: pixels are independent, scene.rs:149-160); the bands --
 * for render_to_image already quantised to 3 bytes per pixel -- are gathered on devices[0] with one peer copy each
 * (hipMemcpyPeerAsync: every peer pushes over its own xGMI link; no communicator to set up), de-interleaved there and
 * copied to the host once.  Pixel values do not depend on n_devices or the order of
 * the list.  An entry may repeat (two bands on one device) -- that is how the path is tested on a one-GPU box.
 * rtx_render(scene, w, h, out) == rtx_render_devices(scene, w, h, {0}, 1, out). */
int32_t rtx_render_devices(const RtxScene *scene, uint32_t width, uint32_t height,
                           const int32_t *devices, uint32_t n_devices, double *out_rgb);
int32_t rtx_render_to_image_devices(const RtxScene *scene, uint32_t width, uint32_t height,
                                    const int32_t *devices, uint32_t n_devices, uint8_t *out_rgb8);

/* -- split form: upload once, render many (rows / devices), output stays on the device ------ */
int32_t rtx_scene_upload(const RtxScene *scene, int32_t device, RtxSceneHandle *out);
int32_t rtx_scene_free(RtxSceneHandle scene);      /* non-zero: an earlier asynchronous render on it had failed (see "Streams") */

/* Replace the Config of an uploaded scene (rays_per_pixel, seed, kernel, ...). */
int32_t rtx_scene_set_config(RtxSceneHandle scene, const RtxConfig *config);

/* Upper bound of the per-render scratch this handle allocates on its device (sample records, ray queues), in bytes;
 * 0 = the default (24 GiB, never more than 3/4 of the memory free at render time).  A frame that needs more is traced in
 * sample batches -- same bits (the fold continues across batches).  For several handles, ranks or frameworks sharing a GPU.
 * What a launch allocates whatever its batch (a queue chunk per resident wave, counters, stack columns: 0.15-0.3 GB on a
 * 256-CU part) is taken off the limit before the batch is sized.  The floor is ONE sample of the band per launch: a limit
 * below that size is exceeded by that one sample's records + the fixed part (a frame cannot be cut finer). */
int32_t rtx_scene_set_scratch_limit(RtxSceneHandle scene, uint64_t bytes);

/* Scene::add_object (scene.rs:126-128) on an uploaded scene: the objects are appended in order (they get the next scene
 * indices, so earlier objects still win ties, scene.rs:250), the shape arrays, filter records and the BVH are rebuilt on
 * the host and replace the resident ones; config, camera and scratch stay.  Waits for renders in flight on the handle's
 * device.  On failure (e.g. an unsupported kind) the resident scene is unchanged. */
int32_t rtx_scene_append_objects(RtxSceneHandle scene, const RtxObject *objects, uint64_t n_objects);

/* Replace the Camera of an uploaded scene without re-uploading the objects (the reference's analogue is
 * Scene.camera being a pub field, scene.rs:82, with Camera::set_direction camera.rs:35-40).  Only fov, position and
 * to_world_space are read by render. */
int32_t rtx_scene_set_camera(RtxSceneHandle scene, const RtxCamera *camera);

/*
 * Renders image rows row_begin, row_begin+row_stride, ... (n_rows of them, all < height) of the
 * width x height image into d_out_rgb, a DEVICE buffer of n_rows*width*3 doubles (row k of the
 * buffer is image row row_begin + k*row_stride).  Pixel values do not depend on the
 * partition: a pixel's RNG key is its index in the full image.  stream is a hipStream_t
 * (NULL = the device's null stream).  The call enqueues all work on `stream`; it returns
 * after the work is enqueued when stats == NULL, and after it completed (stream
 * synchronised, stats filled) otherwise.
 */
int32_t rtx_render_rows(RtxSceneHandle scene, uint32_t width, uint32_t height,
                        uint32_t row_begin, uint32_t row_stride, uint32_t n_rows,
                        double *d_out_rgb, void *stream, RtxStats *stats);

/*
 * The same for a band made of BLOCKS of rows -- what one rank / device of a multi-GPU job renders.  The frame is cut into
 * blocks of block_rows rows (block b = image rows [b * block_rows, min((b + 1) * block_rows, height))), dealt out
 * round-robin: part `part` of `n_parts` owns blocks part, part + n_parts, ...  d_out_rgb: DEVICE buffer of
 * rtx_blocks_row_count(height, block_rows, part, n_parts) * width * 3 doubles, the part's rows in increasing image order.
 * block_rows = 8 keeps the BVH kernels' 8x8 ray tiles whole (a tile of rtx_render_rows' single-row bands is 8 columns x
 * 8 * n_parts image rows: the packet walks lose a quarter to a third of their rate on it); the interleaving still balances
 * the uneven per-pixel cost (1080 rows on 8 parts: 17 or 16 blocks each).  Pixels do not depend on the partition.
 * rtx_render_blocks(h, w, hgt, 1, r, n, ...) == rtx_render_rows(h, w, hgt, r, n, rows, ...).
 */
uint32_t rtx_blocks_row_count(uint32_t height, uint32_t block_rows, uint32_t part, uint32_t n_parts);
int32_t rtx_render_blocks(RtxSceneHandle scene, uint32_t width, uint32_t height,
                          uint32_t block_rows, uint32_t part, uint32_t n_parts,
                          double *d_out_rgb, void *stream, RtxStats *stats);

/* Device epilogue of render_to_image on a full device image (scene.rs:175-178):
 * d_rgb height*width*3 doubles -> d_rgb8 height*width*3 bytes, flipped vertically. */
int32_t rtx_quantize_image_device(const double *d_rgb, uint32_t width, uint32_t height,
                                  uint8_t *d_rgb8, int32_t device, void *stream);

/* Test hook (lab library; the product returns RTX_ERR_UNSUPPORTED -- same sources and flags, so the arithmetic it shows is the
 * product's): evaluates one f64 operation per element on the device (op 0: a/b, 1: sqrt(a),
 * 2: sin(a), 3: cos(a), 4 / 5: sincos(a)'s two results; 6: the packet walks' v_writelane -- out[i] = (int)b[0] in lane (int)b[1]
 * of every wave, (int)a[i] elsewhere; 7 / 8: v_min_f64 / v_max_f64 on the raw bits of a[i], b[i] -- the child sort's (key, link)
 * pairs are denormal f64 patterns when the key is +0.0 and must come back bit for bit; 9 / 10: the sin / cos the path itself uses
 * for Vector3::random_direction's angle in [0, 2 pi] -- rtx_math.h sincos_2pi); a, b, out are HOST arrays of n doubles.  Used by tests/ to check that the
 * device's / and sqrt are correctly rounded and to measure how far its sin/cos are from libm. */
int32_t rtx_debug_math(int32_t op, const double *a, const double *b, double *out, uint64_t n);

/* Test hook (lab library; the product returns RTX_ERR_UNSUPPORTED): the TRANSCRIPT of every path of one image row, as the exhaustive
 * f64 kernel (RTX_KERNEL_EXACT) walks it -- per segment the ray closest_object was asked about (scene.rs:232: position, direction),
 * the winning distance and the winner's index in Scene.objects (scene.rs:243-251; -1 and +inf: no object, the path ends).  Images
 * compare hit SEQUENCES (a colour is a product of materials); a transcript compares the arithmetic itself, every bounce direction
 * and hit point to the last bit (tests: equal to the oracle's transcript when the oracle uses the device's sin / cos routine).
 * steps: HOST array [width][rays_per_pixel][max_steps]; counts: HOST array [width][rays_per_pixel], the path's number of segments
 * (steps beyond max_steps are counted, not stored). */
typedef struct RtxPathStep {
    double  position[3];
    double  direction[3];
    double  distance;
    int64_t object;
} RtxPathStep;
int32_t rtx_debug_paths(RtxSceneHandle scene, uint32_t width, uint32_t height, uint32_t row, uint32_t max_steps,
                        RtxPathStep *steps, uint32_t *counts);

/* Test hook, needs no GPU: runs the host half of rtx_scene_upload (scene packing, filter records, the SAH build of
 * the flat BVH -- SURVEY 8f row N2; the reference's analogue is gpu_state.rs:53-77) and checks the tree's
 * invariants: every child box inside its parent's, every sphere / triangle-footprint inside its leaf's box, every
 * shape of the tree in exactly one leaf, links and layout flags consistent, depth as recorded.  stats[16]:
 * 0 spheres, 1 triangles, 2 triangle filter records, 3 of them in the tree, 4 wide nodes, 5 depth, 6 binary nodes,
 * 7 flags (1 spheres in the tree, 2 triangles in the tree, 4 nothing but (x, y)-footprint triangles: every node is a footprint node, 8 the 64-byte footprint-node form exists, 16 the 64-byte form of a sphere tree exists), 8 sphere leaf entries seen, 9 triangle leaf entries seen,
 * 10 largest leaf, 11 flat (footprint) nodes, 12 stack entries bound (3*depth+2), 13 tree triangles whose footprint lies in
 * the (x, y) plane, 14 those with an (x, z) or (y, z) footprint (zero-pivot row swaps, triangle.rs:60-71,81-87), 15 nodes that also exist in the 64-byte quantised form (flag 8). */
int32_t rtx_debug_host_scene(const RtxScene *scene, uint64_t *stats);

#ifdef __cplusplus
}
#endif
#endif
