//! FFI declarations of `librtx_hip.so` -- a field-for-field mirror of `include/rtx_hip.h`.
//! UNCOMPILED here (no Rust toolchain in the image that wrote it); the layouts are checked against the header's
//! `RTX_STATIC_ASSERT`s by `tests/test_abi_and_host.py::test_rust_shim_matches_the_header`.
#![allow(dead_code)]
use std::os::raw::{c_char, c_void};

pub const RTX_SPHERE: u32 = 0;
pub const RTX_PLANE: u32 = 1;
pub const RTX_TRIANGLE: u32 = 2;
pub const RTX_KERNEL_AUTO: u32 = 0;

/// One entry of `Scene.objects` (scene.rs:80): `Object{shape, material}` flattened.  136 bytes.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtxObject {
    pub kind: u32,
    pub reserved: u32,
    pub geom: [f64; 9],
    pub base_color: [f64; 3],
    pub emission_color: [f64; 3],
    pub roughness: f64,
}

/// `Config` (scene.rs:16-28) + seed, kernel.  56 bytes.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtxConfig {
    pub rays_per_pixel: u64,
    pub max_bounces: u64,
    pub focal_length: f64,
    pub focal_offset: f64,
    pub non_focal_offset: f64,
    pub seed: u64,
    pub kernel: u32,
    /// `RTX_TUNE_*` bits: A/B switches for tests and lab runs; 0 = what ships.
    pub tuning: u32,
}

/// `Camera` (camera.rs:7-15); the matrices as three rows, row-major.  200 bytes.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtxCamera {
    pub fov: f64,
    pub position: [f64; 3],
    pub direction: [f64; 3],
    pub to_cam_space: [f64; 9],
    pub to_world_space: [f64; 9],
}

/// `Scene` (scene.rs:78-85).  272 bytes.
#[repr(C)]
pub struct RtxScene {
    pub config: RtxConfig,
    pub camera: RtxCamera,
    pub n_objects: u64,
    pub objects: *const RtxObject,
}

/// Counters of one `rtx_render_rows` call.  104 bytes.
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RtxStats {
    pub primary_rays: u64,
    pub segments: u64,
    pub exact_tests: u64,
    pub filter_tests: u64,
    pub trace_ms: f64,
    pub resolve_ms: f64,
    pub filter_mismatches: u64,
    pub box_tests: u64,
    pub trace_launches: u32,
    pub kernel: u32,
    pub stage1_ms: f64,
    pub stage1_box_tests: u64,
    pub stage1_filter_tests: u64,
    pub stage1_exact_tests: u64,
}

#[repr(C)]
pub struct RtxSceneHandleOpaque {
    _private: [u8; 0],
}
pub type RtxSceneHandle = *mut RtxSceneHandleOpaque;

extern "C" {
    pub fn rtx_version() -> i32;
    pub fn rtx_last_error() -> *const c_char;
    pub fn rtx_device_count() -> i32;
    pub fn rtx_lab_build() -> i32;
    pub fn rtx_camera_new(position: *const f64, direction: *const f64, fov: f64, out: *mut RtxCamera) -> i32;
    pub fn rtx_render(scene: *const RtxScene, width: u32, height: u32, out_rgb: *mut f64) -> i32;
    pub fn rtx_render_to_image(scene: *const RtxScene, width: u32, height: u32, out_rgb8: *mut u8) -> i32;
    pub fn rtx_render_devices(scene: *const RtxScene, width: u32, height: u32, devices: *const i32, n_devices: u32, out_rgb: *mut f64) -> i32;
    pub fn rtx_render_to_image_devices(scene: *const RtxScene, width: u32, height: u32, devices: *const i32, n_devices: u32, out_rgb8: *mut u8) -> i32;
    pub fn rtx_scene_upload(scene: *const RtxScene, device: i32, out: *mut RtxSceneHandle) -> i32;
    pub fn rtx_scene_free(scene: RtxSceneHandle) -> i32;
    pub fn rtx_scene_set_config(scene: RtxSceneHandle, config: *const RtxConfig) -> i32;
    pub fn rtx_scene_set_scratch_limit(scene: RtxSceneHandle, bytes: u64) -> i32;
    pub fn rtx_scene_append_objects(scene: RtxSceneHandle, objects: *const RtxObject, n_objects: u64) -> i32;
    pub fn rtx_scene_set_camera(scene: RtxSceneHandle, camera: *const RtxCamera) -> i32;
    pub fn rtx_render_rows(scene: RtxSceneHandle, width: u32, height: u32, row_begin: u32, row_stride: u32, n_rows: u32, d_out_rgb: *mut f64, stream: *mut c_void, stats: *mut RtxStats) -> i32;
    pub fn rtx_blocks_row_count(height: u32, block_rows: u32, part: u32, n_parts: u32) -> u32;
    pub fn rtx_render_blocks(scene: RtxSceneHandle, width: u32, height: u32, block_rows: u32, part: u32, n_parts: u32, d_out_rgb: *mut f64, stream: *mut c_void, stats: *mut RtxStats) -> i32;
    pub fn rtx_quantize_image_device(d_rgb: *const f64, width: u32, height: u32, d_rgb8: *mut u8, device: i32, stream: *mut c_void) -> i32;
}

/// The reference reports failures by panicking (scene.rs:168, object.rs:38,50); so does the shim.
pub fn check(rc: i32) {
    if rc != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(rtx_last_error()) }.to_string_lossy().into_owned();
        panic!("rtx_hip (status {rc}): {msg}");
    }
}

// ---- what the device path needs and the reference's types cannot carry -------------------------------------------------
// `Config` and `Scene` are plain `pub` structs that users build with struct literals (scene.rs:16-28, :78-85), so the shim
// adds NO field to them -- and NO process-wide state either: `Scene::render(&self)` is re-entrant in the reference and may run
// on several threads at once (clones share the shapes behind Arc<Mutex>, object.rs:9-15), so two scenes must be able to render
// with different seeds / device lists concurrently.  The two settings the device path adds travel PER CALL:
//   * the render seed: the reference draws from fastrand's never-seeded thread-local generator on one thread per row
//     (math/vector.rs:31-38, scene.rs:151) -- every run renders a different image.  `None` keeps that behaviour (a fresh
//     key per call); `Some(k)` makes a render reproducible (it never was).
//   * the GPUs a frame is partitioned over (blocks of rows, one gather); empty: device 0.
// `SceneHipExt` (patches/scene_render.rs implements it for `Scene`) takes them as arguments; the reference's own
// `Scene::render(&self, w, h)` delegates to it with this THREAD's defaults, which `with_render_defaults` scopes.
use std::cell::RefCell;

#[derive(Clone, Debug, Default)]
pub struct RenderOptions {
    pub seed: Option<u64>,
    pub devices: Vec<i32>,
}

impl RenderOptions {
    pub fn seeded(seed: u64) -> Self { RenderOptions { seed: Some(seed), devices: Vec::new() } }
    pub fn on(devices: &[i32]) -> Self { RenderOptions { seed: None, devices: devices.to_vec() } }
    /// The key of THIS call: the fixed seed, or a fresh one (the reference's "a different image every run").
    pub fn key(&self) -> u64 { self.seed.unwrap_or_else(|| fastrand::u64(..)) }
}

thread_local! {
    static DEFAULTS: RefCell<RenderOptions> = RefCell::new(RenderOptions::default());
}

/// What `Scene::render(&self, w, h)` -- the reference's signature, which has no room for options -- uses on this thread.
pub fn render_defaults() -> RenderOptions {
    DEFAULTS.with(|d| d.borrow().clone())
}

/// Runs `f` with `opts` as this thread's defaults and restores the previous ones afterwards (also when `f` panics, as the
/// reference's render does on failure): `hip::with_render_defaults(RenderOptions::seeded(7), || scene.render(w, h))`.
pub fn with_render_defaults<R>(opts: RenderOptions, f: impl FnOnce() -> R) -> R {
    struct Restore(Option<RenderOptions>);
    impl Drop for Restore {
        fn drop(&mut self) {
            if let Some(prev) = self.0.take() { DEFAULTS.with(|d| *d.borrow_mut() = prev); }
        }
    }
    let _restore = Restore(Some(DEFAULTS.with(|d| std::mem::replace(&mut *d.borrow_mut(), opts))));
    f()
}

/// The per-call form of the render entry points (implemented for `Scene` in patches/scene_render.rs).
pub trait SceneHipExt {
    type Pixel;
    /// `Scene::render` with everything the device path adds passed explicitly.
    fn render_with(&self, width: usize, height: usize, opts: &RenderOptions) -> Vec<Vec<Self::Pixel>>;
    /// A reproducible render: the same seed gives the same image on any partition of the frame.
    fn render_seeded(&self, width: usize, height: usize, seed: u64) -> Vec<Vec<Self::Pixel>> {
        self.render_with(width, height, &RenderOptions::seeded(seed))
    }
    /// The frame partitioned over `devices` (blocks of 8 rows round-robin, one gather on `devices[0]`).
    fn render_on(&self, width: usize, height: usize, devices: &[i32]) -> Vec<Vec<Self::Pixel>> {
        self.render_with(width, height, &RenderOptions::on(devices))
    }
}
