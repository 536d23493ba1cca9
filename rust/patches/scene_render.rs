// patches/scene_render.rs -- new bodies of Scene::render / Scene::render_to_image (src/raytracing/scene.rs:144-178).
// UNCOMPILED.  One FFI call each; everything below them (render_pixel .. random_bounce_dir, scene.rs:194-292) runs in
// hand-written HIP on the MI355X.  Signatures, [y][x] orientation and the panic-on-failure behaviour are the reference's.
use crate::raytracing::hip::{self, RenderOptions, SceneHipExt};
use crate::raytracing::object::Primitive;

// The per-call form: seed and device list are arguments, nothing process-wide (hip.rs).  `scene.render_seeded(w, h, 7)`,
// `scene.render_on(w, h, &[0, 1, 2, 3])` come with the trait.
impl SceneHipExt for Scene {
    type Pixel = Vector3;
    fn render_with(&self, width: usize, height: usize, opts: &RenderOptions) -> Vec<Vec<Vector3>> {
        let packed = self.pack(); // scene order is kept: it decides ties (scene.rs:250)
        let sc = self.to_c(&packed, opts.key());
        let mut flat = vec![0f64; width * height * 3];
        let devices = &opts.devices;
        let rc = unsafe {
            if devices.is_empty() {
                hip::rtx_render(&sc, width as u32, height as u32, flat.as_mut_ptr())
            } else {
                hip::rtx_render_devices(&sc, width as u32, height as u32, devices.as_ptr(), devices.len() as u32, flat.as_mut_ptr())
            }
        };
        hip::check(rc); // the reference panics on failure (scene.rs:168)
        flat.chunks(width * 3)
            .map(|row| row.chunks(3).map(|c| Vector3::new(c[0], c[1], c[2])).collect())
            .collect()
    }
}

impl Scene {
    /// The reference's signature (scene.rs:144): this thread's defaults (hip::with_render_defaults), else a fresh key on device 0.
    pub fn render(&self, width: usize, height: usize) -> Vec<Vec<Vector3>> {
        self.render_with(width, height, &hip::render_defaults())
    }

    #[cfg(feature = "images")]
    pub fn render_to_image(&self, width: usize, height: usize) -> ImageBuffer<Rgb<u8>, Vec<u8>> {
        self.render_to_image_with(width, height, &hip::render_defaults())
    }

    #[cfg(feature = "images")]
    pub fn render_to_image_with(&self, width: usize, height: usize, opts: &RenderOptions) -> ImageBuffer<Rgb<u8>, Vec<u8>> {
        let packed = self.pack();
        let sc = self.to_c(&packed, opts.key());
        let mut buf = vec![0u8; width * height * 3];
        let devices = &opts.devices;
        let rc = unsafe {
            if devices.is_empty() {
                hip::rtx_render_to_image(&sc, width as u32, height as u32, buf.as_mut_ptr())
            } else {
                hip::rtx_render_to_image_devices(&sc, width as u32, height as u32, devices.as_ptr(), devices.len() as u32, buf.as_mut_ptr())
            }
        };
        hip::check(rc);
        ImageBuffer::from_raw(width as u32, height as u32, buf).unwrap() // already `* 256`, `as u8`, flipped (scene.rs:175-178)
    }

    fn pack(&self) -> Vec<hip::RtxObject> {
        self.objects.iter().enumerate().map(|(i, o)| {
            let prim = o.primitive().unwrap_or_else(|| {
                panic!("object {i}: a CustomShape without primitive() cannot run on the GPU (the device library has no CPU fallback)")
            });
            let (kind, geom) = match prim {
                Primitive::Sphere { position: p, radius } => (hip::RTX_SPHERE, [p.x, p.y, p.z, radius, 0., 0., 0., 0., 0.]),
                Primitive::Plane { position: p, normal: n } => (hip::RTX_PLANE, [p.x, p.y, p.z, n.x, n.y, n.z, 0., 0., 0.]),
                Primitive::Triangle { vertices: v } => (hip::RTX_TRIANGLE,
                    [v[0].x, v[0].y, v[0].z, v[1].x, v[1].y, v[1].z, v[2].x, v[2].y, v[2].z]),
            };
            let m = &o.material;
            hip::RtxObject {
                kind, reserved: 0, geom,
                base_color: [m.base_color.x, m.base_color.y, m.base_color.z],
                emission_color: [m.emission_color.x, m.emission_color.y, m.emission_color.z],
                roughness: m.roughness,
            }
        }).collect()
    }

    fn to_c(&self, packed: &[hip::RtxObject], seed: u64) -> hip::RtxScene {
        let c = &self.config;
        let cam = &self.camera;
        let rows = |m: &Mat3x3| [m.x.x, m.x.y, m.x.z, m.y.x, m.y.y, m.y.z, m.z.x, m.z.y, m.z.z]; // mat.rs:11-18: three row vectors
        hip::RtxScene {
            config: hip::RtxConfig {
                rays_per_pixel: c.rays_per_pixel as u64, max_bounces: c.max_bounces as u64,
                focal_length: c.focal_length, focal_offset: c.focal_offset, non_focal_offset: c.non_focal_offset,
                seed, kernel: hip::RTX_KERNEL_AUTO, tuning: 0,
            },
            camera: hip::RtxCamera {
                fov: cam.fov, position: cam.position.into(), direction: cam.get_direction().into(),
                to_cam_space: rows(&cam.to_cam_space), to_world_space: rows(&cam.to_world_space), // same crate: pub(crate) access added
            },
            n_objects: packed.len() as u64,
            objects: packed.as_ptr(),
        }
    }
}
