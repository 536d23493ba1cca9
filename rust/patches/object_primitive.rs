// patches/object_primitive.rs -- additions to src/raytracing/object.rs (trait at :53-76) and the three shapes.
// UNCOMPILED.  `Object.shape` is a type-erased Arc<Mutex<dyn CustomShape>> whose trait has only distance()/normal()
// (object.rs:9-15,53-76): geometry cannot be recovered from an existing Object.  One PROVIDED method fixes that
// without breaking user impls.

/// What the device can run for a shape.
#[derive(Clone, Copy, Debug)]
pub enum Primitive {
    Sphere { position: Vector3, radius: f64 },
    Plane { position: Vector3, normal: Vector3 },
    Triangle { vertices: [Vector3; 3] },
}

// in `pub trait CustomShape` (object.rs:53-76), after `fn normal(..)`:
//     /// Device primitive of this shape; `None` (the default) means "cannot run on the GPU".
//     fn primitive(&self) -> Option<Primitive> { None }

// in `impl CustomShape for Sphere` (object/sphere.rs:18-34):
//     fn primitive(&self) -> Option<Primitive> { Some(Primitive::Sphere { position: self.position, radius: self.radius }) }
// in `impl CustomShape for Plane` (object/plane.rs:18-36):
//     fn primitive(&self) -> Option<Primitive> { Some(Primitive::Plane { position: self.position, normal: self.normal }) }
// in `impl CustomShape for Triangle` (object/triangle.rs:103-128):
//     fn primitive(&self) -> Option<Primitive> { Some(Primitive::Triangle { vertices: self.vertices }) }

impl Object {
    /// Locks the shape like distance()/normal_at() do (object.rs:38,50) and asks it for its device primitive.
    pub(crate) fn primitive(&self) -> Option<Primitive> {
        self.shape.lock().unwrap().primitive()
    }
}

// src/raytracing/camera.rs:12-14: `to_cam_space` / `to_world_space` become `pub(crate)` so that scene.rs (a sibling module)
// can hand the three rows of each matrix to the device library (patches/scene_render.rs, to_c()).
