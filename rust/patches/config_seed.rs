// patches/config_seed.rs -- additions to `Config` (src/raytracing/scene.rs:16-65).  UNCOMPILED.
// The reference's randomness is fastrand's unseedable thread-local generator on one thread per row
// (math/vector.rs:31-38, scene.rs:151); the device path needs an explicit key.  Default: a fresh key per Config, which
// keeps "a different image every run"; `with_seed` makes a render reproducible (it never was).

// new fields of `pub struct Config` (scene.rs:16-28):
//     pub seed: u64,
//     /// GPUs the frame is partitioned over (interleaved row bands, one gather); empty = device 0.
//     pub devices: Vec<i32>,

impl Config {
    pub fn with_seed(self, seed: u64) -> Self {
        Self { seed, ..self }
    }
    pub fn with_devices(self, devices: Vec<i32>) -> Self {
        Self { devices, ..self }
    }
}

// in `impl Default for Config` (scene.rs:55-65):
//     seed: fastrand::u64(..),
//     devices: Vec::new(),
