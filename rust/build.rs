// build.rs -- links the crate against librtx_hip.so (built by `python rust-raytracing_amd/build.py`).
// UNCOMPILED here: the image that produced this file has no Rust toolchain.
fn main() {
    let dir = std::env::var("RTX_HIP_LIB_DIR").unwrap_or_else(|_| "/usr/local/lib".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=rtx_hip");
    println!("cargo:rerun-if-env-changed=RTX_HIP_LIB_DIR");
}
