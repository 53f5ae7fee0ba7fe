// UNVERIFIED (no Rust toolchain in the build image).
// RT_AMD_LIB_DIR = directory that holds librt_amd.so (raytracer_amd/ of this repository).
fn main() {
    if let Ok(dir) = std::env::var("RT_AMD_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=rt_amd");
    println!("cargo:rerun-if-env-changed=RT_AMD_LIB_DIR");
}
