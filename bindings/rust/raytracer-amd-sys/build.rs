// Links librt_amd.so (built by `make product` in the raytracer_amd repository: hipcc, gfx950).
// RT_AMD_LIB_DIR = directory that holds librt_amd.so (default: ../../../raytracer_amd/_lib).
fn main() {
    let dir = std::env::var("RT_AMD_LIB_DIR").unwrap_or_else(|_| {
        let here = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{here}/../../../raytracer_amd/_lib")
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=rt_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=RT_AMD_LIB_DIR");
}
