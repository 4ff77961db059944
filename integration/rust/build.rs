// build.rs for the vectorlite crate once the GPU index is wired in (see ../README.md).
// VECTORLITE_AMD_LIB_DIR = directory holding libvectorlite_amd.so (this repository's vectorlite_amd/).
fn main() {
    if let Ok(dir) = std::env::var("VECTORLITE_AMD_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=vectorlite_amd");
    println!("cargo:rerun-if-env-changed=VECTORLITE_AMD_LIB_DIR");
}
