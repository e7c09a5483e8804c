// Locates libkogarashi_amd.so (built by `python -m kogarashi_amd.build`: hipcc, --offload-arch=gfx950).
//   KOGARASHI_AMD_LIB_DIR   directory holding libkogarashi_amd.so (default: ../../kogarashi_amd next to this crate)
// The library links the HIP runtime itself; a Rust host needs nothing from ROCm at build time.
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=KOGARASHI_AMD_LIB_DIR");
    let dir = env::var_os("KOGARASHI_AMD_LIB_DIR").map(PathBuf::from).unwrap_or_else(|| {
        PathBuf::from(env::var_os("CARGO_MANIFEST_DIR").expect("CARGO_MANIFEST_DIR")).join("../../kogarashi_amd")
    });
    let so = dir.join("libkogarashi_amd.so");
    if !so.exists() {
        panic!("{} not found: run `python -m kogarashi_amd.build` or set KOGARASHI_AMD_LIB_DIR", so.display());
    }
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=kogarashi_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-changed={}", so.display());
}
