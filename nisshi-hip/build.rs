// Links libnisshi_hip.so (built from this repository's tansu_amd/csrc by
// `python -m tansu_amd.build`: hipcc --offload-arch=gfx950).
// NISSHI_HIP_LIB_DIR names the directory that holds it.
use std::{env, path::PathBuf};

fn main() {
    println!("cargo:rerun-if-env-changed=NISSHI_HIP_LIB_DIR");
    let dir = env::var_os("NISSHI_HIP_LIB_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|| PathBuf::from("/usr/local/lib"));
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=nisshi_hip");
    // the HIP runtime the library itself links
    if let Some(rocm) = env::var_os("ROCM_PATH") {
        println!(
            "cargo:rustc-link-search=native={}",
            PathBuf::from(rocm).join("lib").display()
        );
    }
}
