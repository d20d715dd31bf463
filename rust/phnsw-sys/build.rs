// Link against libphnsw.so (built by `make -C parallel_hnsw_amd/csrc`, hipcc --offload-arch=gfx950).
// PHNSW_LIB_DIR points at the directory holding libphnsw.so; default: the in-tree build output.
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("PHNSW_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../parallel_hnsw_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=phnsw");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=PHNSW_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/phnsw.h");
}
