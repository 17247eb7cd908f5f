// Links libcontrast_hip.so (built by `python -c "import __graft_entry__ as g; g.build()"`: hipcc --offload-arch=gfx950).
// CONTRAST_HIP_LIB_DIR overrides the directory; the default is the in-tree location of the library.
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("CONTRAST_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../contrast_renderer_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=contrast_hip"); // depends on libamdhip64.so only; librccl.so is opened on first use of Comm
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=CONTRAST_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/contrast_hip.h");
}
