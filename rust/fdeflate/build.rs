// Links libfdeflate_hip.so (built by `make -C fdeflate_amd/csrc`).  FDEFLATE_HIP_LIB_DIR overrides
// the in-tree location.
fn main() {
    let dir = std::env::var("FDEFLATE_HIP_LIB_DIR").unwrap_or_else(|_| {
        let here = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{here}/../../fdeflate_amd")
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=fdeflate_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=FDEFLATE_HIP_LIB_DIR");
}
