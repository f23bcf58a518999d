// Points the linker at the in-tree libh2w.so (built by ./build.sh with hipcc for gfx950); override with H2W_LIB_DIR.
fn main() {
    let dir = std::env::var("H2W_LIB_DIR").unwrap_or_else(|_| {
        format!("{}/../../halo2-plonky2-verifier_amd", std::env::var("CARGO_MANIFEST_DIR").unwrap())
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=h2w");
    println!("cargo:rerun-if-env-changed=H2W_LIB_DIR");
}
