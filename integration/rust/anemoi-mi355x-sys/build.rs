// Points the linker at libanemoi_mi355x.so (built by `make -C anemoi-rust_amd`).
fn main() {
    let dir = std::env::var("ANEMOI_MI355X_LIB_DIR").unwrap_or_else(|_| "/usr/local/lib".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=anemoi_mi355x");
    println!("cargo:rerun-if-env-changed=ANEMOI_MI355X_LIB_DIR");
}
