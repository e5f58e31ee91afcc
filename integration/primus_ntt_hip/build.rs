// Points the linker at the in-tree build of libpfhe_hip.so (make -C primus-fhe_amd).
fn main() {
    let dir = std::env::var("PFHE_LIB_DIR").expect("set PFHE_LIB_DIR to the directory holding libpfhe_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=pfhe_hip");
    println!("cargo:rerun-if-env-changed=PFHE_LIB_DIR");
}
