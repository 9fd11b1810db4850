// Link against the prebuilt libmirt.so; MIRT_LIB_DIR points at raytracer_amd/csrc of this repository.
fn main() {
    let dir = std::env::var("MIRT_LIB_DIR").unwrap_or_else(|_| "../../../raytracer_amd/csrc".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=mirt");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=MIRT_LIB_DIR");
}
