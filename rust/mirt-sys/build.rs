// Link against libmirt.so built by `make -C weekend-raytracer-wgpu_amd/csrc`.
// MIRT_LIB_DIR must name the directory that holds it.
fn main() {
    let dir = std::env::var("MIRT_LIB_DIR").expect("set MIRT_LIB_DIR to the directory holding libmirt.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=mirt");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=MIRT_LIB_DIR");
}
