// Links librthip.so.  RTHIP_LIB_DIR names the directory that holds it (raytracer_amd/ of the
// librthip source tree after `make -C raytracer_amd/csrc`).
fn main() {
    let dir = std::env::var("RTHIP_LIB_DIR").expect("set RTHIP_LIB_DIR to the directory holding librthip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=rthip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=RTHIP_LIB_DIR");
}
