// Points rustc at libraytracer_amd.so (built by `make -C raytracer_amd`).
fn main() {
    let dir = std::env::var("RAYTRACER_AMD_LIB_DIR").unwrap_or_else(|_| "../../raytracer_amd".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=raytracer_amd");
    println!("cargo:rerun-if-env-changed=RAYTRACER_AMD_LIB_DIR");
}
