// Link libbvh_mi355x.so (built by `python bvh_amd/build_ext.py`; needs libamdhip64.so.7 at run time; librccl.so.1 is dlopen'ed by the library only when a communicator is created).
fn main() {
    let dir = std::env::var("BVH_MI355X_LIB_DIR")
        .expect("set BVH_MI355X_LIB_DIR to the directory that holds libbvh_mi355x.so (<repo>/bvh_amd)");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=bvh_mi355x");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=BVH_MI355X_LIB_DIR");
}
