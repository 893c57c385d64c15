// Links liblurk_hip.so.  LURK_HIP_LIB_DIR names the directory that holds it (the in-tree build puts it in lurk_beta_amd/);
// without the variable the library is expected on the linker's default search path.
use std::env;

fn main() {
    println!("cargo:rerun-if-env-changed=LURK_HIP_LIB_DIR");
    if let Ok(dir) = env::var("LURK_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=lurk_hip");
}
