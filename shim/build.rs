// Links librecgraph_hip.so (built by `make -C recgraph_amd/csrc`): set RECGRAPH_HIP_LIB_DIR to the directory holding it.
fn main() {
    let dir = std::env::var("RECGRAPH_HIP_LIB_DIR").unwrap_or_else(|_| String::from("recgraph_amd"));
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=recgraph_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=RECGRAPH_HIP_LIB_DIR");
}
