fn main() {
    let out = std::env::var("OUT_DIR").unwrap();
    let ok = std::process::Command::new("hipcc")
        .args(["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", "-Wl,-Bsymbolic-functions", "-o"])
        .arg(format!("{out}/libsliceslice_hip.so"))
        .args(["ss_core.hip", "ss_scan.hip", "ss_census.hip", "ss_host.hip", "ss_batched.hip", "ss_service.hip", "ss_comm.hip", "scan_inst_u4_nt1.hip", "scan_inst_find_nt1.hip"]
              .map(|f| format!("sliceslice-rs_amd/csrc/{f}")))
        .arg("-ldl")
        .status().unwrap().success();
    assert!(ok);
    println!("cargo:rustc-link-search=native={out}");
    println!("cargo:rustc-link-lib=dylib=sliceslice_hip");
}
