fn main() {
    let out = std::env::var("OUT_DIR").unwrap();
    let ok = std::process::Command::new("hipcc")
        .args(["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o"])
        .arg(format!("{out}/libsliceslice_hip.so"))
        .args(["sliceslice_hip.hip", "scan_inst_u4_nt0.hip", "scan_inst_u4_nt1.hip", "scan_inst_u8_nt0.hip", "scan_inst_u8_nt1.hip", "scan_inst_find_nt0.hip", "scan_inst_find_nt1.hip"]
              .map(|f| format!("sliceslice-rs_amd/csrc/{f}")))
        .arg("-ldl")
        .status().unwrap().success();
    assert!(ok);
    println!("cargo:rustc-link-search=native={out}");
    println!("cargo:rustc-link-lib=dylib=sliceslice_hip");
}
