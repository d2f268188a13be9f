// The drop-in library (libsliceslice_hip.so: include/sliceslice_hip.h) - and, with the `hip-service` feature, the library that
// holds the resident search service as well (libsliceslice_hip_service.so: the same objects plus ss_service.hip), linked INSTEAD.
fn main() {
    let out = std::env::var("OUT_DIR").unwrap();
    let service = std::env::var("CARGO_FEATURE_HIP_SERVICE").is_ok();
    let name = if service { "sliceslice_hip_service" } else { "sliceslice_hip" };
    let mut units = vec!["ss_core.hip", "ss_scan.hip", "ss_census.hip", "ss_host.hip", "ss_batched.hip", "ss_comm.hip", "scan_inst_u4_nt1.hip", "scan_inst_find_nt1.hip"];
    if service { units.push("ss_service.hip"); }
    let ok = std::process::Command::new("hipcc")
        .args(["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", "-Wl,-Bsymbolic-functions", "-o"])
        .arg(format!("{out}/lib{name}.so"))
        .args(units.iter().map(|f| format!("sliceslice-rs_amd/csrc/{f}")))
        .arg("-ldl")
        .status().unwrap().success();
    assert!(ok);
    println!("cargo:rustc-link-search=native={out}");
    println!("cargo:rustc-link-lib=dylib={name}");
}
