// Builds librt_hip.so from the HIP sources with hipcc and links it.
// RT_HIP_CSRC points at raytracing_amd/csrc of the raytracing_amd checkout (default: ./csrc next to this file),
// RT_HIP_INCLUDE at its include/ directory (default: ./include).
use std::{env, path::PathBuf, process::Command};

fn main() {
    let here = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let csrc = env::var("RT_HIP_CSRC").map(PathBuf::from).unwrap_or_else(|_| here.join("csrc"));
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".to_string());
    let lib = out.join("librt_hip.so");
    let status = Command::new(&hipcc)
        .args(["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-shared", "-o"])
        .arg(&lib)
        .arg(csrc.join("rt_kernels.hip"))
        .arg(csrc.join("rt_pinhole.hip"))
        .arg(csrc.join("rt_paths.hip"))
        .arg(csrc.join("rt_scratch.hip"))
        .arg(csrc.join("rt_capi.cpp"))
        .status()
        .expect("hipcc not found (set HIPCC)");
    assert!(status.success(), "hipcc failed");
    // every prerequisite of csrc/Makefile's library rule (tests/test_abi.py compares the two lists)
    for f in [
        "rt_kernels.hip", "rt_pinhole.hip", "rt_paths.hip", "rt_scratch.hip", "rt_capi.cpp", "rt_internal.h", "rt_device.h", "rt_complex.h",
        "rt_host_shapes.h", "rt_host_bvh.h", "rt_host_grid.h", "rt_grid.h",
    ] {
        println!("cargo:rerun-if-changed={}", csrc.join(f).display());
    }
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=rt_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", out.display());
    println!("cargo:rustc-link-arg=-Wl,-rpath,/opt/rocm/lib");
}
