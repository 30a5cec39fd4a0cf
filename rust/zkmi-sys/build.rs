// Builds libzkmi.so with hipcc for gfx950 from the sources in zk_amd/csrc (what north_star calls "compiled with hipcc
// from build.rs").  NOT COMPILED OR RUN in the build image of this repository (no Rust toolchain there): the command
// line is the one zk_amd/csrc/Makefile uses, which IS exercised.
use std::{env, path::PathBuf, process::Command};

fn main() {
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let src = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../zk_amd/csrc");
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".into());
    // the same list as SRCS in zk_amd/csrc/Makefile (tests/test_abi_host.py fails when the two differ)
    let files = ["kernels.hip", "deg2_kernels.hip", "gkr_kernels.hip", "zkmi.cpp", "gkr.cpp", "sharded.cpp"];
    let status = Command::new(&hipcc)
        .args(["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-shared"])
        .arg("-o")
        .arg(out.join("libzkmi.so"))
        .args(files.iter().map(|f| src.join(f)))
        .args(["-ldl", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--exclude-libs,ALL", "-Wl,-z,defs"])
        .status()
        .expect("hipcc not found: set HIPCC");
    assert!(status.success(), "hipcc failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=zkmi");
    println!("cargo:rerun-if-changed={}", src.display());
    println!("cargo:rerun-if-changed=../../include/zkmi.h");
}
