"""Builds libbitnet_hip.so (gfx950 code objects only) in-tree with hipcc.

    python bitnet-rs_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The .so stays next to this file so it travels
with the source snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB_PATH = os.path.join(HERE, "libbitnet_hip.so")
OBJ_DIR = os.path.join(HERE, "build")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

COMMON_FLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-Wall",
    "-Wno-unused-result",
    f"-I{INCLUDE}",
    f"-I{CSRC}",
]
# Per-file extra flags.  kernels_exact.hip restates the reference's scalar loops
# bit for bit, and Rust never contracts a*b+c.
EXTRA = {"kernels_exact.hip": ["-ffp-contract=off"]}


# The hand-counted `s_waitcnt vmcnt(N)` around the LDS-DMA of k_prefill_attn (and the register / scratch figures DESIGN.md
# quotes) were validated on this toolchain: tests/test_isa_guard.py re-checks them from the emitted ISA.
VALIDATED_HIPCC = "HIP version: 7.2.26015-fc0010cf6a"


def hipcc_version() -> str:
    try:
        out = subprocess.run([HIPCC, "--version"], capture_output=True, text=True, timeout=60).stdout
    except (OSError, subprocess.SubprocessError):
        return ""
    return out.splitlines()[0].strip() if out else ""


def warn_if_unvalidated_hipcc() -> None:
    v = hipcc_version()
    if v != VALIDATED_HIPCC:
        sys.stderr.write(f"warning: hipcc is '{v}', the hand-counted vmcnt waits of kernels_prefill_attn.hip were validated on "
                         f"'{VALIDATED_HIPCC}': run `pytest tests/test_isa_guard.py` before trusting this build\n")


def sources() -> list[str]:
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, diag: bool = False) -> str:
    """diag=True builds libbitnet_hip_diag.so with -DBH_STAMPS (in-kernel time stamps);
    it is a developer tool and never what tests, smoke() or bench.py load."""
    global OBJ_DIR, LIB_PATH
    obj_dir = OBJ_DIR + ("_diag" if diag else "")
    lib_path = LIB_PATH.replace(".so", "_diag.so") if diag else LIB_PATH
    extra_all = ["-DBH_STAMPS"] if diag else []
    ig = os.environ.get("BH_IGLP")  # developer builds of kernels_gemm.hip with an IGroupLP strategy hint in the f16 matmul's K loop
    if ig:
        return _build(force, verbose, OBJ_DIR + "_iglp" + ig, LIB_PATH.replace(".so", f"_ablateiglp{ig}.so"), [f"-DBH_IGLP={ig}"])
    sf = os.environ.get("BH_GEMM_SCHED")  # developer builds of kernels_gemm.hip under another LLVM scheduling strategy (max-ilp, ...)
    if sf:
        EXTRA["kernels_gemm.hip"] = ["-mllvm", f"-amdgpu-sched-strategy={sf}"]
        try:
            return _build(force, verbose, OBJ_DIR + "_sched_" + sf, LIB_PATH.replace(".so", f"_ablatesched{sf.replace('-', '')}.so"), ["-DBH_SCHED_VARIANT"])
        finally:
            EXTRA.pop("kernels_gemm.hip", None)
    ab = os.environ.get("BH_ABLATE")  # developer builds of kernels_gemm.hip with parts of the loop removed (tools/ablate_gemm.py)
    if ab:
        obj_dir, lib_path, extra_all = OBJ_DIR + "_ablate" + ab, LIB_PATH.replace(".so", f"_ablate{ab}.so"), [f"-DBH_ABLATE={ab}", "-DBH_STAMPS"]
    return _build(force, verbose, obj_dir, lib_path, extra_all)


def _build(force: bool, verbose: bool, OBJ_DIR: str, LIB_PATH: str, extra_all: list) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    headers += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    headers.append(os.path.abspath(__file__))
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        objs.append(obj)
        spath = os.path.join(CSRC, src)
        if force or _stale(obj, [spath] + headers):
            cmd = [HIPCC, *COMMON_FLAGS, *extra_all, *EXTRA.get(src, []), "-c", spath, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    if procs:
        warn_if_unvalidated_hipcc()
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- hipcc failed on {src} ---\n{out}\n")
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc compilation failed")
    if force or procs or _stale(LIB_PATH, objs):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    if not extra_all:
        _build_host(force, verbose, LIB_PATH)
    return LIB_PATH


HOST_DIR = os.path.join(HERE, "host")
HOST_LIB_PATH = os.path.join(HERE, "libbitnet_host.so")


def _build_host(force: bool, verbose: bool, kernel_lib: str) -> str:
    """The C++ host layer above the C ABI (decode loop, graph capture).  Plain host C++:
    it only includes include/bitnet_hip.h and the HIP runtime API."""
    srcs = [os.path.join(HOST_DIR, f) for f in sorted(os.listdir(HOST_DIR)) if f.endswith(".cpp")]
    deps = srcs + [os.path.join(HOST_DIR, f) for f in os.listdir(HOST_DIR) if f.endswith(".hpp")]
    deps += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)] + [kernel_lib]
    if force or _stale(HOST_LIB_PATH, deps):
        cmd = [
            "g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-result", "-Wno-int-in-bool-context", "-D__HIP_PLATFORM_AMD__",
            f"-I{INCLUDE}", f"-I{HOST_DIR}", "-I/opt/rocm/include", *srcs, "-o", HOST_LIB_PATH,
            f"-L{HERE}", "-lbitnet_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib",
        ]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return HOST_LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv))
