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


def sources() -> list[str]:
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
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
            cmd = [HIPCC, *COMMON_FLAGS, *EXTRA.get(src, []), "-c", spath, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
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
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
