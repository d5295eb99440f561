#!/usr/bin/env python3
"""Per-kernel resources of one HIP source as hipcc reports them (-Rpass-analysis=kernel-resource-usage):
    python tools/kres.py kernels_gemm.hip [name-filter] [-- extra hipcc flags]
prints name, VGPRs, AGPRs, scratch bytes per lane, occupancy (waves per SIMD), static LDS bytes.  CPU only (cross-compiles gfx950)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--")
        args, extra = args[:i], args[i + 1:]
    src = args[0] if os.path.exists(args[0]) else os.path.join(ROOT, "bitnet-rs_amd", "csrc", args[0])
    flt = args[1] if len(args) > 1 else ""
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{ROOT}/bitnet-rs_amd/csrc", "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = []
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("sgpr", r" SGPRs: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    for r, n in zip(rows, names):
        n = n.replace("bitnet_hip::", "")
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        if flt and flt not in n:
            continue
        print(f"{n:60s} vgpr {r.get('vgpr', '?'):>4} agpr {r.get('agpr', '?'):>4} scratch {r.get('scratch', '?'):>4} occ {r.get('occ', '?'):>2} lds {r.get('lds', '?'):>6}")
    if not rows:
        sys.stderr.write(out[-3000:])
        sys.exit(1)


if __name__ == "__main__":
    main()
