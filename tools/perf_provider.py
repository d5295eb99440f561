"""Developer / profile tool: the KernelProvider trait ops and the a8 composite at the model's
shapes, through the host-pointer drop-ins.  Run under rocprofv3 --kernel-trace (tools/profile_round.sh) to get per-kernel
device times; prints the host-side wall times (which include H2D / D2H, like the reference's GPU provider per call).

    python tools/perf_provider.py
"""
from __future__ import annotations

import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bitnet-rs_amd")


def main():
    hip = pkg.load()
    hip.init(0)
    rng = np.random.default_rng(0)
    rows = []
    for m, n, k in ((1, 2560, 2560), (1, 6912, 2560), (8, 2560, 6912)):
        a = rng.integers(-2, 2, m * k).astype(np.int8)
        b = rng.integers(0, 4, k * n).astype(np.uint8)
        for kern, name in ((pkg.KERNEL_AUTO, "tiled"), (pkg.KERNEL_EXACT, "reference-order")):
            hip.set_kernel(kern)
            hip.matmul_i2s(a, b, m, n, k)
            t0 = time.perf_counter()
            for _ in range(3):
                hip.matmul_i2s(a, b, m, n, k)
            rows.append((f"matmul_i2s {name} m={m} n={n} k={k}", (time.perf_counter() - t0) / 3 * 1e3))
    hip.set_kernel(pkg.KERNEL_AUTO)
    x = rng.normal(0, 1, 6912 * 2560).astype(np.float32)
    for kern, name in ((pkg.KERNEL_AUTO, "coalesced"), (pkg.KERNEL_EXACT, "reference-order")):
        hip.set_kernel(kern)
        hip.quantize(x)
        t0 = time.perf_counter()
        for _ in range(3):
            hip.quantize(x)
        rows.append((f"quantize I2S {name} n={x.size}", (time.perf_counter() - t0) / 3 * 1e3))
    hip.set_kernel(pkg.KERNEL_AUTO)
    r, c = 2560, 2560
    blocks = rng.integers(0, 256, r * (c // 32) * 10, dtype=np.uint8)
    hip.dequant_i2s(blocks, r, c)
    t0 = time.perf_counter()
    for _ in range(3):
        hip.dequant_i2s(blocks, r, c)
    rows.append((f"dequant_i2s rows={r} cols={c} block=32", (time.perf_counter() - t0) / 3 * 1e3))
    xi = rng.normal(0, 1, 4 * 2560).astype(np.float32)
    packed = rng.integers(0, 256, 2560 * 6912 // 4, dtype=np.uint8)
    sc = rng.uniform(0.5, 1.5, 6912).astype(np.float32)
    hip.quantized_matmul_i2s(xi, packed, sc, 32, 4, 6912, 2560)
    t0 = time.perf_counter()
    for _ in range(3):
        hip.quantized_matmul_i2s(xi, packed, sc, 32, 4, 6912, 2560)
    rows.append(("quantized_matmul_i2s composite m=4 n=6912 k=2560", (time.perf_counter() - t0) / 3 * 1e3))
    for name, ms in rows:
        print(f"{name:60s} {ms:9.3f} ms per call (host wall, incl. copies)")


if __name__ == "__main__":
    main()
