"""Developer microbenchmark: per-shape device GEMV timing for each kernel, with the
weights of a whole synthetic model rotated through so every launch streams from HBM
(521 MB of QK256 codes > 256 MiB Infinity Cache).

    python tools/perf_gemv.py [--layers 30] [--iters 20] [--kernels valu,mfma] [--fmt qk256|i2s32]

Prints one line per (kernel, shape): us/launch, GB/s over algorithmic bytes, % of 8 TB/s.
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bitnet-rs_amd")

SHAPES = {"q/o": (2560, 2560), "k/v": (640, 2560), "gate/up": (6912, 2560), "down": (2560, 6912)}
PEAK = 8.0e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=30)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--kernels", default="valu")
    ap.add_argument("--fmt", default="qk256")
    ap.add_argument("--graph", action="store_true")
    args = ap.parse_args()
    hip = pkg.load()
    hip.init(0)
    kid = {"exact": pkg.KERNEL_EXACT, "valu": pkg.KERNEL_VALU, "mfma": pkg.KERNEL_MFMA, "mfma_tiled": pkg.KERNEL_MFMA_TILED, "auto": pkg.KERNEL_AUTO}
    rng = np.random.default_rng(42)
    for name, (rows, cols) in SHAPES.items():
        handles = []
        for l in range(args.layers):
            if args.fmt == "qk256":
                stride = cols // 256 * 64
                qs = rng.integers(0, 256, rows * stride, dtype=np.uint8)
                handles.append(hip.weights_upload_qk256(qs, rows, cols, stride))
            else:
                bs = 32
                w = rng.integers(0, 256, rows * cols // 4, dtype=np.uint8)
                sc = (1.0 / ((np.arange(rows * cols // bs) % 100) + 1)).astype(np.float32)
                handles.append(hip.weights_upload_i2s(w, sc, rows, cols, bs))
        _, _, abytes = hip.weights_info(handles[0])
        abytes += cols * 4 + rows * 4
        x = torch.randn(cols, device="cuda")
        y = torch.empty(rows, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        for kname in args.kernels.split(","):
            hip.set_kernel(kid[kname])
            for h in handles:
                hip.gemv_dev(h, x, y, stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if args.graph:
                # one graph = one pass over every layer's matrix: no host launch cost
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    cs = torch.cuda.current_stream().cuda_stream
                    for h in handles:
                        hip.gemv_dev(h, x, y, cs)
                gr.replay()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(args.iters):
                    gr.replay()
                e1.record()
                torch.cuda.synchronize()
            else:
                e0.record()
                for _ in range(args.iters):
                    for h in handles:
                        hip.gemv_dev(h, x, y, stream)
                e1.record()
                torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (args.iters * len(handles))
            gbs = abytes / us / 1e3
            print(f"{args.fmt:6s} {kname:6s} {name:8s} {rows:5d}x{cols:5d}  {us:8.2f} us/launch  {gbs:8.1f} GB/s  {100 * gbs * 1e9 / PEAK:5.1f}% of 8TB/s", flush=True)
        for h in handles:
            hip.weights_free(h)
    hip.set_kernel(pkg.KERNEL_AUTO)


if __name__ == "__main__":
    main()
