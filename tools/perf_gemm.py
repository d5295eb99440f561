#!/usr/bin/env python3
"""Times the prefill matmul (kernels_gemm.hip) on the bitnet-b1.58-2B-4T layer shapes.
    python tools/perf_gemm.py [--m 4096] [--reps 10]
Prints ms per launch pair (quantise + GEMM), effective TFLOP/s (2*m*n*k) and int8 TOP/s (x digits)."""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bitnet-rs_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--digits", type=int, nargs="+", default=[4, 3, 2])
    ap.add_argument("--shapes", nargs="+", default=["qkv", "o", "gate_up", "down"])
    ap.add_argument("--data", default="random", choices=["random", "zero_x", "const_w", "zeros"], help="operand data (DVFS check: the chip holds a higher clock on trivial operands)")
    ap.add_argument("--fmt", default="qk256", choices=["qk256", "i2s"], help="storage format: QK256 (no scales) or BitNet32-F16 (ternary, f16 scale per 32)")
    ap.add_argument("--check", action="store_true", help="compare each result with the 4-digit one (cosine, max abs / max)")
    args = ap.parse_args()
    hip = pkg.load()
    hip.init(0)
    rng = np.random.default_rng(0)
    shapes = {"qkv": (3840, 2560), "o": (2560, 2560), "gate_up": (13824, 2560), "down": (2560, 6912)}
    m = args.m
    for name, (n, k) in shapes.items():
        if name not in args.shapes:
            continue
        stride = k // 256 * 64
        qs = rng.integers(0, 256, n * stride, dtype=np.uint8)
        if args.data in ("const_w", "zeros"):
            qs[:] = 0xAA
        if args.fmt == "i2s":
            synth = importlib.import_module("bitnet-rs_amd.synth")
            wq, ws_ = synth.ternary_weights(n, k, 32, 42, 0, 1)
            h = hip.weights_upload_i2s(wq, ws_, n, k, 32)
        else:
            h = hip.weights_upload_qk256(qs, n, k, stride)
        x = torch.randn(m, k, device="cuda")
        if args.data in ("zero_x", "zeros"):
            x.zero_()
        y = torch.empty(m, n, device="cuda")
        y4 = None
        if args.check:
            wsb = hip.matmul_workspace_bytes(m, k, 4)
            ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
            y4 = torch.empty(m, n, device="cuda")
            hip.matmul_fused_dev(h, x, y4, m, ws, wsb, digits=4)
            torch.cuda.synchronize()
        for digits in args.digits:
            wsb = hip.matmul_workspace_bytes(m, k, digits)
            ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
            hip.matmul_fused_dev(h, x, y, m, ws, wsb, digits=digits)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                hip.matmul_fused_dev(h, x, y, m, ws, wsb, digits=digits, stream=torch.cuda.current_stream().cuda_stream)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
            fl = 2.0 * m * n * k
            chk = ""
            if y4 is not None:
                d = (y.double() - y4.double())
                cos = float((y.double() * y4.double()).sum() / (y.double().norm() * y4.double().norm()))
                chk = f"  vs4: cos {cos:.8f} maxrel {float(d.abs().max() / y4.abs().max()):.2e}"
            print(f"{name:8s} n={n:6d} k={k:5d} m={m} digits={digits}: {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s eff  {fl * digits / ms / 1e9:8.1f} int8 TOP/s  tile {hip.matmul_last_tile()}{chk}", flush=True)
        hip.weights_free(h)


if __name__ == "__main__":
    main()
