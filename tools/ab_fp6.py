#!/usr/bin/env python3
"""A/B of the 2-digit prefill matmul's two forms on unscaled (QK256) matrices, one process:
   int8 base-256 digit planes (BITNET_HIP_FUSE_INT8_DIGITS) vs three base-32 fp6 digits on the block-scaled fp6 x fp4 MFMA, the latter expanding
   the 2-bit tiles in its K loop (BITNET_HIP_FUSE_FP6_EXPAND, round 4's form) and reading the resident fp4 image (round 5).
    python tools/ab_fp6.py [--m 4096] [--reps 20]
Prints per 2B-4T shape: bit-equality of the two results, max |diff| otherwise, microseconds per launch pair (quantiser + matmul)."""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bitnet-rs_amd")
INT8, FP6, EXPAND = 8, 16, 32  # BITNET_HIP_FUSE_INT8_DIGITS, BITNET_HIP_FUSE_FP6_DIGITS, BITNET_HIP_FUSE_FP6_EXPAND


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, nargs="+", default=[4096])
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--shapes", nargs="+", default=["qkv", "o", "gate_up", "down"])
    args = ap.parse_args()
    hip = pkg.load()
    hip.init(0)
    rng = np.random.default_rng(0)
    shapes = {"qkv": (3840, 2560), "o": (2560, 2560), "gate_up": (13824, 2560), "down": (2560, 6912), "odd": (1000, 1100)}
    for name in args.shapes:
        n, k = shapes[name]
        stride = (k + 255) // 256 * 64
        qs = rng.integers(0, 256, n * stride, dtype=np.uint8)
        h = hip.weights_upload_qk256(qs, n, k, stride)
        for m in args.m:
            x = torch.randn(m, k, device="cuda") * torch.exp(torch.randn(m, 1, device="cuda"))
            gamma = torch.rand(k, device="cuda") + 0.5
            wsb = hip.matmul_workspace_bytes(m, k, 2)
            ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
            res = {}
            for form, fl in (("int8", INT8), ("fp6x", FP6 | EXPAND), ("fp6", FP6)):
                y = torch.full((m, n), float("nan"), device="cuda")
                hip.matmul_fused_dev(h, x, y, m, ws, wsb, digits=2, flags=fl)
                torch.cuda.synchronize()
                tile = hip.matmul_last_tile()
                yl = torch.full((m, n), float("nan"), device="cuda")
                hip.matmul_fused_dev(h, x, yl, m, ws, wsb, ln_gamma=gamma, ln_eps=1e-5, digits=2, flags=fl)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    hip.matmul_fused_dev(h, x, y, m, ws, wsb, digits=2, flags=fl, stream=torch.cuda.current_stream().cuda_stream)
                e1.record()
                torch.cuda.synchronize()
                res[form] = (y, yl, e0.elapsed_time(e1) / args.reps * 1e3, tile)
            (y8, yl8, t8, tile8), (y6, yl6, t6, tile6) = res["int8"], res["fp6"]
            yx, ylx, tx, _ = res["fp6x"]
            assert torch.equal(yx, y6) and torch.equal(ylx, yl6), "resident and expanding fp6 forms differ"
            eq, eql = bool(torch.equal(y8, y6)), bool(torch.equal(yl8, yl6))
            d = float((y8.double() - y6.double()).abs().max() / y8.abs().max()) if not eq else 0.0
            dl = float((yl8.double() - yl6.double()).abs().max() / yl8.abs().max()) if not eql else 0.0
            print(f"{name:8s} n={n:6d} k={k:5d} m={m:5d}: int8 {t8:8.1f} us {tile8}  fp6-expand {tx:8.1f} us  fp6-resident {t6:8.1f} us {tile6} resident={hip.matmul_last_resident_fp4()}  ratio {t6 / t8:.3f}  "
                  f"bit-equal {eq} (maxrel {d:.2e})  with LN {eql} (maxrel {dl:.2e})  nan {bool(torch.isnan(y6).any())}", flush=True)
        hip.weights_free(h)


if __name__ == "__main__":
    main()
