"""Developer tool: where does one GEMV launch spend its time?  Uses the diagnostic
library (python bitnet-rs_amd/build.py --diag) whose MFMA kernel writes s_memrealtime
stamps per workgroup:  0 start | 1 loads issued | 2 absmax known | 3 planes in LDS |
4 MFMA loop done | 5 end.   Never used by tests / bench.

    python tools/stamp_gemv.py [--shape gate/up] [--kernel mfma|mfma_tiled]
"""
from __future__ import annotations

import argparse
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bitnet-rs_amd")
SHAPES = {"q/o": (2560, 2560), "k/v": (640, 2560), "gate/up": (6912, 2560), "down": (2560, 6912)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="gate/up")
    ap.add_argument("--kernel", default="mfma")
    ap.add_argument("--layers", type=int, default=30)
    ap.add_argument("--warm-ms", type=float, default=0)
    ap.add_argument("--replays", type=int, default=3)
    ap.add_argument("--fused", action="store_true")
    ap.add_argument("--fmt", default="qk256", choices=["qk256", "i2s32"], help="i2s32: ternary codes + f16-exact 32-block scales (BASELINE configs[1])")
    args = ap.parse_args()
    hip = pkg.HipLib(pkg.LIB_PATH.replace(".so", "_diag.so"))
    hip.init(0)
    hip.c.bitnet_hip_debug_set_stamps.argtypes = [C.c_void_p]
    rows, cols = SHAPES[args.shape]
    stride = cols // 256 * 64
    rng = np.random.default_rng(0)
    def mk():
        if args.fmt == "i2s32":
            sc = (2.0 / ((np.arange(rows * (cols // 32)) % 100) + 1)).astype(np.float16).astype(np.float32)
            return hip.weights_upload_i2s(rng.integers(0, 256, rows * cols // 4, dtype=np.uint8), sc, rows, cols, 32)
        return hip.weights_upload_qk256(rng.integers(0, 256, rows * stride, dtype=np.uint8), rows, cols, stride)

    if args.fused:  # the decode step's LayerNorm -> gate|up -> silu*mul launch
        handles = []
        for _ in range(args.layers):
            a, b = mk(), mk()
            handles.append(hip.weights_concat([a, b], interleave16=True))
            hip.weights_free(a)
            hip.weights_free(b)
    else:
        handles = [mk() for _ in range(args.layers)]
    gamma = torch.full((cols,), 0.0125, device="cuda")
    if args.fused:  # as the decoder does: LayerNorm applied after the product
        for h in handles:
            hip.weights_bind_ln(h, gamma)
    hip.set_kernel({"mfma": pkg.KERNEL_MFMA, "mfma_tiled": pkg.KERNEL_MFMA_TILED}[args.kernel])
    x = torch.randn(cols, device="cuda")
    y = torch.empty(2 * rows, device="cuda")
    stamps = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
    cs = torch.cuda.current_stream().cuda_stream
    def launch(h, st):
        if args.fused:
            hip.gemv_fused_dev(h, x, y, 1, ln_gamma=gamma, ln_eps=1e-5, flags=1, stream=st)
        else:
            hip.gemv_dev(h, x, y, st)

    for h in handles:  # warm
        launch(h, cs)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        s = torch.cuda.current_stream().cuda_stream
        for i, h in enumerate(handles):
            hip.c.bitnet_hip_debug_set_stamps(C.c_void_p(stamps.data_ptr() if i == len(handles) - 1 else 0))
            launch(h, s)
    if args.warm_ms > 0:  # hold the GPU busy so DPM raises the shader clock
        a = torch.randn(8192, 8192, device="cuda")
        t_end = __import__("time").time() + args.warm_ms / 1e3
        while __import__("time").time() < t_end:
            a = (a @ a).clamp_(-1, 1)
        torch.cuda.synchronize()
    for _ in range(args.replays):
        gr.replay()
    torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(-1, 8)
    st = st[st[:, 0] != 0]
    t0 = st[:, 0].min()
    rel = (st[:, :6] - t0) * 10.0  # ns (100 MHz)
    print(f"{args.shape} {args.kernel}: {len(st)} workgroups; ns since first workgroup start")
    names = ["start", "loads issued", "absmax known", "planes ready", "mfma done", "end"]
    for i, n in enumerate(names):
        c = rel[:, i]
        print(f"  {n:14s} min {c.min():8.0f}  median {np.median(c):8.0f}  max {c.max():8.0f}")
    d = rel[:, 1:6] - rel[:, 0:5]
    print("  per-stage median ns:", ", ".join(f"{n}={np.median(d[:, i]):.0f}" for i, n in enumerate(names[1:])))
    print(f"  last end - first start = {rel[:, 5].max():.0f} ns")
    clk = (st[:, 7] - st[:, 6]) / ((st[:, 5] - st[:, 0]) * 10e-9) / 1e9
    print(f"  shader clock during the kernel (s_memtime / s_memrealtime): median {np.median(clk):.2f} GHz")


if __name__ == "__main__":
    main()
