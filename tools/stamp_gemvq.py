"""Developer tool: where does one k_gemv_q launch spend its time?  Diagnostic library (python bitnet-rs_amd/build.py --diag),
s_memrealtime stamps per workgroup (wave 0 and the last wave): 0 start | 1 loads issued | 2 QAct in LDS | 3 MFMA loop done |
4 past the barrier | 5 end.  Never used by tests / bench.

    python tools/stamp_gemvq.py [--shape gateup|qkv|o|down] [--fmt i2s32|qk256]
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
SHAPES = {"qkv": (3840, 2560), "o": (2560, 2560), "gateup": (6912, 2560), "down": (2560, 6912)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="gateup")
    ap.add_argument("--layers", type=int, default=30)
    ap.add_argument("--replays", type=int, default=3)
    ap.add_argument("--fmt", default="i2s32", choices=["qk256", "i2s32"])
    ap.add_argument("--same", action="store_true", help="every launch of the graph reads the SAME matrix (L2 / MALL resident): the upper bound a weight prefetch could reach")
    args = ap.parse_args()
    hip = pkg.HipLib(pkg.LIB_PATH.replace(".so", "_diag.so"))
    hip.init(0)
    hip.c.bitnet_hip_debug_set_stamps.argtypes = [C.c_void_p]
    rows, cols = SHAPES[args.shape]
    rng = np.random.default_rng(0)

    def mk(r=rows):
        if args.fmt == "i2s32":
            sc = (2.0 / ((np.arange(r * (cols // 32)) % 100) + 1)).astype(np.float16).astype(np.float32)
            return hip.weights_upload_i2s(rng.integers(0, 256, r * cols // 4, dtype=np.uint8), sc, r, cols, 32)
        return hip.weights_upload_qk256(rng.integers(0, 256, r * cols // 4, dtype=np.uint8), r, cols, cols // 4)

    paired = args.shape == "gateup"
    ln = args.shape in ("gateup", "qkv")
    handles = []
    for _ in range(1 if args.same else args.layers):
        if paired:
            a, b = mk(), mk()
            handles.append(hip.weights_concat([a, b], interleave16=True))
            hip.weights_free(a), hip.weights_free(b)
        else:
            handles.append(mk())
    if args.same:
        handles = handles * args.layers
    gamma = torch.full((cols,), 0.0125, device="cuda")
    if ln:
        for h in handles:
            hip.weights_bind_ln(h, gamma)
    out_rows = rows  # paired: 2 * rows / 2
    x = torch.randn(cols, device="cuda")
    qin = torch.zeros(hip.qact_bytes(cols), dtype=torch.uint8, device="cuda")
    st_in = torch.zeros(cols // 16 * 2, dtype=torch.float64, device="cuda")
    hip.quantize_act_dev(x, gamma if ln else None, cols, qin, st_in)
    y = torch.empty(out_rows, device="cuda")
    res = torch.randn(out_rows, device="cuda")
    gout = torch.full((out_rows,), 0.0125, device="cuda")
    qout = torch.zeros(hip.qact_bytes(out_rows), dtype=torch.uint8, device="cuda")
    st_out = torch.zeros(out_rows // 16 * 2, dtype=torch.float64, device="cuda")
    stamps = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")

    def launch(h, st):
        if args.shape == "gateup":
            hip.gemv_q_dev(h, qin, stats_in=st_in, ln_gamma=gamma, ln_eps=1e-5, flags=1, qact_out=qout, stream=st)
        elif args.shape == "qkv":
            hip.gemv_q_dev(h, qin, y=y, stats_in=st_in, ln_gamma=gamma, ln_eps=1e-5, stream=st)
        else:
            hip.gemv_q_dev(h, qin, y=y, residual=res, qact_out=qout, gamma_out=gout, stats_out=st_out, stream=st)

    cs = torch.cuda.current_stream().cuda_stream
    for h in handles:
        launch(h, cs)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        s = torch.cuda.current_stream().cuda_stream
        for i, h in enumerate(handles):
            hip.c.bitnet_hip_debug_set_stamps(C.c_void_p(stamps.data_ptr() if i == len(handles) - 1 else 0))
            launch(h, s)
    for _ in range(args.replays):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{args.shape} {args.fmt}: {e0.elapsed_time(e1) * 1e3 / 20 / len(handles):.2f} us per launch (diagnostic build, graph of {len(handles)})")
    st = stamps.cpu().numpy().reshape(-1, 16)
    st = st[st[:, 0] != 0]
    t0 = st[:, 0].min()
    names = ["start", "loads issued", "QAct in LDS", "mfma done", "past barrier", "end"]
    for off, who in ((0, "wave 0"), (8, "last wave")):
        rel = (st[:, off:off + 6] - t0) * 10.0
        print(f" {who}: {len(st)} workgroups; ns since the first workgroup's start")
        for i, n in enumerate(names):
            c = rel[:, i]
            c = c[st[:, off + i] != 0]
            if len(c):
                print(f"   {n:14s} min {c.min():8.0f}  median {np.median(c):8.0f}  max {c.max():8.0f}")


if __name__ == "__main__":
    main()
