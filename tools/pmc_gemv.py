"""PMC-friendly driver: the decode step's fused GEMV launches (q|k|v, o, gate|up, down) issued
eagerly on the default stream, no graphs and no host C++ layer, so `rocprofv3 --pmc ...` can
attribute HBM counters per launch.  Same kernels, same shapes, same synthetic weights as
bench.py; used only to fill roofline.traffic.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 tools/pmc_gemv.py [--workload c2|c3] [--layers 8]
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
synth = importlib.import_module("bitnet-rs_amd.synth")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    hip = pkg.load()
    hip.init(0)
    cfg = synth.ModelConfig(**synth.BITNET_2B_4T)
    shapes = cfg.shapes()
    layers = []
    for l in range(args.layers):
        w = synth.make_layer(cfg, l, fmt="qk256" if args.workload == "c3" else "i2s", block=32)
        h = {}
        for name, (rows, cols) in shapes.items():
            if args.workload == "c3":
                h[name] = hip.weights_upload_qk256(w[name], rows, cols, cols // 256 * 64)
            else:
                h[name] = hip.weights_upload_i2s(w[name], w[name + "_scales"], rows, cols, 32)
        fused = {
            "qkv": hip.weights_concat([h["q"], h["k"], h["v"]]),
            "o": h["o"],
            "gateup": hip.weights_concat([h["gate"], h["up"]], interleave16=True),
            "down": h["down"],
            "g1": torch.from_numpy(w["attn_norm"]).cuda(),
            "g2": torch.from_numpy(w["ffn_norm"]).cuda(),
        }
        # the decoder binds the LayerNorm weights (LayerNorm applied after the product): same kernel variant here
        hip.weights_bind_ln(fused["qkv"], fused["g1"])
        hip.weights_bind_ln(fused["gateup"], fused["g2"])
        for n in ("q", "k", "v", "gate", "up"):
            hip.weights_free(h[n])
        layers.append(fused)
    x = torch.randn(cfg.hidden, device="cuda")
    x2 = torch.empty(cfg.hidden, device="cuda")
    qkv = torch.empty((cfg.n_heads + 2 * cfg.n_kv_heads) * cfg.head_dim, device="cuda")
    att = torch.randn(cfg.n_heads * cfg.head_dim, device="cuda")
    hbuf = torch.empty(cfg.ffn, device="cuda")
    for _ in range(args.reps):
        for L in layers:
            hip.gemv_fused_dev(L["qkv"], x, qkv, 1, ln_gamma=L["g1"], ln_eps=cfg.eps)
            hip.gemv_fused_dev(L["o"], att, x2, 1, residual=x)
            hip.gemv_fused_dev(L["gateup"], x2, hbuf, 1, ln_gamma=L["g2"], ln_eps=cfg.eps, flags=1)
            hip.gemv_fused_dev(L["down"], hbuf, x, 1, residual=x2)
    torch.cuda.synchronize()
    print("done", args.workload, args.layers, "layers")


if __name__ == "__main__":
    main()
