"""Developer tool: where does one decode-attention launch (k_attn_partial) spend its time?  Diagnostic library
(python bitnet-rs_amd/build.py --diag), s_memrealtime stamps of thread 0 of every live workgroup:
0 position arrived | 1 all loads requested | 2 past the RoPE barrier | 3 scores done (K arrived) | 4 softmax weights in LDS | 5 end.
The launch sits where it sits in the real step: behind the q|k|v GEMV that produces its input (another kernel, other CUs) and
ahead of the merging o-projection, `layers` such triples in one hipGraph; the stamps are taken in the LAST triple.
Never used by tests / bench.

    python tools/stamp_attn.py [--keys 150] [--kv16]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keys", type=int, default=150)
    ap.add_argument("--layers", type=int, default=30)
    ap.add_argument("--kv16", action="store_true")
    ap.add_argument("--wide", action="store_true", help="128-position workgroups")
    args = ap.parse_args()
    hip = pkg.HipLib(pkg.LIB_PATH.replace(".so", "_diag.so"))
    hip.init(0)
    hip.c.bitnet_hip_debug_set_stamps.argtypes = [C.c_void_p]
    H, NH, NK, D, max_pos = 2560, 20, 5, 128, 1024
    rng = np.random.default_rng(0)
    rows = (NH + 2 * NK) * D
    mk = lambda r, c: hip.weights_upload_i2s(rng.integers(0, 256, r * c // 4, dtype=np.uint8),
                                             (2.0 / ((np.arange(r * (c // 32)) % 100) + 1)).astype(np.float16).astype(np.float32), r, c, 32)
    gamma = torch.full((H,), 0.0125, device="cuda")
    qkvs, os_ = [], []
    for _ in range(args.layers):
        h = mk(rows, H)
        hip.weights_bind_ln(h, gamma)
        qkvs.append(h)
        os_.append(mk(H, NH * D))
    x = torch.randn(H, device="cuda")
    qin = torch.zeros(hip.qact_bytes(H), dtype=torch.uint8, device="cuda")
    st_in = torch.zeros(H // 16 * 2, dtype=torch.float64, device="cuda")
    hip.quantize_act_dev(x, gamma, H, qin, st_in)
    qkv = torch.zeros(rows, device="cuda")
    from oracle import oracle as orc  # RoPE tables only (developer tool)

    sin, cos = orc.rope_tables(D, max_pos, 10000.0)
    sin_d, cos_d = torch.from_numpy(sin).cuda(), torch.from_numpy(cos).cuda()
    n = NK * max_pos * D
    kc = [torch.randn(n // (2 if args.kv16 else 1), device="cuda") * 0.5 for _ in range(args.layers)]
    vc = [torch.randn(n // (2 if args.kv16 else 1), device="cuda") * 0.5 for _ in range(args.layers)]
    if args.kv16:
        kc = [t.half().view(torch.float32) if False else torch.zeros(n // 2, device="cuda") for t in kc]
        vc = [torch.zeros(n // 2, device="cuda") for t in vc]
    sb = hip.c.bitnet_hip_attention_scratch_bytes(NK, max_pos)
    scratch = torch.zeros(sb // 4 + 16, device="cuda")
    pos_d = torch.tensor([args.keys - 1], dtype=torch.int32, device="cuda")
    y = torch.zeros(H, device="cuda")
    res = torch.randn(H, device="cuda")
    qout = torch.zeros(hip.qact_bytes(H), dtype=torch.uint8, device="cuda")
    qatt = torch.zeros(hip.qact_bytes(NH * D), dtype=torch.uint8, device="cuda")
    st_out = torch.zeros(H // 16 * 2, dtype=torch.float64, device="cuda")
    stamps = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
    merge = args.keys <= 256 and not args.wide

    stamps_q = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
    stamps_o = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
    setst = lambda t: hip.c.bitnet_hip_debug_set_stamps(C.c_void_p(t.data_ptr() if t is not None else 0))  # read by a launcher when its call is captured

    def triple(i, s, stamp=False):
        setst(stamps_q if stamp else None)
        hip.gemv_q_dev(qkvs[i], qin, y=qkv, stats_in=st_in, ln_gamma=gamma, ln_eps=1e-5, stream=s)
        setst(stamps if stamp else None)
        attn(i, s)
        setst(stamps_o if stamp else None)
        try:
            tail(i, s)
        finally:
            setst(None)

    def tail(i, s):
        if merge:
            hip.gemv_attn_merge_q_dev(os_[i], scratch, NH, NK, max_pos, pos_d, y, qout, residual=res, gamma_out=gamma, stats_out=st_out, stream=s)
        else:
            hip.gemv_q_dev(os_[i], qatt, y=y, residual=res, qact_out=qout, gamma_out=gamma, stats_out=st_out, stream=s)

    def attn(i, s):
        if merge:
            hip.attention_decode_q_dev(qkv, sin_d, cos_d, kc[i], vc[i], NH, NK, D, max_pos, pos_d, scratch, None, None, kv_f16=args.kv16, partial=True, stream=s)
        else:
            hip.attention_decode_q_dev(qkv, sin_d, cos_d, kc[i], vc[i], NH, NK, D, max_pos, pos_d, scratch, None, qatt, wide=args.wide, kv_f16=args.kv16, stream=s)

    cs = torch.cuda.current_stream().cuda_stream
    for i in range(args.layers):
        triple(i, cs)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        s = torch.cuda.current_stream().cuda_stream
        for i in range(args.layers):
            triple(i, s, stamp=i == args.layers - 1)
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{args.keys} keys ({'merging o-projection' if merge else 'combine kernel'}{', f16 cache' if args.kv16 else ''}): "
          f"{e0.elapsed_time(e1) * 1e3 / 20 / args.layers:.2f} us per (q|k|v, attention, o) triple (diagnostic build)")
    st = stamps.cpu().numpy().reshape(-1, 8)
    st = st[st[:, 0] != 0]
    t0 = st[:, 0].min()
    names = ["position arrived", "loads requested", "past RoPE barrier", "scores done", "softmax in LDS", "end"]
    rel = (st[:, :6] - t0) * 10.0
    print(f" {len(st)} live workgroups; ns since the first workgroup had its position")
    for i, nme in enumerate(names):
        c = rel[:, i]
        print(f"   {nme:18s} min {c.min():8.0f}  median {np.median(c):8.0f}  max {c.max():8.0f}")
    # the neighbours (k_gemv_q stamps: 16 words per workgroup, wave 0 in words 0..5: start .. end)
    sq = stamps_q.cpu().numpy().reshape(-1, 16)
    sq = sq[sq[:, 0] != 0]
    so = stamps_o.cpu().numpy().reshape(-1, 16)
    so = so[so[:, 0] != 0]
    if len(sq) and len(so):
        print(f" q|k|v GEMV before it: first start {(sq[:, 0].min() - t0) * 10.0:.0f} ns, last end {(sq[:, 5][sq[:, 5] != 0].max() - t0) * 10.0:.0f} ns")
        print(f" o-projection behind it: first start {(so[:, 0].min() - t0) * 10.0:.0f} ns, median start {(np.median(so[:, 0]) - t0) * 10.0:.0f} ns, last end "
              f"{(so[:, 5][so[:, 5] != 0].max() - t0) * 10.0:.0f} ns")


if __name__ == "__main__":
    main()
