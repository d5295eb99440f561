#!/usr/bin/env python3
"""Developer tool: per-workgroup overhead of the f16 chain matmul (k_gemm_f16a).  Output rows x 4096 tokens, K = 1280 .. 7680, f16 rows in,
f16 rows out (the chain's hand-over): the slope of time over K is the steady K step, the intercept what a workgroup costs beyond its K
loop (launch, first loads, epilogue).  python tools/perf_f16a_ksweep.py [i2s|qk256]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bitnet-rs_amd"); synth = importlib.import_module("bitnet-rs_amd.synth")
hip = pkg.load(); hip.init(0)
fmt = sys.argv[1] if len(sys.argv) > 1 else "i2s"
rng = np.random.default_rng(0)
m = 4096
for n in (13824, 2560):
    pts = []
    for k in (1280, 2560, 3840, 5120, 7680):
        if fmt == "i2s":
            wq, ws_ = synth.ternary_weights(n, k, 32, 42, 0, 1)
            h = hip.weights_upload_i2s(wq, ws_, n, k, 32)
        else:
            h = hip.weights_upload_qk256(rng.integers(0, 256, n * (k // 256) * 64, dtype=np.uint8), n, k, k // 256 * 64)
        xh = torch.randn(m, k, device="cuda").half(); yh = torch.empty(m, n, device="cuda", dtype=torch.float16)
        for _ in range(3): hip.matmul_f16_dev(h, xh, m, yh=yh)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): hip.matmul_f16_dev(h, xh, m, yh=yh)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        pts.append((k // 256, us))
        print(f"{fmt} rows {n} K {k} ({k // 256} steps): {us:.1f} us  rows/wave {hip.matmul_last_wave_rows()}", flush=True)
        hip.weights_free(h)
    (s0, t0), (s1, t1) = pts[0], pts[-1]
    slope = (t1 - t0) / (s1 - s0)
    print(f"  slope {slope:.2f} us per K step (whole launch), intercept {t0 - slope * s0:.1f} us")
