#!/usr/bin/env python3
"""Developer tool: per-workgroup overhead of the prefill matmul.  13824 output rows x 4096 tokens, 2 digits, K = 1280 .. 7680: the slope
of time over K is the steady K-step, the intercept what a workgroup costs beyond its K loop (launch, first loads, epilogue stores).
    python tools/perf_gemm_ksweep.py [flags = 0: e.g. 8 int8 planes, 16 fp6 on the resident image, 48 fp6 expanding in the loop]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bitnet-rs_amd")
hip = pkg.load(); hip.init(0)
rng = np.random.default_rng(0)
m = 4096
FL = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for n in (13824, 2560):
    pts = []
    for k in (1280, 2560, 3840, 5120, 7680):
        stride = k // 256 * 64
        h = hip.weights_upload_qk256(rng.integers(0, 256, n * stride, dtype=np.uint8), n, k, stride)
        x = torch.randn(m, k, device="cuda"); y = torch.empty(m, n, device="cuda")
        wsb = hip.matmul_workspace_bytes(m, k, 2); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        for _ in range(3): hip.matmul_fused_dev(h, x, y, m, ws, wsb, digits=2, flags=FL)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): hip.matmul_fused_dev(h, x, y, m, ws, wsb, digits=2, flags=FL)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        pts.append((k // 256, us))
        print(f"rows {n} K {k} ({k // 256} steps): {us:.1f} us (quantiser included)", flush=True)
        hip.weights_free(h)
    (s0, t0), (s1, t1) = pts[0], pts[-1]
    slope = (t1 - t0) / (s1 - s0)
    print(f"  slope {slope:.2f} us per K step (whole launch), intercept {t0 - slope * s0:.1f} us")
