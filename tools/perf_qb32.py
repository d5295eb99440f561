#!/usr/bin/env python3
"""Same-input timing of the fp6 x fp4 prompt matmul's two input forms on the gate|up launch (13824 x 2560, 4096 tokens, LayerNorm + silu * up -> f16 rows):
   row-scaled fp6 digits behind their quantiser (BITNET_HIP_FUSE_FP6_DIGITS) vs QB32 rows (bitnet_hip_matmul_qb32_dev); HIP events, back-to-back launches.
    python tools/perf_qb32.py [reps = 20]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bitnet-rs_amd"); hip = pkg.load(); hip.init(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n, k, m = 6912, 2560, 4096
rng = np.random.default_rng(0)
hg = hip.weights_upload_qk256(rng.integers(0, 256, n * k // 4, dtype=np.uint8), n, k, k // 4)
hu = hip.weights_upload_qk256(rng.integers(0, 256, n * k // 4, dtype=np.uint8), n, k, k // 4)
h = hip.weights_concat([hg, hu], interleave16=True)
gamma = (torch.rand(k, device="cuda") + 0.5) / 80
hip.weights_bind_ln(h, gamma)
x = torch.randn(m, k, device="cuda") * 1.3 + 0.1
yh = torch.empty(m, n, device="cuda", dtype=torch.float16)
wsb = hip.matmul_workspace_bytes(m, k, 2); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
qb = torch.zeros(hip.qb32_bytes(m, k), dtype=torch.uint8, device="cuda")
# 40 statistics partials per token as the o- / down-projection leaves them (here: the row's sums in partial 0, zeros elsewhere)
st = torch.zeros(40 * m * 2, device="cuda")
hip.rows_to_qb32_dev(x, gamma, m, k, qb, st)

def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

t_q = timed(lambda: hip.rows_to_qb32_dev(x, gamma, m, k, qb, st))
t_row = timed(lambda: hip.matmul_fused_dev(h, x, yh, m, ws, wsb, ln_gamma=gamma, ln_eps=1e-5, digits=2, flags=1 | 4 | 16))
t_qb1 = timed(lambda: hip.matmul_qb32_dev(h, qb, m, stats_in=st, n_stats=1, ln_gamma=gamma, ln_eps=1e-5, flags=1, yh=yh))
t_qb40 = timed(lambda: hip.matmul_qb32_dev(h, qb, m, stats_in=st, n_stats=40, ln_gamma=gamma, ln_eps=1e-5, flags=1, yh=yh))
print(f"row-scaled fp6 digits (quantiser + matmul) {t_row:7.1f} us | QB32 matmul, 1 partial {t_qb1:7.1f} us, 40 partials {t_qb40:7.1f} us | rows_to_qb32 {t_q:6.1f} us")
