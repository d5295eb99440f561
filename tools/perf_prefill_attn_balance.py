import importlib, sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("bitnet-rs_amd"); hip = pkg.load(); hip.init(0)
from oracle import oracle as orc
NH, NK, D, T = 20, 5, 128, 4096
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
sin, cos = orc.rope_tables(D, T, 10000.0)
sin_d, cos_d = torch.from_numpy(sin).cuda(), torch.from_numpy(cos).cuda()
kc, vc = torch.zeros(NK * T * D, device="cuda"), torch.zeros(NK * T * D, device="cuda")
q = torch.randn(T, NH * D, device="cuda"); kv = torch.randn(T, 2 * NK * D, device="cuda")
wsb = hip.attention_prefill_sharded_workspace_bytes(NH, NK, T, T); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
out = torch.empty(T, NH * D, device="cuda")
for name, pos in (("causal 0..4095", np.arange(0, T, 64)), ("every block at 2048 (balanced, same work)", np.full(T // 64, 2048)), ("every block at 4032 (2x work)", np.full(T // 64, 4032))):
    bp = torch.from_numpy(pos.astype(np.int32)).cuda()
    us = t(lambda: hip.attention_prefill_sharded_dev(q, NH * D, bp, T, kv, 2 * NK * D, T, sin_d, cos_d, kc, vc, NH, NK, D, T, ws, wsb, out))
    print(f"{name}: {us:.1f} us (prep + attention)", flush=True)
