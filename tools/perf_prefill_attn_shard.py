"""time one rank's gathered attention at world 8, T 8192 (2B-4T heads) and the unsharded 4096 / 1024-token calls"""
import importlib, sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("bitnet-rs_amd"); hip = pkg.load(); hip.init(0)
from oracle import oracle as orc
NH, NK, D = 20, 5, 128
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for T, world in ((8192, 8), (8192, 4), (4096, 1), (1024, 1), (512, 1)):
    max_pos = T
    sin, cos = orc.rope_tables(D, max_pos, 10000.0)
    sin_d, cos_d = torch.from_numpy(sin).cuda(), torch.from_numpy(cos).cuda()
    kc, vc = torch.zeros(NK * max_pos * D, device="cuda"), torch.zeros(NK * max_pos * D, device="cuda")
    if world == 1:
        qkv = torch.randn(T, (NH + 2 * NK) * D, device="cuda")
        wsb = hip.attention_prefill_workspace_bytes(NH, NK, T); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        out = torch.empty(T, NH * D, device="cuda")
        us = t(lambda: hip.attention_prefill_dev(qkv, sin_d, cos_d, kc, vc, NH, NK, D, max_pos, T, ws, wsb, out))
        print(f"unsharded T={T}: {us:.1f} us (prep + attention [+ merge])")
    else:
        nq, chunk = T // world, T // (2 * world)
        for rank in (0, world - 1):
            rows = np.concatenate([np.arange(rank * chunk, (rank + 1) * chunk), np.arange((2 * world - 1 - rank) * chunk, (2 * world - rank) * chunk)])
            q = torch.randn(nq, NH * D, device="cuda"); kv = torch.randn(T, 2 * NK * D, device="cuda").half()
            bp = torch.from_numpy(rows[::64].astype(np.int32)).cuda()
            wsb = hip.attention_prefill_sharded_workspace_bytes(NH, NK, nq, T); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
            out = torch.empty(nq, NH * D, device="cuda")
            us = t(lambda: hip.attention_prefill_gathered_dev(q, NH * D, bp, nq, kv, T, world, True, sin_d, cos_d, kc, vc, False, NH, NK, D, max_pos, ws, wsb, out))
            print(f"world {world} rank {rank} T={T} nq={nq}: {us:.1f} us (prep of all {T} keys + attention + merge)")
