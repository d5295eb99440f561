"""Developer tool: time the whole-prompt attention (4096 tokens, 20 / 5 heads) with parts of its key loop compiled out
(BH_ABLATE bits: 64 no LDS operand reads, 128 no softmax arithmetic, 256 no tile staging).  Results are wrong by construction.
BH_ABLATE=n python bitnet-rs_amd/build.py; python tools/ablate_attn.py n ..."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("bitnet-rs_amd")
from oracle import oracle
for tag in sys.argv[1:] or ["0"]:
    path = pkg.LIB_PATH if tag == "0" else pkg.LIB_PATH.replace(".so", f"_ablate{tag}.so")
    hip = pkg.HipLib(path); hip.init(0)
    T, H, KV, D, MP = 4096, 20, 5, 128, 4160
    qkv = torch.randn(T, (H + 2 * KV) * D, device="cuda")
    sin, cos = oracle.rope_tables(D, MP, 10000.0)
    sd, cd = torch.from_numpy(sin).cuda(), torch.from_numpy(cos).cuda()
    kc = torch.zeros(KV * 4160 * D, device="cuda"); vc = torch.zeros(KV * 4160 * D, device="cuda")
    wsb = hip.attention_prefill_workspace_bytes(H, KV, T)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    out = torch.empty(T, H * D, device="cuda")
    run = lambda: hip.attention_prefill_dev(qkv, sd, cd, kc, vc, H, KV, D, MP, T, ws, wsb, out)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print(f"ablate {tag:>3s}: attention 4096 tokens: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us (prep + attention)", flush=True)
