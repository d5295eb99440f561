"""Developer probe: the tiled matmul at small activation-row counts (one rank's share of a token-parallel prefill, short prompts)."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("bitnet-rs_amd")
hip = pkg.load(); hip.init(0)
rng = np.random.default_rng(0)
for m in (128, 512, 1024):
    for (n, k) in ((13824, 2560), (3840, 2560), (2560, 2560), (2560, 6912)):
        h = hip.weights_upload_qk256(rng.integers(0, 256, n * (k // 256) * 64, dtype=np.uint8), n, k, k // 256 * 64)
        x = torch.randn(m, k, device="cuda"); y = torch.empty(m, n, device="cuda")
        wsb = hip.matmul_workspace_bytes(m, k, 2); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        for _ in range(3): hip.matmul_fused_dev(h, x, y, m, ws, wsb, digits=2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): hip.matmul_fused_dev(h, x, y, m, ws, wsb, digits=2)
        e1.record(); torch.cuda.synchronize()
        print(f"m={m}: {n}x{k}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
        hip.weights_free(h)
