"""Developer tool: time the tiled matmul (gate|up shape, 4096 tokens, 2 digits) with parts of its K loop compiled out
(BH_ABLATE bit mask: 1 no LDS operand reads, 2 no code expansion, 4 no LDS staging stores, 8 no activation loads).
Results are wrong by construction; only the time matters.   BH_ABLATE=n python bitnet-rs_amd/build.py; python tools/ablate_gemm.py n"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("bitnet-rs_amd")
for tag in sys.argv[1:] or ["0"]:
    path = pkg.LIB_PATH if tag == "0" else pkg.LIB_PATH.replace(".so", f"_ablate{tag}.so")
    hip = pkg.HipLib(path); hip.init(0)
    rng = np.random.default_rng(0)
    for (n, k) in ((13824, 2560), (2560, 6912)):
        m = 4096
        h = hip.weights_upload_qk256(rng.integers(0, 256, n * (k // 256) * 64, dtype=np.uint8), n, k, k // 256 * 64)
        x = torch.randn(m, k, device="cuda"); y = torch.empty(m, n, device="cuda")
        wsb = hip.matmul_workspace_bytes(m, k, 2); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        for _ in range(3): hip.matmul_fused_dev(h, x, y, m, ws, wsb, digits=2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): hip.matmul_fused_dev(h, x, y, m, ws, wsb, digits=2)
        e1.record(); torch.cuda.synchronize()
        print(f"ablate {tag:>2s}: {n}x{k} m={m}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us (incl. the row quantiser)", flush=True)
        hip.weights_free(h)
