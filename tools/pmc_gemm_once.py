import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("bitnet-rs_amd"); hip = pkg.load(); hip.init(0)
rng = np.random.default_rng(0)
n, k, m = 13824, 2560, 4096
h = hip.weights_upload_qk256(rng.integers(0, 256, n * k // 4, dtype=np.uint8), n, k, k // 4)
x = torch.randn(m, k, device="cuda"); y = torch.empty(m, n, device="cuda")
wsb = hip.matmul_workspace_bytes(m, k, 2); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
for _ in range(4):
    hip.matmul_fused_dev(h, x, y, m, ws, wsb, digits=2)
torch.cuda.synchronize()
