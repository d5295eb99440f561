"""One process for the counter passes over the prefill matmul (tools/pmc_gemm_clock.sh):
    python3 tools/pmc_gemm_once.py [m = 4096] [random|zeros] [launches = 4]"""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("bitnet-rs_amd"); hip = pkg.load(); hip.init(0)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
data = sys.argv[2] if len(sys.argv) > 2 else "random"
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rng = np.random.default_rng(0)
n, k = 13824, 2560
qs = rng.integers(0, 256, n * k // 4, dtype=np.uint8)
if data == "zeros":
    qs[:] = 0xAA
h = hip.weights_upload_qk256(qs, n, k, k // 4)
x = torch.randn(m, k, device="cuda")
if data == "zeros":
    x.zero_()
y = torch.empty(m, n, device="cuda")
wsb = hip.matmul_workspace_bytes(m, k, 2); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
for _ in range(launches):
    hip.matmul_fused_dev(h, x, y, m, ws, wsb, digits=2)
torch.cuda.synchronize()
