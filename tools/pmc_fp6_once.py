"""One process for the counter passes over the 2-digit prompt matmul (tools/pmc_fp6.sh):
    python3 tools/pmc_fp6_once.py [fp6|fp6x|int8] [launches = 3]   -- gate|up shape, 4096 tokens, LayerNorm in, silu * up as f16 rows out"""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("bitnet-rs_amd"); hip = pkg.load(); hip.init(0)
form = sys.argv[1] if len(sys.argv) > 1 else "fp6"
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 3
fl = {"fp6": 16, "fp6x": 16 | 32, "int8": 8}[form] | 1 | 4  # + FUSE_SILU_MUL | FUSE_Y_F16
n, k, m = 6912, 2560, 4096
rng = np.random.default_rng(0)
hg = hip.weights_upload_qk256(rng.integers(0, 256, n * k // 4, dtype=np.uint8), n, k, k // 4)
hu = hip.weights_upload_qk256(rng.integers(0, 256, n * k // 4, dtype=np.uint8), n, k, k // 4)
h = hip.weights_concat([hg, hu], interleave16=True)
gamma = (torch.rand(k, device="cuda") + 0.5) / 80
x = torch.randn(m, k, device="cuda")
yh = torch.empty(m, n, device="cuda", dtype=torch.float16)
wsb = hip.matmul_workspace_bytes(m, k, 2); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
for _ in range(launches):
    hip.matmul_fused_dev(h, x, yh, m, ws, wsb, ln_gamma=gamma, ln_eps=1e-5, digits=2, flags=fl)
torch.cuda.synchronize()
print(form, hip.matmul_last_tile(), "resident", hip.matmul_last_resident_fp4())
