"""One process for the counter passes over the f16 chain matmul (tools/pmc_f16a.sh):
    python3 tools/pmc_f16a_once.py [i2s|qk256] [launches = 3]   -- gate|up shape, 4096 tokens, f16 rows in, f16 rows out"""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("bitnet-rs_amd"); synth = importlib.import_module("bitnet-rs_amd.synth"); hip = pkg.load(); hip.init(0)
fmt = sys.argv[1] if len(sys.argv) > 1 else "i2s"
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n, k, m = 13824, 2560, 4096
if fmt == "i2s":
    wq, ws_ = synth.ternary_weights(n, k, 32, 42, 0, 1); h = hip.weights_upload_i2s(wq, ws_, n, k, 32)
else:
    h = hip.weights_upload_qk256(np.random.default_rng(0).integers(0, 256, n * k // 4, dtype=np.uint8), n, k, k // 4)
xh = torch.randn(m, k, device="cuda").half(); yh = torch.empty(m, n, device="cuda", dtype=torch.float16)
for _ in range(launches):
    hip.matmul_f16_dev(h, xh, m, yh=yh)
torch.cuda.synchronize()
