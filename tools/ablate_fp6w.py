"""Developer tool: time k_gemm_fp6w (gate|up shape, 4096 tokens, LayerNorm in, silu * up as f16 rows out; quantiser included) with parts compiled out
(BH_ABLATE bit mask: 64 no epilogue stores (and no silu arithmetic), 128 no LDS-DMA staging after the first tile, 256 no weight loads after the first step).
Results are wrong by construction; only the time matters.   BH_ABLATE=n python bitnet-rs_amd/build.py; python tools/ablate_fp6w.py n   (EXPERIMENTS 8.7)"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("bitnet-rs_amd")
tag = sys.argv[1] if len(sys.argv) > 1 else "0"
hip = pkg.HipLib(pkg.LIB_PATH if tag == "0" else pkg.LIB_PATH.replace(".so", f"_ablate{tag}.so")); hip.init(0)
rng = np.random.default_rng(0)
n, k, m = 6912, 2560, 4096
hg = hip.weights_upload_qk256(rng.integers(0, 256, n * k // 4, dtype=np.uint8), n, k, k // 4)
hu = hip.weights_upload_qk256(rng.integers(0, 256, n * k // 4, dtype=np.uint8), n, k, k // 4)
h = hip.weights_concat([hg, hu], interleave16=True)
gamma = (torch.rand(k, device="cuda") + 0.5) / 80
x = torch.randn(m, k, device="cuda")
yh = torch.empty(m, n, device="cuda", dtype=torch.float16)
wsb = hip.matmul_workspace_bytes(m, k, 2); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
for rnd in range(3):
    for _ in range(3): hip.matmul_fused_dev(h, x, yh, m, ws, wsb, ln_gamma=gamma, ln_eps=1e-5, digits=2, flags=16 | 1 | 4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): hip.matmul_fused_dev(h, x, yh, m, ws, wsb, ln_gamma=gamma, ln_eps=1e-5, digits=2, flags=16 | 1 | 4, stream=torch.cuda.current_stream().cuda_stream)
    e1.record(); torch.cuda.synchronize()
    print("ablate", tag, "gate|up quant+matmul", round(e0.elapsed_time(e1) * 100, 1), "us", flush=True)
    torch.cuda._sleep(200000000)
