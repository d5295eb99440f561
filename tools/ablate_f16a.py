"""Developer tool: time k_gemm_f16a (the f16 chain's matmul; gate|up and down shapes, 4096 tokens) with parts of its K loop compiled out
(BH_ABLATE bit mask: 1 no LDS operand reads, 2 no code expansion).  Results are wrong by construction; only the time matters.
    BH_ABLATE=n python bitnet-rs_amd/build.py; python tools/ablate_f16a.py [i2s|qk256] 0 n ..."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("bitnet-rs_amd"); synth = importlib.import_module("bitnet-rs_amd.synth")
fmt = sys.argv[1] if len(sys.argv) > 1 else "i2s"
for tag in sys.argv[2:] or ["0"]:
    path = pkg.LIB_PATH if tag == "0" else pkg.LIB_PATH.replace(".so", f"_ablate{tag}.so")
    hip = pkg.HipLib(path); hip.init(0)
    rng = np.random.default_rng(0)
    for (n, k) in ((13824, 2560), (2560, 6912), (2560, 2560)):
        m = 4096
        if fmt == "i2s":
            wq, ws_ = synth.ternary_weights(n, k, 32, 42, 0, 1)
            h = hip.weights_upload_i2s(wq, ws_, n, k, 32)
        else:
            h = hip.weights_upload_qk256(rng.integers(0, 256, n * (k // 256) * 64, dtype=np.uint8), n, k, k // 256 * 64)
        xh = torch.randn(m, k, device="cuda").half(); y = torch.empty(m, n, device="cuda")
        for _ in range(3): hip.matmul_f16_dev(h, xh, m, y=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): hip.matmul_f16_dev(h, xh, m, y=y)
        e1.record(); torch.cuda.synchronize()
        print(f"{fmt} ablate {tag:>2s}: {n}x{k} m={m}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us  tile {hip.matmul_last_tile()} rows/wave {hip.matmul_last_wave_rows()}", flush=True)
        hip.weights_free(h)
