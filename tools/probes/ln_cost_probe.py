"""How much of a fused LayerNorm -> GEMV launch is the LayerNorm prologue?  30 distinct matrices,
one hipGraph of 30 launches, with and without ln_gamma (same shapes)."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("bitnet-rs_amd")
hip = pkg.load(); hip.init(0)
rng = np.random.default_rng(0)
for name, (n, k) in {"qkv": (3840, 2560), "o-like": (2560, 2560)}.items():
    stride = k // 256 * 64
    hs = [hip.weights_upload_qk256(rng.integers(0, 256, n * stride, dtype=np.uint8), n, k, stride) for _ in range(30)]
    x = torch.randn(k, device="cuda"); y = torch.empty(n, device="cuda"); g = torch.full((k,), 0.0125, device="cuda")
    for ln in (True, False):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            def launch():
                for h in hs:
                    hip.gemv_fused_dev(h, x, y, 1, ln_gamma=g if ln else None, ln_eps=1e-5, stream=s.cuda_stream)
            launch(); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                launch()
            gr.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(20): gr.replay()
            e1.record(s); torch.cuda.synchronize()
            print(f"{name} ln={ln}: {e0.elapsed_time(e1) * 1e3 / 20 / 30:.2f} us/launch", flush=True)
    for h in hs: hip.weights_free(h)
