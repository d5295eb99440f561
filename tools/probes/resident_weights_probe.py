"""Does the fused gate|up GEMV get faster when its weights are cache resident?  One hipGraph of 30 launches:
(a) 30 distinct matrices (HBM stream, as in the decode step), (b) the SAME matrix 30 times (8.8 MB: L2 + Infinity Cache)."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("bitnet-rs_amd")
hip = pkg.load(); hip.init(0)
rng = np.random.default_rng(0)
n, k = 6912, 2560
stride = k // 256 * 64
def mk():
    a = hip.weights_upload_qk256(rng.integers(0, 256, n * stride, dtype=np.uint8), n, k, stride)
    b = hip.weights_upload_qk256(rng.integers(0, 256, n * stride, dtype=np.uint8), n, k, stride)
    h = hip.weights_concat([a, b], interleave16=True); hip.weights_free(a); hip.weights_free(b); return h
hs = [mk() for _ in range(30)]
x = torch.randn(k, device="cuda"); y = torch.empty(n, device="cuda"); g = torch.full((k,), 0.0125, device="cuda")
for h in hs: hip.weights_bind_ln(h, g)
for label, seq in (("30 distinct matrices", hs), ("same matrix x30", [hs[0]] * 30)):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        def launch():
            for h in seq: hip.gemv_fused_dev(h, x, y, 1, ln_gamma=g, ln_eps=1e-5, flags=1, stream=s.cuda_stream)
        launch(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s): launch()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20): gr.replay()
        e1.record(s); torch.cuda.synchronize()
        print(f"{label}: {e0.elapsed_time(e1) * 1e3 / 600:.2f} us/launch", flush=True)
