import importlib, os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("bitnet-rs_amd")
hip = pkg.HipLib(pkg.LIB_PATH.replace(".so", "_diag.so")); hip.init(0)
hip.c.bitnet_hip_debug_set_stamps.argtypes = [C.c_void_p]
rng = np.random.default_rng(0)
n, k = 6912, 2560; stride = k // 256 * 64
def mk():
    a = hip.weights_upload_qk256(rng.integers(0, 256, n * stride, dtype=np.uint8), n, k, stride)
    b = hip.weights_upload_qk256(rng.integers(0, 256, n * stride, dtype=np.uint8), n, k, stride)
    h = hip.weights_concat([a, b], interleave16=True); hip.weights_free(a); hip.weights_free(b); return h
hs = [mk() for _ in range(30)]
x = torch.randn(k, device="cuda"); y = torch.empty(n, device="cuda"); g = torch.full((k,), 0.0125, device="cuda")
for h in hs: hip.weights_bind_ln(h, g)
stamps = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
for rep in range(3):
    for i, h in enumerate(hs):
        hip.c.bitnet_hip_debug_set_stamps(C.c_void_p(stamps.data_ptr() if i == 29 else 0))
        hip.gemv_fused_dev(h, x, y, 1, ln_gamma=g, ln_eps=1e-5, flags=1)
torch.cuda.synchronize()
st = stamps.cpu().numpy().reshape(-1, 8); st = st[st[:, 0] != 0]
t0 = st.min(); rel = (st - t0) * 10.0
print("WGs", len(st), "per-wave MFMA-done ns (median over WGs):", np.median(rel, axis=0).round(0))
print("within-WG spread (max-min) median ns:", np.median(rel.max(axis=1) - rel.min(axis=1)), " first wave done median", np.median(rel.min(axis=1)), " last wave done median", np.median(rel.max(axis=1)))
