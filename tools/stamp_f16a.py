"""Developer tool (GPU box, diagnostic build: python -c "import importlib.util as u; ..." or bitnet-rs_amd/build.py's build(diag=True)):
where a wave of k_gemm_f16a spends its cycles.  s_memtime stamps per wave, summed over the K steps of its tile:
  0 staging stores (incl. the wait for the tile's global loads) | 1 issuing the next loads | 2 expansion + operand reads + MFMAs | 3 barrier
  4 prologue (first loads, first tile into LDS) | kernel begin / end stamps.
    python tools/stamp_f16a.py [i2s|qk256] [gateup|down|o|qkv]"""
import ctypes as C, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("bitnet-rs_amd"); synth = importlib.import_module("bitnet-rs_amd.synth")
fmt = sys.argv[1] if len(sys.argv) > 1 else "i2s"
shape = sys.argv[2] if len(sys.argv) > 2 else "gateup"
hip = pkg.HipLib(pkg.LIB_PATH.replace(".so", "_diag.so")); hip.init(0)
hip.c.bitnet_hip_debug_set_stamps.argtypes = [C.c_void_p]
n, k = {"gateup": (13824, 2560), "down": (2560, 6912), "o": (2560, 2560), "qkv": (3840, 2560)}[shape]
m = 4096
if fmt == "i2s":
    wq, ws_ = synth.ternary_weights(n, k, 32, 42, 0, 1); h = hip.weights_upload_i2s(wq, ws_, n, k, 32)
else:
    h = hip.weights_upload_qk256(np.random.default_rng(0).integers(0, 256, n * k // 4, dtype=np.uint8), n, k, k // 4)
xh = torch.randn(m, k, device="cuda").half(); yh = torch.empty(m, n, device="cuda", dtype=torch.float16)
stamps = torch.zeros(8192 * 4 * 8, dtype=torch.int64, device="cuda")
for _ in range(3):
    hip.matmul_f16_dev(h, xh, m, yh=yh)
torch.cuda.synchronize()
hip.c.bitnet_hip_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
hip.matmul_f16_dev(h, xh, m, yh=yh)
torch.cuda.synchronize()
hip.c.bitnet_hip_debug_set_stamps(C.c_void_p(0))
st = stamps.cpu().numpy().reshape(-1, 8)
st = st[st[:, 7] > 0]
ph = st[:, :4].astype(np.float64)
tot = (st[:, 7] - st[:, 6]).astype(np.float64)
loop = ph.sum(axis=1)
print(f"{fmt} {shape}: {len(st)} waves, tile {hip.matmul_last_tile()} rows/wave {hip.matmul_last_wave_rows()}")
print(f"  wave lifetime (cycles of s_memtime): median {np.median(tot):.0f}; K loop {np.median(loop):.0f} ({np.median(loop / tot):.2f}), prologue {np.median(st[:, 4]):.0f}, "
      f"epilogue {np.median(tot - loop - st[:, 4]):.0f}")
names = ["staging stores (+ wait for the loads)", "issuing next loads", "expansion + reads + MFMAs", "barrier"]
for i, nm in enumerate(names):
    print(f"  phase {i} {nm:40s}: median {np.median(ph[:, i]):9.0f} cycles = {np.median(ph[:, i] / loop):.3f} of the K loop (p10 {np.percentile(ph[:, i] / loop, 10):.3f}, p90 {np.percentile(ph[:, i] / loop, 90):.3f})")
span = st[:, 7].max() - st[:, 6].min()
print(f"  launch span {span} cycles; sum of wave lifetimes / (span x 2048 wave slots) = {tot.sum() / (span * 2048.0):.3f}")
