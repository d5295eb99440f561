"""Developer tool (GPU box): the provider-trait drop-ins (matmul_i2s, quantize, quantized_matmul_i2s, dequant_i2s, i2s_matmul_f32,
gemv_qk256 host-pointer forms) at random sizes against the oracle; integer / reference-order results must be bit-identical.
python tools/random_sweep_provider.py [n] [seed]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bitnet-rs_amd")
from oracle import oracle  # noqa: E402
hip = pkg.load(); hip.init(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 8)
bad = 0
def fail(*a):
    global bad
    bad += 1
    print("FAIL", *a, flush=True)
for case in range(n_cases):
    kind = int(rng.integers(0, 5))
    try:
        if kind == 0:  # KernelProvider::matmul_i2s: C = A_i8 . B_u8, exact
            m, n, k = int(rng.choice([1, 2, 5, 8, 9, 33])), int(rng.choice([1, 3, 4, 12, 64, 100, 640, 2560])), int(rng.choice([1, 4, 7, 64, 256, 300, 2560, 6912]))
            full = bool(rng.integers(0, 2)) and 128 * 255 * k < 2 ** 24
            a = rng.integers(-128, 128, m * k).astype(np.int8) if full else rng.integers(-2, 2, m * k).astype(np.int8)
            b = rng.integers(0, 256, k * n).astype(np.uint8) if full else rng.integers(0, 4, k * n).astype(np.uint8)
            if not np.array_equal(hip.matmul_i2s(a, b, m, n, k), oracle.matmul_i2s(a, b, m, n, k)):
                fail("matmul_i2s", m, n, k, full)
        elif kind == 1:  # KernelProvider::quantize I2S
            n = int(rng.choice([1, 31, 32, 33, 100, 4096, 65536, 100000, 1 << 20]))
            x = rng.normal(0, 1, n).astype(np.float32)
            x[rng.integers(0, n, max(1, n // 7))] = 0.0
            go, gs = hip.quantize(x, out_len=-(-n // 4))
            wo, ws = oracle.quantize_i2s(x, out_len=-(-n // 4))
            if not (np.array_equal(go, wo) and np.array_equal(gs, ws)):
                fail("quantize", n)
        elif kind == 2:  # QuantizedLinear::quantized_matmul_i2s composite
            m, n, k = int(rng.choice([1, 2, 4])), int(rng.choice([4, 8, 640, 2560])), int(rng.choice([4, 16, 20, 2560, 6912]))
            bs = int(rng.choice([4, 32]))
            x = rng.normal(0, 1.2, m * k).astype(np.float32)
            packed = rng.integers(0, 256, -(-k * n // 4), dtype=np.uint8)
            per_feature = bool(rng.integers(0, 2))
            scales = rng.uniform(0.1, 2.0, n if per_feature else max(1, k * n // bs // 3)).astype(np.float32)
            if not np.array_equal(hip.quantized_matmul_i2s(x, packed, scales, bs, m, n, k), oracle.quantized_matmul_i2s(x, packed, scales, bs, m, n, k)):
                fail("quantized_matmul_i2s", m, n, k, bs, per_feature)
        elif kind == 3:  # i2s_matmul_f32 (ternary family), default kernel: within the reference's own tolerance
            m, n, k = int(rng.choice([1, 2, 7])), int(rng.choice([1, 3, 16, 100, 640])), int(rng.choice([32, 64, 256, 288, 2560]))
            block = int(rng.choice([b for b in (32, 64, 128, 256) if k % b == 0]))
            codes = rng.choice(np.array([0, 1, 3], np.uint8), size=(n, k), p=[0.5, 0.25, 0.25])
            packed = (codes[:, 0::4] | codes[:, 1::4] << 2 | codes[:, 2::4] << 4 | codes[:, 3::4] << 6).astype(np.uint8).reshape(-1)
            scales = (1.0 / ((np.arange(n * (k // block)) % 100) + 1)).astype(np.float32)
            x = rng.uniform(-4, 4, m * k).astype(np.float32)
            got = hip.i2s_matmul_f32(x, packed, scales, m, n, k, block)
            want = oracle.i2s_matmul(x, packed, scales, m, n, k, block)
            if not np.all(np.abs(got - want) <= 3e-5 * max(1.0, np.max(np.abs(want))) + 2e-4):
                fail("i2s_matmul_f32", m, n, k, block, float(np.max(np.abs(got - want))))
        else:  # gemv_qk256 host-pointer drop-in incl. ragged tails
            rows, cols = int(rng.choice([1, 2, 17, 300, 640])), int(rng.choice([4, 256, 260, 300, 511, 512, 2560, 2563]))
            stride = -(-cols // 256) * 64
            qs = rng.integers(0, 256, rows * stride, dtype=np.uint8)
            x = rng.uniform(-10, 10, cols).astype(np.float32)
            got = hip.gemv_qk256(qs, x, rows, cols, stride)
            want = oracle.gemv_qk256(qs, x, rows, cols, stride)
            tol = min(2e-4 * np.sqrt(cols / 256.0), 1e-3)
            d = np.abs(got.astype(np.float64) - want)
            if not np.all((d < tol) | (d / np.maximum(np.maximum(np.abs(got), np.abs(want)), 1e-30) < 2e-2)):
                fail("gemv_qk256", rows, cols, float(d.max()))
    except pkg.BitNetHipError as e:
        fail(kind, repr(e))
print(f"{n_cases - bad}/{n_cases} cases agree", flush=True)
sys.exit(1 if bad else 0)
