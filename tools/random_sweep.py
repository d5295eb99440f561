"""Developer tool (GPU box): random shapes through the device-resident GEMV / matmul entry points against the CPU oracle --
a differential sweep beyond the fixed shapes of tests/ (any failure here becomes a test case).  python tools/random_sweep.py [n] [seed]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bitnet-rs_amd")
from oracle import oracle  # noqa: E402  (checker)

hip = pkg.load(); hip.init(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n_cases):
    fmt = rng.choice(["qk256", "i2s32", "i2s32h", "i2s256"])
    rows = int(rng.choice([1, 3, 16, 17, 48, 100, 256, 333, 640, 1000, 2560]))
    cols = int(rng.choice([4, 32, 36, 256, 260, 300, 512, 1024, 2560, 2816, 6912])) if fmt == "qk256" else int(rng.choice([32, 64, 256, 288, 512, 1024, 2560, 6912]))
    m = int(rng.choice([1, 1, 1, 2, 15, 16, 17, 40, 130]))
    if fmt == "i2s256" and cols % 256:
        cols = 256 * (cols // 256 + 1)
    x = rng.uniform(-4, 4, (m, cols)).astype(np.float32)
    if fmt == "qk256":
        stride = -(-cols // 256) * 64
        qs = rng.integers(0, 256, rows * stride, dtype=np.uint8)
        h = hip.weights_upload_qk256(qs, rows, cols, stride)
        want = np.stack([oracle.gemv_qk256(qs, x[i], rows, cols, stride) for i in range(m)])
    else:
        block = 256 if fmt == "i2s256" else 32
        codes = rng.choice(np.array([0, 1, 3], np.uint8), size=(rows, cols), p=[0.5, 0.25, 0.25])
        packed = (codes[:, 0::4] | codes[:, 1::4] << 2 | codes[:, 2::4] << 4 | codes[:, 3::4] << 6).astype(np.uint8).reshape(-1)
        scales = rng.uniform(0.05, 1.5, rows * (cols // block)).astype(np.float32)
        if fmt == "i2s32h":
            scales = scales.astype(np.float16).astype(np.float32)
        h = hip.weights_upload_i2s(packed, scales, rows, cols, block)
        want = oracle.i2s_matmul(x.reshape(-1), packed, scales, m, rows, cols, block).reshape(m, rows)
    xd = torch.from_numpy(x).cuda()
    yd = torch.full((m, rows), float("nan"), device="cuda")
    try:
        if m == 1:
            hip.gemv_dev(h, xd, yd)
        else:
            hip.matmul_dev(h, xd, yd, m)
        torch.cuda.synchronize()
        got = yd.cpu().numpy()
        tol = 3e-5 * max(1.0, float(np.max(np.abs(want)))) + 2e-4 * np.sqrt(cols / 256.0)
        err = float(np.max(np.abs(got - want)))
        ok = np.isfinite(got).all() and err <= tol
    except Exception as e:  # noqa: BLE001
        ok, err, tol = False, repr(e), 0
    if not ok:
        bad += 1
        print("FAIL", fmt, rows, cols, m, err, tol, flush=True)
    hip.weights_free(h)
print(f"{n_cases - bad}/{n_cases} cases agree with the oracle", flush=True)
sys.exit(1 if bad else 0)
