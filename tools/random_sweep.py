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
    # half of the cases through the fused entry points: LayerNorm prologue (cols >= 32) and / or residual epilogue, random digit count
    fused = bool(rng.integers(0, 2))
    ln = fused and cols >= 32 and bool(rng.integers(0, 2))
    res = fused and bool(rng.integers(0, 2))
    digits = int(rng.choice([2, 3, 4]))
    # 2 digits: the form is the launcher's choice (0), the int8 planes on request (8: BITNET_HIP_FUSE_INT8_DIGITS) or, on unscaled
    # matrices, the fp6 x fp4 form (16: BITNET_HIP_FUSE_FP6_DIGITS)
    form = int(rng.choice([0, 8, 16] if fmt == "qk256" else [0, 8])) if digits == 2 else 0
    if fused:
        gam = rng.uniform(0.5, 1.5, cols).astype(np.float32)
        resid = rng.normal(0, 1, (m, rows)).astype(np.float32)
        xin = np.stack([oracle.layernorm(x[i], gam, 1e-5) for i in range(m)]) if ln else x
        if fmt == "qk256":
            want = np.stack([oracle.gemv_qk256(qs, xin[i], rows, cols, stride) for i in range(m)])
        else:
            want = oracle.i2s_matmul(xin.reshape(-1), packed, scales, m, rows, cols, block).reshape(m, rows)
        if res:
            want = want + resid
    try:
        if not fused:
            if m == 1:
                hip.gemv_dev(h, xd, yd)
            else:
                hip.matmul_dev(h, xd, yd, m)
        else:
            gd, rd = torch.from_numpy(gam).cuda(), torch.from_numpy(resid).cuda()
            if ln and bool(rng.integers(0, 2)):
                hip.weights_bind_ln(h, gd)  # LayerNorm applied after the product (bound g = W . gamma)
            if m == 1:
                hip.gemv_fused_dev(h, xd, yd, 1, gd if ln else None, 1e-5, rd if res else None, 0)
            else:
                wsb = hip.matmul_workspace_bytes(m, cols, digits)
                ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
                hip.matmul_fused_dev(h, xd, yd, m, ws, wsb, ln_gamma=gd if ln else None, ln_eps=1e-5, residual=rd if res else None, digits=digits, flags=form)
        torch.cuda.synchronize()
        got = yd.cpu().numpy()
        # 2 digits: 14-bit activations; on f16-scaled 32-blocks (cols % 256 == 0) the f16 matrix cores with f16 activations (2^-12 per element)
        f16w = fused and m > 1 and digits == 2 and fmt == "i2s32h" and cols % 256 == 0 and form != 8
        tol = (3e-5 if not (fused and m > 1 and digits == 2) else 1.5e-3 if f16w else 3e-4) * max(1.0, float(np.max(np.abs(want)))) + 2e-4 * np.sqrt(cols / 256.0)
        err = float(np.max(np.abs(got - want)))
        ok = np.isfinite(got).all() and err <= tol
    except pkg.BitNetHipError as e:
        ok, err, tol = False, repr(e), 0
    if not ok:
        bad += 1
        print("FAIL", fmt, rows, cols, m, "fused" if fused else "plain", "ln" if ln else "", "res" if res else "", digits, form, err, tol, flush=True)
    hip.weights_free(h)
print(f"{n_cases - bad}/{n_cases} cases agree with the oracle", flush=True)
sys.exit(1 if bad else 0)
