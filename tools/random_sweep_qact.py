"""Developer tool (GPU box): random supported shapes through bitnet_hip_gemv_q_dev (QAct in, f32 / QAct out, optional LayerNorm
applied after the product, residual) against W . dequantise(QAct) in f64 -- the differential companion of tests/test_qact_gpu.py.
python tools/random_sweep_qact.py [n] [seed]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bitnet-rs_amd")
from tests.qact_ref import dequantize_qact, quantize_qact, QREC  # noqa: E402

hip = pkg.load(); hip.init(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
bad = 0
for case in range(n_cases):
    fmt = rng.choice(["qk256", "f16", "f32"])
    rows = 16 * int(rng.choice([1, 2, 3, 7, 10, 40, 63, 160, 161, 240, 432]))
    cols = 256 * int(rng.choice([1, 2, 3, 5, 8, 10, 16, 27]))
    p = rng.integers(0, 256, rows * cols // 4, dtype=np.uint8)
    codes = np.stack([(p.reshape(rows, cols // 4) >> (2 * i)) & 3 for i in range(4)], axis=-1).reshape(rows, cols)
    if fmt == "qk256":
        h = hip.weights_upload_qk256(p, rows, cols, cols // 4)
        wd = np.array([-2, -1, 1, 2], np.float64)[codes]
    else:
        scales = rng.uniform(0.01, 2.0, rows * cols // 32).astype(np.float32)
        if fmt == "f16":
            scales = scales.astype(np.float16).astype(np.float32)
        h = hip.weights_upload_i2s(p, scales, rows, cols, 32)
        wd = np.array([0, 1, 0, -1], np.float64)[codes] * np.repeat(scales.reshape(rows, cols // 32).astype(np.float64), 32, axis=1)
    if not hip.gemv_q_supported(h):
        print("unsupported", fmt, rows, cols); hip.weights_free(h); continue
    ln, res, qout = bool(rng.integers(0, 2)) and cols <= 4096, bool(rng.integers(0, 2)), bool(rng.integers(0, 2))  # LN rows: <= 4096 columns (header)
    x = (rng.normal(0.1, 1, cols) * rng.choice([0.01, 1.0, 30.0], cols)).astype(np.float32)
    gam = rng.uniform(0.5, 1.5, cols).astype(np.float32)
    resid = rng.normal(0, 1, rows).astype(np.float32)
    q = torch.zeros(hip.qact_bytes(cols), dtype=torch.uint8, device="cuda")
    st = torch.zeros(cols // 16 * 2, dtype=torch.float64, device="cuda")
    hip.quantize_act_dev(dev(x), dev(gam) if ln else None, cols, q, st)
    gd = dev(gam)
    if ln:
        hip.weights_bind_ln(h, gd)
    y = torch.full((rows,), float("nan"), device="cuda")
    qo = torch.zeros(hip.qact_bytes(rows), dtype=torch.uint8, device="cuda")
    go = rng.uniform(0.5, 1.5, rows).astype(np.float32)
    try:
        hip.gemv_q_dev(h, q, y=y, stats_in=st if ln else None, ln_gamma=gd if ln else None, ln_eps=1e-5, residual=dev(resid) if res else None,
                       qact_out=qo if qout else None, gamma_out=dev(go) if qout else None)
        torch.cuda.synchronize()
        got = y.cpu().numpy().astype(np.float64)
        uq = dequantize_qact(q.cpu().numpy(), cols)  # = gamma * x (LN) or x, as quantised
        if ln:
            x64 = x.astype(np.float64)
            mean = x64.mean(); denom = np.sqrt(((x64 - mean) ** 2).mean() + 1e-5)
            want = (wd @ uq - mean * (wd @ gam.astype(np.float64))) / denom
            scale_ref = (np.abs(wd) @ np.abs(uq) + abs(mean) * (np.abs(wd) @ gam)) / denom
        else:
            want = wd @ uq
            scale_ref = np.abs(wd) @ np.abs(uq)
        if res:
            want = want + resid
        err = np.abs(got - want)
        ok = np.isfinite(got).all() and np.all(err <= 4e-6 * scale_ref + 1e-6 * np.abs(want) + 1e-30)
        if ok and qout:
            ok = np.array_equal(qo.cpu().numpy()[: (rows + 255) // 256 * QREC], quantize_qact(y.cpu().numpy(), go)[: (rows + 255) // 256 * QREC])
        detail = float(np.max(err / (4e-6 * scale_ref + 1e-6 * np.abs(want) + 1e-30)))
    except pkg.BitNetHipError as e:
        ok, detail = False, repr(e)
    if not ok:
        bad += 1
        print("FAIL", fmt, rows, cols, "ln" if ln else "", "res" if res else "", "qout" if qout else "", detail, flush=True)
    hip.weights_free(h)
print(f"{n_cases - bad}/{n_cases} cases agree", flush=True)
sys.exit(1 if bad else 0)
